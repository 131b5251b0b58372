from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class SamplerOutput:
    sampled_token_ids: Any
    logprobs_tensors: Any = None


@dataclass
class ModelRunnerOutput:
    req_ids: list
    req_id_to_index: dict
    sampled_token_ids: list
    spec_token_ids: Optional[list]
    logprobs: Any
    prompt_logprobs_dict: dict
    pooler_output: list = field(default_factory=list)
    finished_sending: Any = None
    finished_recving: Any = None
    num_nans_in_logits: Any = None


EMPTY_MODEL_RUNNER_OUTPUT = ModelRunnerOutput(req_ids=[], req_id_to_index={}, sampled_token_ids=[], spec_token_ids=None,
                                              logprobs=None, prompt_logprobs_dict={})
