"""tests-only stand-in package (see tests/stubs/README.md).  Field names and "nothing set" values follow what vLLM 0.9.2's
InputBatch._make_sampling_metadata builds: temperature / top_p / top_k / min_p are None when no request uses them,
logit_bias is ALWAYS a list with one entry (None) per request, min_tokens / bad_words_token_ids are dicts keyed by
batch index, generators holds the seeded requests' torch.Generators."""
from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class SamplingMetadata:
    temperature: Any = None            # f32 [num_reqs]; greedy rows carry -1.0; None when all_greedy
    all_greedy: bool = True
    all_random: bool = False
    top_p: Any = None
    top_k: Any = None
    min_p: Any = None
    generators: dict = field(default_factory=dict)
    max_num_logprobs: Optional[int] = None
    no_penalties: bool = True
    prompt_token_ids: Any = None
    frequency_penalties: Any = None
    presence_penalties: Any = None
    repetition_penalties: Any = None
    output_token_ids: list = field(default_factory=list)
    min_tokens: dict = field(default_factory=dict)
    logit_bias: list = field(default_factory=list)
    allowed_token_ids_mask: Any = None
    bad_words_token_ids: dict = field(default_factory=dict)
