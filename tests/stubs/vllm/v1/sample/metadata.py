from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class SamplingMetadata:
    temperature: Any = None
    all_greedy: bool = True
    all_random: bool = False
    top_p: Any = None
    top_k: Any = None
    generators: dict = field(default_factory=dict)
    max_num_logprobs: Optional[int] = None
    no_penalties: bool = True
    allowed_token_ids_mask: Any = None
    bad_words_token_ids: dict = field(default_factory=dict)
    logit_bias: list = field(default_factory=list)
