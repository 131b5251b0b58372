from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class NewRequestData:
    req_id: str
    prompt_token_ids: list
    block_ids: list
    num_computed_tokens: int = 0
    sampling_params: Any = None


@dataclass
class CachedRequestData:
    req_id: str
    resumed_from_preemption: bool
    new_token_ids: list
    new_block_ids: list
    num_computed_tokens: int


@dataclass
class SchedulerOutput:
    scheduled_new_reqs: list
    scheduled_cached_reqs: list
    num_scheduled_tokens: dict
    total_num_scheduled_tokens: int
    scheduled_spec_decode_tokens: dict = field(default_factory=dict)
    finished_req_ids: set = field(default_factory=set)
    grammar_bitmask: Optional[Any] = None
