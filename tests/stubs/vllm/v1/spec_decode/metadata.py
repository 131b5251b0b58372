from dataclasses import dataclass

import torch


@dataclass
class SpecDecodeMetadata:
    draft_token_ids: torch.Tensor          # int32 [num_draft_total]
    num_draft_tokens: list                 # per request
    cu_num_draft_tokens: torch.Tensor      # int32 [B], inclusive cumulative sum
    target_logits_indices: torch.Tensor    # rows of the sampled logits that verify a draft token
    bonus_logits_indices: torch.Tensor     # last sampled row of each request
    logits_indices: torch.Tensor           # rows of the model output that are sampled at all

    def __post_init__(self):
        self.max_spec_len = max(self.num_draft_tokens)
