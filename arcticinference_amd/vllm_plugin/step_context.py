"""Per-step geometry the model-runner patch publishes for the attention patch.

vLLM's Attention.forward sees device tensors only; the runner knows on the HOST how many tokens each request of
the step has (scheduler_output.num_scheduled_tokens).  aic_verify_attention_ex wants the batch partitioned by
query length (short requests / long suffix drafts, include/arctic_hip.h), so the runner publishes the step's
per-request query lengths here and every attention layer of the step reuses one partition built from them.
One runner per process (vLLM worker), one step at a time: a module-level slot is the whole mechanism."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

_q_lens: Optional[np.ndarray] = None
_shift_mode: bool = False
_splits: Dict[Tuple[int, int], object] = {}     # (group size, device index) -> ops.split_requests(...) of this step
calls = {"verify": 0, "fallback": 0}          # attention calls routed to the HIP kernel / left to vLLM's backend


def publish(num_scheduled_tokens: np.ndarray, shift_mode: bool) -> None:
    global _q_lens, _shift_mode
    _q_lens = np.asarray(num_scheduled_tokens)
    _shift_mode = bool(shift_mode)
    _splits.clear()


def clear() -> None:
    global _q_lens
    _q_lens = None
    _splits.clear()


def q_lens() -> Optional[np.ndarray]:
    return _q_lens


def shift_mode() -> bool:
    return _shift_mode


def request_split(group_size: int, device):
    """(short ids, n_short, long ids, n_long) for the step, or None when every request is short / nothing is published."""
    if _q_lens is None:
        return None
    key = (group_size, device.index)
    if key not in _splits:
        from .. import ops
        _splits[key] = ops.split_requests(_q_lens, group_size, device)
    return _splits[key]
