"""Process-group bookkeeping of the stand-in: GroupCoordinator over torch.distributed groups."""
import logging
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, List, Optional

import torch
import torch.distributed as dist

logger = logging.getLogger("vllm.distributed.parallel_state")

_WORLD = None
_TP = None
_PP = None
_DP = None
_EP = None
_TP_STATE_PATCHED = False
# (stand-in only) a one-process REHEARSAL of an N-rank layout: set to N before the groups are built and every
# GroupCoordinator becomes rank 0 of a group whose other members do not exist (no torch.distributed groups are made);
# `device_group` is then a VirtualGroup and the test supplies the collectives (mirror ranks: every peer holds what this
# rank holds).  Lets a single GPU process walk the SP / shift code paths with collectives that a HIP graph can capture.
VIRTUAL_WORLD = None


class VirtualGroup:
    def __init__(self, size: int):
        self.size = size

    def __repr__(self):
        return f"VirtualGroup({self.size})"


class GroupCoordinator:
    def __init__(self, group_ranks: List[List[int]], local_rank: int, backend: Optional[str], group_name: str,
                 use_message_queue_broadcaster: bool = False):
        self.unique_name = group_name
        self.rank = dist.get_rank() if dist.is_initialized() and not VIRTUAL_WORLD else 0
        self.local_rank = local_rank
        self.device_group = None
        self.ranks = None
        self.all_group_ranks = group_ranks
        for ranks in group_ranks:                      # every rank creates every group, in the same order
            if VIRTUAL_WORLD:
                g = VirtualGroup(len(ranks)) if len(ranks) > 1 else None
            else:
                g = dist.new_group(ranks, backend=backend) if dist.is_initialized() and len(ranks) > 1 else None
            if self.rank in ranks:
                self.ranks, self.device_group = ranks, g
        assert self.ranks is not None, (group_name, self.rank, group_ranks)
        self.world_size = len(self.ranks)
        self.rank_in_group = self.ranks.index(self.rank)
        self.use_message_queue_broadcaster = use_message_queue_broadcaster
        self.captures = 0
        self.destroyed = False

    @property
    def is_first_rank(self) -> bool:
        return self.rank_in_group == 0

    @property
    def is_last_rank(self) -> bool:
        return self.rank_in_group == self.world_size - 1

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1 and isinstance(self.device_group, VirtualGroup):
            return t.mul_(self.world_size)             # mirror ranks: the sum of world_size equal shares
        if self.world_size > 1:
            if t.is_cuda and dist.get_backend(self.device_group) == "gloo":     # ranks sharing one GPU in tests
                host = t.cpu()
                dist.all_reduce(host, group=self.device_group)
                t.copy_(host)
            else:
                dist.all_reduce(t, group=self.device_group)
        return t

    @contextmanager
    def graph_capture(self, context=None):
        self.captures += 1
        yield context

    def destroy(self):
        self.destroyed = True

    def send_tensor_dict(self, *a, **k):
        raise NotImplementedError

    def broadcast_tensor_dict(self, d, src=0):
        return d


@dataclass
class GraphCaptureContext:
    stream: Any


def init_model_parallel_group(group_ranks, local_rank, backend, use_message_queue_broadcaster=False, group_name=None):
    return GroupCoordinator(group_ranks, local_rank, backend, group_name, use_message_queue_broadcaster)


def init_world_group(local_rank: int = 0):
    global _WORLD
    n = VIRTUAL_WORLD or (dist.get_world_size() if dist.is_initialized() else 1)
    _WORLD = GroupCoordinator([list(range(n))], local_rank, dist.get_backend() if dist.is_initialized() else None, "world")
    return _WORLD


def get_world_group():
    assert _WORLD is not None
    return _WORLD


def get_tp_group():
    assert _TP is not None
    return _TP


def get_pp_group():
    assert _PP is not None
    return _PP


def initialize_model_parallel(tensor_model_parallel_size: int = 1, pipeline_model_parallel_size: int = 1,
                              backend: Optional[str] = None) -> None:
    global _TP, _PP, _DP, _EP
    n = VIRTUAL_WORLD or (dist.get_world_size() if dist.is_initialized() else 1)
    tp, pp = tensor_model_parallel_size, pipeline_model_parallel_size
    lr = get_world_group().local_rank
    _TP = init_model_parallel_group([list(range(i, i + tp)) for i in range(0, n, tp)], lr, backend, True, "tp")
    _PP = init_model_parallel_group([list(range(i, n, n // pp)) for i in range(n // pp)], lr, backend, group_name="pp")
    _DP = init_model_parallel_group([[r] for r in range(n)], lr, backend, group_name="dp")
    _EP = init_model_parallel_group([list(range(i, i + tp)) for i in range(0, n, tp)], lr, backend, group_name="ep")


@contextmanager
def graph_capture(device):
    context = GraphCaptureContext(torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None)
    with _TP.graph_capture(context), _PP.graph_capture(context):
        yield context


def destroy_model_parallel():
    global _TP, _PP, _DP, _EP
    for g in (_TP, _PP, _DP, _EP):
        if g:
            g.destroy()
    _TP = _PP = _DP = _EP = None


def destroy_distributed_environment():
    global _WORLD
    _WORLD = None


def reset_for_tests():
    """(stand-in only) forget every group, including the ones the plugin adds."""
    import sys
    mod = sys.modules[__name__]
    for name in ("_WORLD", "_TP", "_PP", "_DP", "_EP", "_SP", "_SP_TP", "_SP_AA", "_SP_AG", "_ORIG_TP"):
        if hasattr(mod, name):
            setattr(mod, name, None)
