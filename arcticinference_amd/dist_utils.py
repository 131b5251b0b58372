"""The two collectives of the N > 1 path behind one seam.  With the `nccl` (= RCCL) backend they are the plain
torch.distributed calls on device tensors.  With `gloo` — the CPU tests, and the multi-process rehearsal of bench.py on a
box whose ranks share one GPU, where RCCL refuses duplicate devices — device tensors are staged through host memory, so
the same code path (group logic, buffer shapes, ordering of collectives across ranks) runs without RCCL."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _staged(group) -> bool:
    return dist.get_backend(group) == "gloo"


def _need_dense(*tensors: torch.Tensor) -> None:
    """RCCL takes dense buffers only; checked on every backend so that the gloo rehearsal cannot pass a view that the
    real group would refuse."""
    for t in tensors:
        if not t.is_contiguous():
            raise ValueError("collective buffers must be contiguous (got shape %s, strides %s)" % (tuple(t.shape), t.stride()))


def all_gather_into_tensor(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    _need_dense(out, inp)
    if inp.is_cuda and _staged(group):
        n = dist.get_world_size(group)
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(n)]
        dist.all_gather(parts, inp.cpu(), group=group)
        out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.shape))
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def all_to_all_single(recv: torch.Tensor, send: torch.Tensor, group=None) -> None:
    _need_dense(recv, send)
    if send.is_cuda and _staged(group):
        n = dist.get_world_size(group)
        src = send.cpu().contiguous()
        ins = list(src.chunk(n))
        outs = [torch.empty_like(c) for c in ins]
        # gloo has no all-to-all for every dtype / build: n gathers of one chunk each do the same exchange
        me = dist.get_rank(group)
        for r in range(n):
            got = [torch.empty_like(ins[r]) for _ in range(n)] if me == r else None
            dist.gather(ins[r], got, dst=dist.get_global_rank(group, r) if group is not None else r, group=group)
            if me == r:
                outs = got
        recv.copy_(torch.cat(outs).view(recv.shape))
        return
    dist.all_to_all_single(recv, send, group=group)


def all_reduce(t: torch.Tensor, group=None) -> None:
    """Sum in place."""
    _need_dense(t)
    if t.is_cuda and _staged(group):
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
        return
    dist.all_reduce(t, group=group)
