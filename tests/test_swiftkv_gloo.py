"""CPU, world_size 2 over gloo: SwiftKV's all-gather of the prefill half's outputs over the SP group (C7,
llama_swiftkv.py:250-257): rank-major concatenation along the token dimension for every tensor."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from arcticinference_amd.swiftkv import sp_all_gather
        n = 5
        hid = torch.arange(n * 4, dtype=torch.float32).view(n, 4) + 100 * rank
        pos = torch.arange(n) + 10 * rank
        h, p = sp_all_gather([hid[:, :3], pos], world, dist.group.WORLD)      # a non-contiguous input too
        want_h = torch.cat([torch.arange(n * 4, dtype=torch.float32).view(n, 4)[:, :3] + 100 * r for r in range(world)])
        want_p = torch.cat([torch.arange(n) + 10 * r for r in range(world)])
        same = sp_all_gather([hid], 1, None)[0]
        # the seam itself refuses strided buffers on every backend (RCCL would; the staged gloo path must not hide it)
        from arcticinference_amd import dist_utils
        refused = 0
        for call in (lambda: dist_utils.all_reduce(hid[:, :3]),
                     lambda: dist_utils.all_gather_into_tensor(torch.empty(world * n, 3), hid[:, :3]),
                     lambda: dist_utils.all_to_all_single(torch.empty(n, 4)[:, :2], hid[:, :2])):
            try:
                call()
            except ValueError:
                refused += 1
        q.put((rank, bool(torch.equal(h, want_h) and torch.equal(p, want_p) and same is hid and refused == 3)))
    finally:
        dist.destroy_process_group()


def test_sp_all_gather_is_rank_major():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(30)
    assert all(ok for _, ok in res)
