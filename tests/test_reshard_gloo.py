"""CPU, world_size 2 and 4 over gloo: ulysses.ReshardStream — the SP -> replicated resharding of a step's hidden states
(the reference's synchronous all-gather, model_runner.py:202-209) in its full form and in its row form (only the rows the
step samples from, one all-reduce over int32 bit patterns).  Both must equal the literal expression
`all_gather(local)[rows]`, bit for bit, -0.0 and NaN payloads included."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from arcticinference_amd.ulysses import ReshardStream, pad_tokens_for_sp
        ok = True
        for N_raw, H, dtype in ((37, 64, torch.bfloat16), (4096, 128, torch.bfloat16), (64, 32, torch.float32)):
            N = pad_tokens_for_sp(N_raw, world)
            n = N // world
            g = torch.Generator().manual_seed(N)
            full = torch.randn(N, H, generator=g).to(dtype)
            full[3, 5] = -0.0                        # the sign of a zero survives the row form's integer sum
            full[N - 1, 0] = float("nan")
            local = full[rank * n:(rank + 1) * n].clone()
            rs = ReshardStream(world, rank, dist.group.WORLD)
            got = rs.gather(local, N).wait()
            ok = ok and torch.equal(got.view(torch.uint8), full.view(torch.uint8))
            for R in (1, 5, min(64, N)):
                rows = torch.randperm(N, generator=g)[:R].sort().values
                rows[0] = 3 if R > 1 else rows[0]
                rows[-1] = N - 1
                got = rs.gather(local, N, rows=rows).wait()
                ok = ok and got.shape == (R, H) and torch.equal(got.view(torch.uint8), full[rows].view(torch.uint8))
            # the row form was really taken where it moves fewer bytes, the full form where it does not
            took_rows = rs.calls["rows"]
            ok = ok and (took_rows > 0 if N >= 4096 else True) and rs.calls["full"] >= 1
        out_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_reshard_stream_forms_equal_the_all_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world)) and all(r[1] for r in res), res


def test_reshard_stream_single_process_rehearsal():
    """group = None (bench.py --rehearse-sp): local copies, same shapes."""
    from arcticinference_amd.ulysses import ReshardStream
    rs = ReshardStream(4, 0, None)
    local = torch.arange(8 * 6, dtype=torch.float32).view(8, 6)
    full = rs.gather(local, 32).wait()
    assert full.shape == (32, 6) and torch.equal(full[8:16], local)
    rows = torch.tensor([0, 9, 31])
    assert rs.gather(local, 32, rows=rows).wait().shape == (3, 6)
