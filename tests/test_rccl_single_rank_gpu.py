"""RCCL executes the collective seam.  A one-GPU box cannot form a group of two RCCL ranks (RCCL refuses two ranks on one
device; the multi-rank runs are the gloo tests and rehearsals), but a ONE-rank `nccl` group still goes through RCCL's
initialisation and its all-to-all / all-gather / all-reduce entry points with the buffers, dtypes and sub-groups the N > 1
path uses — which catches a backend that is missing, refuses a dtype or trips over the process environment before the
driver's multi-GPU bench does.  Runs in a child process (a process group is process-wide state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["AIC_ROOT"])
from arcticinference_amd import dist_utils as D
from arcticinference_amd.ulysses import UlyssesAttention

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % int(os.environ["AIC_PORT"]), rank=0, world_size=1,
                        device_id=dev)
assert dist.get_backend() == "nccl"
sub = dist.new_group([0])                       # the SP_AA / SP_AG sub-groups of the KV-replicated layout are made this way
g = torch.Generator(device="cuda").manual_seed(0)
for group in (None, sub):
    # all-to-all of the packed [SP * n, qw + 2 kw] bf16 rows (C1 / C2)
    send = torch.randn(96, 6144, device=dev, generator=g).to(torch.bfloat16)
    recv = torch.empty_like(send)
    D.all_to_all_single(recv, send, group=group)
    assert torch.equal(recv, send)
    # hidden-state all-gather (C6) and the draft head's packed (value, index) int64 all-gather
    for t in (torch.randn(24, 4096, device=dev, generator=g).to(torch.bfloat16),
              torch.randint(-2**62, 2**62, (128,), device=dev, generator=g, dtype=torch.int64)):
        out = torch.empty_like(t)
        D.all_gather_into_tensor(out, t, group=group)
        assert torch.equal(out, t)
    # sharded-embedding all-reduce (C9)
    z = torch.randn(64, 4096, device=dev, generator=g).to(torch.bfloat16)
    want = z.clone()
    D.all_reduce(z, group=group)
    assert torch.equal(z, want)
    # strided views are refused before RCCL sees them
    try:
        D.all_to_all_single(recv[:, :128], send[:, :128], group=group)
    except ValueError:
        pass
    else:
        raise AssertionError("a strided buffer reached the collective")
# the attention wrapper over the real group (sp = 1: no exchange, the local kernel's result as it is)
ua = UlyssesAttention(1, dist.group.WORLD, 8, 2, 128)
q = torch.randn(4, 8 * 128, device=dev, generator=g).to(torch.bfloat16)
k = torch.randn(4, 2 * 128, device=dev, generator=g).to(torch.bfloat16)
assert torch.equal(ua.forward(q, k, k, lambda a, b, c: a * 2), q * 2)
# the resharding stream (ulysses.ReshardStream): both forms over the real group on a side HIP stream, with main-stream work
# enqueued between gather() and wait() — the int32 all-reduce of the row form goes through RCCL here
from arcticinference_amd.ulysses import ReshardStream
rs = ReshardStream(1, 0, dist.group.WORLD, dev)
local = torch.randn(300, 4096, device=dev, generator=g).to(torch.bfloat16)
local[7, 9] = -0.0
busy = torch.randn(2048, 2048, device=dev, generator=g)
h1 = rs.gather(local, 300)
rows = torch.tensor([0, 7, 150, 299], device=dev)
h2 = rs.gather(local, 300, rows=rows)
for _ in range(4):
    busy = busy @ busy.t() * 1e-3                   # runs on the main stream while the side stream reshards
full, picked = h1.wait(), h2.wait()
assert torch.equal(full.view(torch.int16), local.view(torch.int16))
assert torch.equal(picked.view(torch.int16), local[rows].view(torch.int16))
assert rs._stream is not None and rs._stream != torch.cuda.current_stream() and rs.calls == {"full": 2, "rows": 0}
rs4 = ReshardStream(4, 0, None, dev)                # rehearsal form on the device: the row form is taken (2 R < 3 n)
p4 = rs4.gather(local, 1200, rows=rows).wait()
assert p4.shape == (4, 4096) and rs4.calls["rows"] == 1
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK")
"""


def test_one_rank_rccl_group_runs_the_collective_seam():
    import socket
    import torch
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AIC_ROOT=root, AIC_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("extra", [[], ["--no-shift-parallel"]])
def test_bench_runs_through_a_one_rank_rccl_group(extra):
    """bench.py's N > 1 code (process group, Ulysses context over it, barrier, max-over-ranks reduction on the device) with
    RCCL underneath, on a group of one rank: shift mode, and the all-to-all path (`--no-shift-parallel`)."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AIC_BENCH_ONE_RANK_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--prompt-len", "512",
           "--gen-len", "64", "--layers", "4", "--no-cpu-baseline", "--no-replay-check"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # (a group of one never shifts: there is no second replica to shift to; both invocations take the SP code path, the
    # second one with the extra all-to-all-path steps of bench.py switched on as well)
    assert line["steps_in_sp_mode"] + line["steps_in_shift_mode"] > 0
