"""Diagnostic (GPU): the attention calls of the (f)-2 capture test's eager run, each checked against the fp32 oracle in
both call forms (host-partitioned lists / device geometry), per request."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), ROOT]
import torch  # noqa: E402
import vllm_harness as H  # noqa: E402

H.install()
import test_vllm_capture_gpu as T  # noqa: E402
import torch.distributed as dist  # noqa: E402
from vllm.compilation import cuda_graphs  # noqa: E402
from vllm.config import CompilationLevel  # noqa: E402

H.load_plugin()
import socket  # noqa: E402
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)


class MP:
    def setattr(self, o, n, v):
        setattr(o, n, v)


mp = MP()
mp.setattr(dist, "get_world_size", lambda group=None: T.SP)
T._mirror_collectives(mp)
from arcticinference_amd import ops  # noqa: E402
from oracle import spec_oracle as O  # noqa: E402

real = ops.verify_attention
count = [0]


def checked(q, k_cache, v_cache, block_table, seq_lens, qsl, max_q, max_seq, scale, out=None, req_split=None, **kw):
    count[0] += 1
    res = real(q, k_cache, v_cache, block_table, seq_lens, qsl, max_q, max_seq, scale, out=out, req_split=req_split, **kw)
    if count[0] <= 4:
        torch.cuda.synchronize()
        want = O.verify_attention(q.float().cpu().to(torch.bfloat16), k_cache.cpu(), v_cache.cpu(), block_table.cpu(),
                                  seq_lens.cpu().tolist(), qsl.cpu().tolist(), scale)
        gen = real(q, k_cache, v_cache, block_table, seq_lens, qsl, max_q, max_seq, scale, req_split=None, **kw)
        torch.cuda.synchronize()
        e1 = (res.float().cpu() - want).abs().amax(dim=(1, 2))
        e2 = (gen.float().cpu() - want).abs().amax(dim=(1, 2))
        print("call", count[0], "q", tuple(q.shape), "stride", q.stride(), "B", seq_lens.numel(), "seq", seq_lens.cpu().tolist(),
              "split", None if req_split is None else (req_split[1], req_split[3]))
        qs = qsl.cpu().tolist()
        per_req = lambda e: [round(float(e[qs[i]:qs[i + 1]].max()), 4) for i in range(len(qs) - 1)]
        print("   partitioned err/request", per_req(e1))
        print("   generic     err/request", per_req(e2))
        bad = [i for i in range(len(e2)) if e2[i] > 2e-3 or e1[i] > 2e-3]
        print("   bad tokens", bad[:40])
    return res


ops.verify_attention = checked


class Hook:
    hidden = None

    def __call__(self, h, l):
        self.hidden = h.float().cpu()
        return l


cuda_graphs.enabled = False
r = T._runner(CompilationLevel.PIECEWISE, True)
hook = Hook()
r.model.logit_hook = hook
T._drive(r, lambda: hook.hidden)
