"""Host-side cost of one per-layer call of the plugin's attention route (arcticinference_amd/vllm_plugin/ulysses.py: one KV
write + one verify-attention call per layer when vLLM runs the layer eagerly, as its piecewise graphs do): wall time per call
with the queue never full, i.e. argument marshalling + the launches.  MI355X, r03: verify_attention 18 us, KV writer 8.6 us."""
import time, torch, numpy as np, sys
sys.path.insert(0, "/root/repo")
from arcticinference_amd import ops, _native as N
dev = "cuda"
B, ctx, Hq, Hkv, D, bs = 64, 256, 32, 8, 128, 16
nblk = ctx // bs
kv = torch.randn(2, B * nblk, bs, Hkv, D, device=dev, dtype=torch.bfloat16)
bt = torch.arange(B * nblk, device=dev, dtype=torch.int32).view(B, nblk)
ql = [1] * 60 + [4] * 4
T = sum(ql)
q = torch.randn(T, Hq, D, device=dev, dtype=torch.bfloat16)
seq = torch.full((B,), ctx, dtype=torch.int32, device=dev)
qsl = torch.tensor(np.concatenate([[0], np.cumsum(ql)]).astype(np.int32), device=dev)
out = torch.empty_like(q)
rs = ops.split_requests(ql, Hq // Hkv, dev)
key = torch.randn(T, Hkv * D, device=dev, dtype=torch.bfloat16)
slots = torch.arange(T, device=dev, dtype=torch.int64)
w = ops.KvBulkWriter([kv[0]], [kv[1]], "auto", [torch.ones(1, device=dev)], [torch.ones(1, device=dev)], Hkv, D)
def host_time(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6
print("verify_attention host us/call (split lists):", host_time(lambda: ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, 4, ctx, D ** -0.5, out=out, req_split=rs)))
print("verify_attention host us/call (no lists):   ", host_time(lambda: ops.verify_attention(q, kv[0], kv[1], bt, seq, qsl, 4, ctx, D ** -0.5, out=out)))
print("kv writer host us/call:                     ", host_time(lambda: w(key, key, slots)))
print("torch.empty host us/call:                   ", host_time(lambda: torch.empty((T, Hq * D), dtype=torch.bfloat16, device=dev)))
