"""Experiment driver: the 8B speculator's draft at 32 rows (fp8 head), N calls in the given cell form (argv[1] = 1 | 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from arcticinference_amd import _native as N
from arcticinference_amd.speculator import ArcticLSTMSpeculator, LSTMSpeculatorConfig, random_lstm_weights
cell = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = LSTMSpeculatorConfig(vocab_size=128256, input_hidden_dim=4096)
m = ArcticLSTMSpeculator(cfg, max_num_seqs=64, device="cuda", quantize_lm_head=True, use_graph=False)
m.load_weights(random_lstm_weights(cfg, seed=0).items())
hid = torch.randn(B, 4096, device="cuda", dtype=torch.bfloat16)
ids = torch.randint(0, 128256, (B,), device="cuda")
N.lib().aic_debug_lstm_cell_launches(cell)
for _ in range(30):
    m.generate_proposals(ids, hid, 3)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "trace":
    import numpy as np
    buf = torch.zeros(3, 64, 4, 12, dtype=torch.int64, device="cuda")
    N.lib().aic_debug_lstm_cell_trace(buf.data_ptr())
    m.generate_proposals(ids, hid, 3)
    torch.cuda.synchronize()
    N.lib().aic_debug_lstm_cell_trace(None)
    t = buf.cpu().numpy()[:, :B].astype(np.float64) / 100.0     # us
    names = ["issue independent loads", "argmax reduce + token", "z loads issued", "row pass + block_sum (waits for loads)",
             "own columns + block_sum", "row-share store + poll", "state + |max|", "batch |max| store + poll"]
    for h in range(3):
        t0 = t[h, :, :, 0].min()
        rel = t[h] - t0
        print(f"head {h}: workgroup start spread {rel[:, :, 0].max():.2f} us; kernel end (last stamp) median {np.median(rel[:, :, 7]):.2f} max {rel[:, :, 7].max():.2f}")
        for i in range(1, 8):
            d = t[h, :, :, i] - t[h, :, :, i - 1]
            print(f"   phase {i}: {names[i - 1]:45s} median {np.median(d):6.2f}  max {d.max():6.2f} us")
        print(f"   inside phase 6->7: compute loop {np.median(t[h, :, :, 8] - t[h, :, :, 5]):.2f}, wave reduce {np.median(t[h, :, :, 9] - t[h, :, :, 8]):.2f}, "
              f"barrier + h stores {np.median(t[h, :, :, 6] - t[h, :, :, 9]):.2f} us")
