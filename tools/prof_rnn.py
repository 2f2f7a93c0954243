"""LAS_PROF build only: dump s_memtime stamps of the recurrent sweep (wave 0 of workgroup 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
os.environ["LAS_DBG_PTR"] = hex(dbg.data_ptr())
from las import _hip
for cell in (1, 0):
    G = 4 if cell else 1
    B, T, H = 48, 1274, 256
    g = torch.Generator().manual_seed(0)
    xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).cuda()
    w0 = (torch.randn(H, G * H, generator=g) * 0.05).cuda(); w1 = w0.clone()
    out = torch.zeros(B, T, 2 * H, device="cuda"); cst = torch.zeros(B, T, 2, H, device="cuda") if cell else None
    for _ in range(2):
        dbg.zero_()
        _hip.rnn_seq_fwd(cell, 1, B, T, H, xp.clone(), w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
        torch.cuda.synchronize()
    d = dbg.cpu().tolist()
    cyc, wall = d[2] - d[0], d[3] - d[1]
    print("cell", cell, "total shader cycles", cyc, "wall ticks(100MHz)", wall, "=> shader clock %.2f GHz, %.0f cycles/step" % (cyc / (wall * 10.0), cyc / T))
    for s in range(8):
        st = d[8 + s * 8: 8 + s * 8 + 5]
        nxt = d[8 + (s + 1) * 8] if s < 7 else None
        print("  step", 200 + s, "mfma-issue %5d  mfma-drain %5d  gate-math+stores %5d  barrier %5d" % (st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]),
              ("loop-top %5d" % (nxt - st[4])) if nxt else "")
