#!/usr/bin/env python
"""us per las_lstm_cell_rows launch at the beam search's shapes (HIP events, 200 launches back to back): LM layer 2 (x fp32 512 + h 512 -> 4 x 512),
LM layer 1 (one-hot + h 512), the Speller's cell (bf16 rows 1152 -> 4 x 512, fast) at M = 256 and 1024 rows.  LAS_LIB_PATH selects the build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from las import _hip
dev = "cuda"
lib = _hip.lib()
g = torch.Generator().manual_seed(0)

def timeit(fn, n=200):
    for _ in range(3): fn()                 # (the first launch of an instantiation sets its LDS attribute)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M in [int(v) for v in os.environ.get("MS", "256 1024").split()]:
    H, I = 512, 512
    kern = (torch.randn(I + H, 4 * H, generator=g) * 0.05).to(dev)
    bias = torch.zeros(4 * H, device=dev)
    x, h, c = (torch.randn(M, 512, generator=g).to(dev) for _ in range(3))
    ih, hh = _hip.skinny_pack(kern, I, 4 * H), _hip.skinny_pack(kern, H, 4 * H, row0=I)
    c1, h1 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    lm2 = lambda: lib.las_lstm_cell_rows(_hip.p(x), I, I, None, 0, None, _hip.p(h), H, _hip.p(ih), _hip.p(hh), _hip.p(bias), _hip.p(c), M, H, 0.0, _hip.p(c1), _hip.p(h1), _hip.stream())
    ids = torch.randint(0, 30, (M,), generator=g).to(torch.int32).to(dev)
    wx = torch.randn(28, 4 * H, generator=g).to(dev)
    lm1 = lambda: lib.las_lstm_cell_rows(None, 0, 0, _hip.p(ids), 2, _hip.p(wx), _hip.p(h), H, None, _hip.p(hh), _hip.p(bias), _hip.p(c), M, H, 0.0, _hip.p(c1), _hip.p(h1), _hip.stream())
    I0 = 1152
    k0 = (torch.randn(I0, 4 * H, generator=g) * 0.05).to(dev)
    p0 = _hip.skinny_pack(k0, I0, 4 * H)
    xb = torch.randn(M, I0, generator=g).to(dev).to(torch.bfloat16).contiguous()
    gt = torch.empty(M, 4 * H, device=dev)
    a = _hip.LstmCellArgs()
    a.x, a.x_bf16, a.ldx, a.I = xb.data_ptr(), 1, I0, I0
    a.h, a.Wx, a.Wh, a.bias, a.c_prev, a.fb = None, p0.data_ptr(), None, bias.data_ptr(), c.data_ptr(), 1.0
    a.c_out, a.h_out, a.gates_out, a.M, a.H, a.fast = c1.data_ptr(), h1.data_ptr(), gt.data_ptr(), M, H, 1
    sp = lambda: lib.las_lstm_cell_rows_args(ctypes.byref(a), _hip.stream())
    print("M = %4d: LM layer 2 %.1f us, LM layer 1 %.1f us, Speller cell %.1f us" % (M, timeit(lm2), timeit(lm1), timeit(sp)))
    if os.environ.get("STAMP") == "1" and M >= 384:
        # phase stamps of the 128-row body (a -DLB_STAMP=1 build: gates_out is the stamp buffer): medians over the workgroups, us from entry
        import numpy as np
        nwg = (H // 16) * ((M + 127) // 128)
        for name, which in (("Speller cell (K = 1152 bf16 rows)", "sp"), ("LM layer 2 (K = 512 + 512 fp32 rows)", "lm2")):
            stamps = torch.zeros(nwg, 16, dtype=torch.int64, device=dev)
            if which == "sp":
                a.gates_out = stamps.data_ptr()
                for _ in range(3): sp()
            else:
                b = _hip.LstmCellArgs()
                b.x, b.x_bf16, b.ldx, b.I = x.data_ptr(), 0, I, I
                b.h, b.ldh, b.Wx, b.Wh, b.bias, b.c_prev, b.fb = h.data_ptr(), H, ih.data_ptr(), hh.data_ptr(), bias.data_ptr(), c.data_ptr(), 0.0
                b.c_out, b.h_out, b.gates_out, b.M, b.H, b.fast = c1.data_ptr(), h1.data_ptr(), stamps.data_ptr(), M, H, 0
                for _ in range(3): lib.las_lstm_cell_rows_args(ctypes.byref(b), _hip.stream())
            torch.cuda.synchronize()
            st = stamps.cpu().numpy().astype(np.float64)
            t0 = st[:, 0].min()
            used = [k for k in range(14) if st[:, k].max() > 0]
            print("  %s: workgroup entry spread %.2f us; stamp: median (min .. max) us after the FIRST workgroup's entry" % (name, (st[:, 0].max() - t0) / 100))
            labels = {0: "entry", 1: "first loads issued", 12: "gates exchanged", 13: "done"}
            for k in used:
                lab = labels.get(k, "chunk %d multiplied" % (k - 2))
                v = (st[:, k] - t0) / 100
                print("    %-22s %6.2f (%5.2f .. %5.2f)   in-workgroup since entry: %5.2f" % (lab, np.median(v), v.min(), v.max(), np.median((st[:, k] - st[:, 0]) / 100)))
