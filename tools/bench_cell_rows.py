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
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M in (256, 1024):
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
