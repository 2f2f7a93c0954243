#!/usr/bin/env python
"""The SAME recurrent sweep (forward, then BPTT) repeated N times on identical inputs: are the outputs -- h, the saved activations, dZ, the bias
gradients -- bit-identical every time?  (Round 6: looking for the source of the ~2e-6 run-to-run difference tools/probe_determinism.py found in
steps with 64-unit listener layers.)   H=64 T=96 B=8 N=200 [FLAGS=<las seq flags>] python tools/probe_sweep_repeat.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
from las import _hip
H, T, B, N = (int(os.environ.get(k, d)) for k, d in (("H", "64"), ("T", "96"), ("B", "8"), ("N", "200")))
flags = int(os.environ.get("FLAGS", "0"))
cell, G = _hip.CELL_LSTM, 4
GH = G * H
g = torch.Generator().manual_seed(1)
bf = torch.bfloat16
xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).to(bf).cuda()
whh = [((torch.rand(H, GH, generator=g) * 2 - 1) * 0.3).cuda() for _ in range(2)]
dout = (torch.randn(B, T, 2 * H, generator=g) * 0.1).to(bf).cuda()
noise = torch.randn(64 << 20, device="cuda")          # something else to stir the caches between repetitions


side = torch.cuda.Stream()
CONC = os.environ.get("CONC", "0") == "1"


def once(stir):
    gates = xp.clone()
    if CONC and stir:                         # memory traffic on ANOTHER stream while the sweeps run (in a train step: chunk products, prepare, weight gradients)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                noise.mul_(1.0001)
    out = torch.empty(B, T, 2 * H, device="cuda", dtype=bf)
    cst = torch.empty(B, T, 2, H, device="cuda", dtype=bf)
    if stir:
        noise.mul_(1.0001)
    _hip.rnn_seq_fwd(cell, _hip.PREC_BF16, B, T, H, gates, whh[0], whh[1], GH, out, 2 * H, T * 2 * H, cst, 1.0, flags=flags)
    saved = (gates.clone(), out.clone(), cst.clone())
    dbf, dbb = torch.zeros(GH, device="cuda"), torch.zeros(GH, device="cuda")
    if stir:
        noise.mul_(0.9999)
    _hip.rnn_seq_bwd(cell, _hip.PREC_BF16, B, T, H, gates, whh[0], whh[1], GH, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H, 1.0,
                     db_fw=dbf, db_bw=dbb, flags=flags)
    torch.cuda.synchronize()
    _hip.check_status()
    if os.environ.get("ROWS") and not hasattr(once, "said"):
        once.said = True
    return saved + (gates.clone(), dbf, dbb)


names = ("activated gates", "h", "c", "dZ", "db_fw", "db_bw")
ref = once(False)
bad = {n: 0 for n in names}
for i in range(N):
    cur = once(i % 2 == 1)
    for n, a, b in zip(names, ref, cur):
        if not torch.equal(a, b):
            bad[n] += 1
            if bad[n] == 1:
                d = (a.float() - b.float()).abs()
                idx = torch.nonzero(d.reshape(-1) > 0).reshape(-1)
                rows = sorted(set((idx // (a.numel() // a.shape[0])).tolist())) if a.dim() > 1 else []
                print("first mismatch in %s at repetition %d: %d elements differ, max %.3e, batch rows %s, shape %s"
                      % (n, i, idx.numel(), d.max().item(), rows, tuple(a.shape)), flush=True)
print("H=%d T=%d B=%d flags=%d: repetitions with a mismatch, of %d: %s" % (H, T, B, flags, N, bad))
