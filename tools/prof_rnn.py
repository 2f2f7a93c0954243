"""Phase-stamp breakdown of one dependent step of the recurrent sweeps (development build: `make -C csrc prof` ->
lib/liblas_hip_prof.so, s_memtime stamps taken by lane 0 of workgroup 0 for steps 200..207 of a B=48, T=1274, H=256 sweep).
    python tools/prof_rnn.py > profiles/r2_phase_stamps.txt
Units: shader cycles (s_memtime); the stamps themselves cost ~11 % (MI355X_MICROARCH.md), read the split, not the total."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip
_hip.LIB_PATH = os.path.join(ROOT, "automatic-speech-recognition_amd", "lib", "liblas_hip_prof.so")
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
os.environ["LAS_DBG_PTR"] = hex(dbg.data_ptr())

PREC = int(os.environ.get("PROF_PREC", "1"))        # 0: the parity mode's exact-fp32 cluster kernels (rnn_seq_f32.hip)
FWD = ["x-ring read + 32 MFMAs (h_{t-1}.W_hh slice)", "gate math + publish h granule", "gather partners' granules (poll)", "LDS barrier"]
BWD = ["gate backward (28 values) + dG tile to LDS", "operand prefetch issue (14 loads)", "LDS barrier", "32 MFMAs (K-split partial dh)",
       "send 3 partial tiles (granules)", "receive 3 partial tiles (poll) + sum", "dZ stores (8)"]

if PREC == 0:
    FWD = ["poll this wave's K quarter of h (granules -> registers)", "64 MFMAs + partial tiles to LDS", "LDS barrier",
           "sum of K-quarters + gate math (accurate)", "publish the member's h"]
    BWD = ["gate backward + dz stores + dz tile to LDS", "LDS barrier", "dz tile read + 64 MFMAs", "send 15 partial tiles (granules)",
           "receive 15 partial tiles (poll) + sum", "LDS barrier"]
for cell, name in ((1, "lstm"), (0, "rnn")):
    G = 4 if cell else 1
    B, T, H = 48, 1274, 256
    g = torch.Generator().manual_seed(0)
    io = _hip.rnn_seq_io_dtype(cell, PREC, H)
    xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).cuda().to(io)
    w0 = (torch.randn(H, G * H, generator=g) * 0.05).cuda(); w1 = w0.clone()
    out = torch.zeros(B, T, 2 * H, device="cuda", dtype=io)
    cst = torch.zeros(B, T, 2, H, device="cuda", dtype=io) if cell else None
    dout = (torch.randn(B, T, 2 * H, generator=g) * 1e-2).cuda().to(io)
    for which, labels in (("fwd", FWD), ("bwd", BWD)):
        for _ in range(2):
            dbg.zero_()
            gates = xp.clone()
            if which == "fwd":
                _hip.rnn_seq_fwd(cell, PREC, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
            else:
                os.environ.pop("LAS_DBG_PTR")                      # stamps only for the sweep under test
                _hip.rnn_seq_fwd(cell, PREC, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
                os.environ["LAS_DBG_PTR"] = hex(dbg.data_ptr())
                dbg.zero_()
                _hip.rnn_seq_bwd(cell, PREC, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H)
            torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        cyc, wall = d[2] - d[0], d[3] - d[1]
        if cyc <= 0 or wall <= 0:
            print("%s %s: no stamps (kernel variant without instrumentation)" % (name, which))
            continue
        ghz = cyc / (wall * 10.0)
        print("%s %s sweep, B=%d T=%d H=%d: %d shader cycles, %.3f ms, clock %.2f GHz -> %.0f cycles = %.2f us per dependent step" % (
            name, which, B, T, H, cyc, wall / 1e5, ghz, cyc / T, cyc / T / ghz / 1e3))
        n = len(labels)
        acc = [0.0] * (n + 1)
        cnt = 0
        for s in range(7):
            st = d[8 + s * 8: 8 + s * 8 + n + 1]
            nxt = d[8 + (s + 1) * 8]
            if min(st) <= 0 or nxt <= 0:
                continue
            for k in range(n):
                acc[k] += st[k + 1] - st[k]
            acc[n] += nxt - st[n]
            cnt += 1
        if cnt:
            tot = sum(acc) / cnt
            for k in range(n):
                print("    %-52s %6.0f cycles  %5.1f %%" % (labels[k], acc[k] / cnt, 100 * acc[k] / cnt / tot))
            print("    %-52s %6.0f cycles  %5.1f %%   (step total %.0f)" % ("loop back / stores / pointer advance", acc[n] / cnt, 100 * acc[n] / cnt / tot, tot))
