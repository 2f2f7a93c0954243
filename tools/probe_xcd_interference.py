"""Does a BPTT sweep slow down next to L2-streaming workgroups because they share ITS XCD's L2, or because they share the device?
The stamped workgroup of the development build (liblas_hip_prof.so) is cluster 0 = XCD 0.  A streaming kernel (tools/micro/streamer.hip)
runs beside the sweep on another stream, confined to a set of XCDs: none / all eight / 1-7 (not the stamped one) / 4-7 / only XCD 0.
    python tools/probe_xcd_interference.py > profiles/r4_xcd_interference.txt"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip
_hip.LIB_PATH = os.path.join(ROOT, "automatic-speech-recognition_amd", "lib", "liblas_hip_prof.so")
S = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libstreamer.so"))
S.streamer_launch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
buf = torch.empty(512 << 20, dtype=torch.uint8, device="cuda").random_(0, 255)
sink = torch.zeros(4096, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
cell, G, B, T, H = 1, 4, 48, 1274, 256
g = torch.Generator().manual_seed(0)
io = _hip.rnn_seq_io_dtype(cell, 1, H)
xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).cuda().to(io)
w0 = (torch.randn(H, G * H, generator=g) * 0.05).cuda(); w1 = w0.clone()
out = torch.zeros(B, T, 2 * H, device="cuda", dtype=io)
cst = torch.zeros(B, T, 2, H, device="cuda", dtype=io)
dout = (torch.randn(B, T, 2 * H, generator=g) * 1e-2).cuda().to(io)


def run(which, mask, iters):
    res = []
    for rep in range(3):
        gates = xp.clone()
        if which == "bwd":
            os.environ.pop("LAS_DBG_PTR", None)
            _hip.rnn_seq_fwd(cell, 1, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
        os.environ["LAS_DBG_PTR"] = hex(dbg.data_ptr())
        dbg.zero_()
        torch.cuda.synchronize()
        e0, e1, s0, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        if mask:
            with torch.cuda.stream(side):
                s0.record()
                assert S.streamer_launch(buf.data_ptr(), buf.numel(), mask, iters, sink.data_ptr(), ctypes.c_void_p(side.cuda_stream)) == 0
                s1.record()
        e0.record()
        if which == "fwd":
            _hip.rnn_seq_fwd(cell, 1, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
        else:
            _hip.rnn_seq_bwd(cell, 1, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H)
        e1.record()
        torch.cuda.synchronize()
        d = dbg.cpu().tolist()
        cyc, wall = d[2] - d[0], d[3] - d[1]
        n = len(PH[which])
        acc, cnt = [0.0] * (n + 1), 0
        for s_ in range(7):                              # steps 200 .. 206: the phases of the stamped workgroup (tools/prof_rnn.py)
            st = d[8 + s_ * 8: 8 + s_ * 8 + n + 1]
            nxt = d[8 + (s_ + 1) * 8]
            if min(st) <= 0 or nxt <= 0:
                continue
            for k in range(n):
                acc[k] += st[k + 1] - st[k]
            acc[n] += nxt - st[n]
            cnt += 1
        res.append((e0.elapsed_time(e1), s0.elapsed_time(s1) if mask else 0.0, cyc / T if cyc > 0 else 0.0, cyc / (wall * 10.0) if wall > 0 else 0.0,
                    [x / max(cnt, 1) for x in acc]))
    res.sort(key=lambda r: r[0])
    return res[1]


PH = {"fwd": ["x-ring read + MFMAs", "gate math + publish h", "gather partners' granules (poll)", "LDS barrier"],
      "bwd": ["gate backward + dG tile", "operand prefetch issue", "LDS barrier", "MFMAs (K-split partial dh)", "send partial tiles", "receive partial tiles (poll) + sum", "dZ stores"]}


for which in ("fwd", "bwd"):
    print("%s sweep, B=%d T=%d H=%d (lstm, bf16); workgroup 0 = cluster 0 = XCD 0 carries the stamps" % (which, B, T, H))
    for name, mask in (("no streamer", 0), ("streamer on all 8 XCDs", 0xFF), ("streamer on XCDs 1-7 (not the stamped cluster's)", 0xFE),
                       ("streamer on XCDs 4-7", 0xF0), ("streamer on XCD 0 only", 0x01)):
        n = bin(mask).count("1")
        ms, sms, cps, ghz, ph = run(which, mask, 12 if n else 0)
        print("  %-52s sweep %.3f ms   stamped cluster %.0f cycles per dependent step (%.2f GHz)   streamer %.2f ms (%s)" % (
            name, ms, cps, ghz, sms, ("%.0f GB/s" % (buf.numel() * 12 * n / 8 / sms / 1e6)) if n else "-"))
        print("      steps 200-206 (cycles): " + "; ".join("%s %.0f" % (l, v) for l, v in zip(PH[which] + ["loop back / stores"], ph)))
