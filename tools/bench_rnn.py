"""Micro-benchmark of the recurrent sweep (K2/K2b) at the bench shape. Prints us/step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip

def run(cell, B=48, T=1274, H=256, prec=1, reps=3):
    G = 4 if cell else 1
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    io = _hip.rnn_seq_io_dtype(cell, prec, H)
    xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).to(dev).to(io)
    w0 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
    w1 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
    out = torch.zeros(B, T, 2 * H, device=dev, dtype=io)
    cst = torch.zeros(B, T, 2, H, device=dev, dtype=io) if cell else None
    dout = torch.randn(B, T, 2 * H, generator=g).to(dev).to(io)
    res = {}
    for name in ("fwd", "bwd"):
        ts = []
        for _ in range(reps):
            gates = xp.clone()
            if name == "bwd":
                _hip.rnn_seq_fwd(cell, prec, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "fwd":
                _hip.rnn_seq_fwd(cell, prec, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
            else:
                _hip.rnn_seq_bwd(cell, prec, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[name] = min(ts)
    print("cell=%s H=%d B=%d T=%d prec=%d : fwd %.2f ms (%.2f us/step)  bwd %.2f ms (%.2f us/step)" % (
        "lstm" if cell else "rnn", H, B, T, prec, res["fwd"], res["fwd"] * 1e3 / T, res["bwd"], res["bwd"] * 1e3 / T), flush=True)

if __name__ == "__main__":
    run(1); run(0)
    if len(sys.argv) > 1:
        run(1, H=128); run(1, prec=0, T=200)
