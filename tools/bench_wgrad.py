"""Weight-gradient (TN) contractions of the Listener at the bench shape: time and check against torch.
    python tools/bench_wgrad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip

dev = "cuda"
B, T, H = 48, 1274, 256
GH = 4 * H
g = torch.Generator().manual_seed(0)
dZ = (torch.randn(B, T, 2, GH, generator=g) * 0.1).to(dev).to(torch.bfloat16)
out = (torch.randn(B, T, 2 * H, generator=g) * 0.5).to(dev).to(torch.bfloat16)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for I in (64, 512, 1024):
    X = (torch.randn(B * T, I, generator=g) * 0.5).to(dev).to(torch.bfloat16)
    for d in range(2):
        C = torch.zeros(I, GH, device=dev)
        fn = lambda: _hip.gemm(_hip.PREC_BF16, X, dZ, C, True, False, I, GH, B * T, I, 2 * GH, GH, beta=0.0, b_off=d * GH)
        us = timeit(fn)
        ref = X.float().t() @ dZ.view(B * T, 2, GH)[:, d].float()
        err = (C - ref).abs().max().item() / ref.abs().max().item()
        print("dW_ih I=%4d dir %d: %7.1f us  %6.0f TFLOP/s   rel err %.2e" % (I, d, us, 2.0 * I * GH * B * T / us / 1e6, err), flush=True)
for d in range(2):
    part = torch.empty(B, H, GH, device=dev)
    a_off = d * H + (0 if d == 0 else 2 * H)
    b_off = d * GH + (2 * GH if d == 0 else 0)
    fn = lambda: _hip.gemm(_hip.PREC_BF16, out, dZ, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                           strideA=T * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=a_off, b_off=b_off)
    us = timeit(fn)
    o = out.view(B, T, 2, H)[:, :, d].float()
    z = dZ[:, :, d].float()
    ref = torch.einsum("bth,btg->bhg", o[:, :-1], z[:, 1:]) if d == 0 else torch.einsum("bth,btg->bhg", o[:, 1:], z[:, :-1])
    err = (part - ref).abs().max().item() / ref.abs().max().item()
    print("dW_hh batched dir %d: %7.1f us  %6.0f TFLOP/s   rel err %.2e" % (d, us, 2.0 * H * GH * B * (T - 1) / us / 1e6, err), flush=True)
