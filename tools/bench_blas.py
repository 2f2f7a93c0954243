#!/usr/bin/env python
"""hipBLASLt / rocBLAS (through torch.matmul) on the step's GEMM shapes, next to las_gemm / las_gemm_kk: what a library GEMM
would give for the plain products (weight gradients, x-projection, dX).   python tools/bench_blas.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip

dev = torch.device("cuda", 0)
bf = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, T, H = 48, 1274, 256
GH, Ik = 4 * H, 2 * H
R = B * T
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(R, Ik, generator=g).to(dev).to(bf)
dz = torch.randn(R, 2 * GH, generator=g).to(dev).to(bf)
w_ihT = torch.randn(2 * GH, Ik, generator=g).to(dev).to(bf)          # [N, K]
w_ih = torch.randn(Ik, 2 * GH, generator=g).to(dev).to(bf)           # [N=Ik, K=2GH]
dW = torch.zeros(Ik, GH, device=dev)
gates = torch.empty(R, 2 * GH, device=dev, dtype=bf)
dx = torch.empty(R, Ik, device=dev, dtype=bf)


def flops(m, n, k):
    return 2.0 * m * n * k


rows = []
# weight gradient dW_ih (one direction): x^T . dZ[:, :GH]
t_mine = timeit(lambda: _hip.gemm(_hip.PREC_BF16, x, dz, dW, True, False, Ik, GH, R, Ik, 2 * GH, GH, beta=1.0))
dzd = dz[:, :GH]
t_lib = timeit(lambda: torch.mm(x.t(), dzd))
t_lib2 = timeit(lambda: torch.mm(x.t(), dz))                          # both directions at once
rows.append(("dW_ih TN  M=%d N=%d K=%d" % (Ik, GH, R), t_mine, t_lib, flops(Ik, GH, R)))
rows.append(("dW_ih TN both dirs N=%d" % (2 * GH), 2 * t_mine, t_lib2, flops(Ik, 2 * GH, R)))
# x-projection: x . W^T + b   (las_gemm_kk: both operands K-contiguous)
t_mine = timeit(lambda: _hip.gemm_kk(x, w_ihT, gates, R, 2 * GH, Ik, Ik, Ik, 2 * GH))
t_lib = timeit(lambda: torch.mm(x, w_ihT.t(), out=gates))
rows.append(("x-proj NT M=%d N=%d K=%d" % (R, 2 * GH, Ik), t_mine, t_lib, flops(R, 2 * GH, Ik)))
# dX: dZ . W_ih^T  (A [R, 2GH], B [Ik, 2GH] K-contiguous)
t_mine = timeit(lambda: _hip.gemm_kk(dz, w_ih, dx, R, Ik, 2 * GH, 2 * GH, 2 * GH, Ik))
t_lib = timeit(lambda: torch.mm(dz, w_ih.t(), out=dx))
rows.append(("dX NT     M=%d N=%d K=%d" % (R, Ik, 2 * GH), t_mine, t_lib, flops(R, Ik, 2 * GH)))
# dW_hh: batched over utterances  h^T . dZ   [B, H, T] x [B, T, GH]
h = torch.randn(B, T, 2 * H, generator=g).to(dev).to(bf)
dz3 = dz.view(B, T, 2 * GH)
part = torch.empty(B, H, GH, device=dev)
t_mine = timeit(lambda: _hip.gemm(_hip.PREC_BF16, h, dz3, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                                  strideA=T * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=0, b_off=2 * GH))
hh = h[:, :-1, :H]
dd = dz3[:, 1:, :GH]
t_lib = timeit(lambda: torch.bmm(hh.transpose(1, 2), dd))
rows.append(("dW_hh TN batch %d M=%d N=%d K=%d" % (B, H, GH, T - 1), t_mine, t_lib, B * flops(H, GH, T - 1)))
hf = h.view(R, 2 * H)[:-1, :H]
df = dz[1:, :GH]
t_lib = timeit(lambda: torch.mm(hf.t(), df))
rows.append(("dW_hh as ONE TN product K=%d" % (R - 1), t_mine, t_lib, flops(H, GH, R - 1)))
for big in (0, 1):
    _hip.lib().las_dev_gemm_kk_big(big)
    t1 = timeit(lambda: _hip.gemm_kk(x, w_ihT, gates, R, 2 * GH, Ik, Ik, Ik, 2 * GH))
    t2 = timeit(lambda: _hip.gemm_kk(dz, w_ih, dx, R, Ik, 2 * GH, 2 * GH, 2 * GH, Ik))
    print("las_gemm_kk %s tiles: x-projection %.1f us, dX %.1f us" % ("256 x 256" if big else "128 x 128", t1, t2))
print("%-44s %10s %10s %12s %12s" % ("product", "ours us", "library us", "ours TF/s", "library TF/s"))
for name, a, b, f in rows:
    print("%-44s %10.1f %10.1f %12.0f %12.0f" % (name, a, b, f / a / 1e6, f / b / 1e6))
