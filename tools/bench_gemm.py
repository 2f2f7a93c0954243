"""Micro-benchmark of the chain contractions at the bench shapes (B=48, T=1274): TF/s of las_gemm_kk (bf16 operands,
LDS-DMA) next to las_gemm (fp32 operands converted in the loader)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip

def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

BT = 48 * 1274
for name, M, N, K in [("x-proj L1 (K=512 -> 2GH=2048)", BT, 2048, 512), ("dense L0 (512 -> 512, tanh)", BT, 512, 512),
                      ("dense L1 pairs (1024 -> 512)", BT // 2, 512, 1024), ("dx of x-proj (2048 -> 512)", BT, 512, 2048),
                      ("x-proj L0 (K=64 pad -> 2048)", BT, 2048, 64), ("x-proj L3 (T=319)", 48 * 319, 2048, 512)]:
    A = (torch.randn(M, K, device="cuda") * 0.5)
    W = (torch.randn(K, N, device="cuda") * 0.05)
    Ab, WTb = A.to(torch.bfloat16), W.t().contiguous().to(torch.bfloat16)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    Cf = torch.empty(M, N, device="cuda")
    t_kk = timeit(lambda: _hip.gemm_kk(Ab, WTb, Cb, M, N, K, K, K, N))
    t_kkf = timeit(lambda: _hip.gemm_kk(Ab, WTb, Cf, M, N, K, K, K, N))
    t_old = timeit(lambda: _hip.gemm(_hip.PREC_BF16, A, W, Cf, False, False, M, N, K, K, N, N))
    fl = 2.0 * M * N * K
    print("%-34s M=%6d N=%5d K=%5d | kk->bf16 %7.1f us %6.0f TF/s | kk->f32 %7.1f us %6.0f TF/s | fp32-operand %7.1f us %6.0f TF/s" % (
        name, M, N, K, t_kk * 1e3, fl / t_kk / 1e9, t_kkf * 1e3, fl / t_kkf / 1e9, t_old * 1e3, fl / t_old / 1e9), flush=True)
