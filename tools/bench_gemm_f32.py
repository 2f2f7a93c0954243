"""Micro-benchmark of the PARITY mode's contractions (las_gemm, LAS_PREC_F32: exact-fp32 MFMA kernels) at the bench shapes
(B = 48, T = 1274): TF/s against the 157 TF/s fp32 matrix peak, and the Speller's per-step skinny products in us."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


BT = 48 * 1274
P = _hip.PREC_F32
rows = [("x-proj L0  NN", BT, 2048, 39, 0, 0, 1), ("x-proj L0 (K padded to 40, as the step runs it)", BT, 2048, 40, 0, 0, 1), ("x-proj L1  NN", BT, 2048, 512, 0, 0, 1), ("dense L0   NN", BT, 512, 512, 0, 0, 1),
        ("dense L1   NN", BT // 2, 512, 1024, 0, 0, 1), ("dX x-proj  NT", BT, 512, 2048, 0, 1, 1), ("dX dense   NT", BT, 512, 512, 0, 1, 1),
        ("dW_ih      TN", 512, 1024, BT, 1, 0, 1), ("dW dense   TN", 512, 512, BT, 1, 0, 1), ("dW_hh batched TN", 256, 1024, 1273, 1, 0, 48),
        ("cell step  NN", 48, 2048, 1152, 0, 0, 1), ("cell step  NT", 48, 1152, 2048, 0, 1, 1), ("dcellW     TN", 1152, 2048, 48 * 191, 1, 0, 1),
        ("vocab      NN", 48 * 191, 30, 512, 0, 0, 1)]
for name, M, N, K, tA, tB, bt in rows:
    A = torch.randn((bt, K, M) if tA else (bt, M, K), device="cuda") * 0.5
    B = torch.randn((bt, N, K) if tB else (bt, K, N), device="cuda") * 0.05
    C = torch.empty(bt, M, N, device="cuda")
    f = lambda: _hip.gemm(P, A, B, C, bool(tA), bool(tB), M, N, K, A.shape[2], B.shape[2], N, batch=bt, strideA=A.stride(0) if bt > 1 else 0,
                          strideB=B.stride(0) if bt > 1 else 0, strideC=M * N if bt > 1 else 0)
    t = timeit(f)
    fl = 2.0 * M * N * K * bt
    print("%-18s M=%6d N=%5d K=%6d x%2d | %9.1f us %6.1f TF/s (%4.1f %% of 157)" % (name, M, N, K, bt, t * 1e3, fl / t / 1e9, fl / t / 1e9 / 1.573), flush=True)
