"""Micro-benchmark of las_gemm at the shapes the train step uses."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip
def run(name, M, N, K, tA, tB, reps=5):
    A = torch.randn((K, M) if tA else (M, K), device="cuda"); B = torch.randn((N, K) if tB else (K, N), device="cuda")
    C = torch.empty(M, N, device="cuda")
    f = lambda: _hip.gemm(1, A, B, C, bool(tA), bool(tB), M, N, K, A.shape[1], B.shape[1], N)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s M=%6d N=%5d K=%6d %s%s : %8.3f ms  %7.1f TF" % (name, M, N, K, "T" if tA else "N", "T" if tB else "N", ms, 2.0 * M * N * K / ms / 1e9), flush=True)
BT = 48 * 1274
run("xproj l1 (NN)", BT, 1024, 512, 0, 0)
run("xproj l0 (NN K=39)", BT, 1024, 39, 0, 0)
run("dense l0 (NN)", BT, 512, 512, 0, 0)
run("dense pyr (NN)", BT // 2, 512, 1024, 0, 0)
run("dx (NT)", BT, 512, 1024, 0, 1)
run("dW_ih (TN)", 512, 1024, BT, 1, 0)
run("dW dense (TN)", 1024, 512, BT // 2, 1, 0)
run("dcellW (TN)", 1152, 2048, 48 * 191, 1, 0)
run("square 4096", 4096, 4096, 4096, 0, 0)
