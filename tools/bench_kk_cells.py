import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from las import _hip
lib = _hip.lib()
def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (256, 512, 1024):
    for K in (512, 1024, 1152):
        N = 2048
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda")
        bias = torch.zeros(N, device="cuda")
        t = timeit(lambda: _hip.gemm_kk(A, B, C, M, N, K, K, K, N, bias=bias))
        print("gemm_kk M=%d K=%d: %.1f us (%.0f TF/s)" % (M, K, t, 2 * M * N * K / t / 1e6))
    z = torch.randn(M, 2048, device="cuda"); c = torch.randn(M, 512, device="cuda")
    c1, h1 = torch.empty(M, 512, device="cuda"), torch.empty(M, 512, device="cuda")
    t = timeit(lambda: lib.las_lstm_pointwise(_hip.p(z), _hip.p(c), M, 512, 0.0, _hip.p(c1), _hip.p(h1), _hip.stream()))
    print("pointwise M=%d: %.1f us" % (M, t))
    x = torch.randn(M, 1024, device="cuda")
    t = timeit(lambda: x.to(torch.bfloat16))
    print("torch fp32->bf16 [M,1024]: %.1f us" % t)
