import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip
for (M, N, K) in [(1276, 1024, 512), (640, 1024, 512), (2548, 1024, 512), (5096, 512, 2048), (1276, 512, 1024)]:
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    B = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    ref = (A.float() @ B.float().t())
    for rep in range(4):
        junk = torch.full((64, 1024, 1024), float("nan"), device="cuda")      # poison freed memory
        del junk
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        _hip.gemm_kk(A, B, C, M, N, K, K, K, N)
        torch.cuda.synchronize()
        d = (C.float() - ref)
        bad = ~torch.isfinite(C.float()) | (d.abs() > 0.05)
        nb = int(bad.sum())
        msg = ""
        if nb:
            idx = bad.nonzero()
            msg = " rows %d..%d cols %d..%d first %s" % (idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(), idx[:6].tolist())
        print("M=%d N=%d K=%d rep %d: bad %d%s" % (M, N, K, rep, nb, msg), flush=True)
