import torch, time
dev = "cuda"
BT, GH2 = 48 * 1274, 2048
for I in (64, 512, 1024):
    X = torch.randn(BT, I, device=dev).to(torch.bfloat16)
    dZ = torch.randn(BT, GH2, device=dev).to(torch.bfloat16)
    dZd = dZ[:, :1024]
    for name, fn in (("mm bf16 out", lambda: torch.mm(X.t(), dZd)),
                     ("mm bf16 out both dirs", lambda: torch.mm(X.t(), dZ)),
                     ("mm out_dtype f32", lambda: torch.mm(X.t(), dZd, out_dtype=torch.float32))):
        try:
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): y = fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            n = y.shape[1]
            print("I=%4d %-24s %7.1f us  %6.0f TFLOP/s  out %s" % (I, name, ms * 1e3, 2.0 * I * n * BT / ms / 1e9, y.dtype), flush=True)
        except Exception as e:
            print("I=%4d %-24s failed: %s" % (I, name, str(e)[:100]), flush=True)
