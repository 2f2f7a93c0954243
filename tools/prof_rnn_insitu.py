"""Phase-stamp split of one dependent step of the LAST recurrent sweep of a train step (the bottom layer's BPTT, T = 1274) IN SITU: the
bench's B = 48 / T = 1274 step with the development build (lib/liblas_hip_prof.so, `make -C csrc prof`), i.e. with the weight-gradient
GEMMs of the layer above running beside the sweep.  Compare with tools/prof_rnn.py (the same sweep alone).
    python tools/prof_rnn_insitu.py [LAS_NO_SIDE=1 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from las import _hip
_hip.LIB_PATH = os.path.join(ROOT, "automatic-speech-recognition_amd", "lib", "liblas_hip_prof.so")
dbg = torch.zeros(128, dtype=torch.int64, device="cuda")
os.environ["LAS_DBG_PTR"] = hex(dbg.data_ptr())
import bench
from helpers import synthetic_batch
from las import layers as L, variables as V
from las.las import LAS, Listener, Speller
dev = torch.device("cuda", 0)
L.set_cell("lstm"); L.set_precision("bf16")
V.reset_default_store(device=dev, seed=0)
args = bench.bench_args("lstm", 1)
las = LAS(args, Listener, Speller, {})
las.build_variables()
xs, ys = synthetic_batch(48, 1274, 256, args.vocab_size, seed=0, min_frac=0.834)
xs = (torch.tensor(xs[0], device=dev), xs[1]); ys = (torch.tensor(ys[0], device=dev), ys[1])
for _ in range(4):
    dbg.zero_()
    las.train(xs, ys)
    torch.cuda.synchronize()
BWD = ["gate backward (28 values) + dG tile to LDS", "operand prefetch issue (14 loads)", "LDS barrier", "32 MFMAs (K-split partial dh)",
       "send 3 partial tiles (granules)", "receive 3 partial tiles (poll) + sum", "dZ stores (8)"]
d = dbg.cpu().tolist()
cyc, wall = d[2] - d[0], d[3] - d[1]
ghz = cyc / (wall * 10.0)
T = 1274
print("last sweep of the step (bottom layer BPTT) in situ: %d shader cycles, %.3f ms, clock %.2f GHz -> %.0f cycles = %.2f us per dependent step" % (
    cyc, wall / 1e5, ghz, cyc / T, cyc / T / ghz / 1e3))
n = len(BWD)
acc = [0.0] * (n + 1); cnt = 0
for s in range(7):
    st = d[8 + s * 8: 8 + s * 8 + n + 1]; nxt = d[8 + (s + 1) * 8]
    if min(st) <= 0 or nxt <= 0:
        continue
    for k in range(n):
        acc[k] += st[k + 1] - st[k]
    acc[n] += nxt - st[n]; cnt += 1
tot = sum(acc) / max(cnt, 1)
for k, lab in enumerate(BWD + ["loop back / stores / pointer advance"]):
    print("    %-52s %6.0f cycles  %5.1f %%" % (lab, acc[k] / max(cnt, 1), 100 * acc[k] / max(cnt, 1) / tot))
print("    (step total %.0f; stamps of steps 200..207 of workgroup 0)" % tot)
