"""The loop train.py runs (DeviceFeeder + LAS.train, synthetic source, headline bucket) for a few steps -- for rocprofv3 traces:
   rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/lt -o l -- python3 tools/loop_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import bench
from data import BUCKET_BOUNDARIES, SyntheticBatches
from las import _hip, layers as L, variables as V
from las.input_pipeline import feeder_for
from las.las import LAS, Listener, Speller
dev = torch.device("cuda", 0)
L.set_cell("lstm"); L.set_precision("bf16")
V.reset_default_store(device=dev, seed=0)
las = LAS(bench.bench_args("lstm"), Listener, Speller, {})
las.build_variables()
k = BUCKET_BOUNDARIES.index(1275)
feed = feeder_for(SyntheticBatches(13, 30, seed=0, buckets=[k]), dev, 13)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
for i in range(n):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if os.environ.get("LAS_PHASES"):
            _hip.prof_begin()
    xs, ys = next(feed)
    las.train(xs, ys)
torch.cuda.synchronize()
print("ms per step: %.3f" % ((time.perf_counter() - t0) / (n - 4) * 1e3))
if os.environ.get("LAS_PHASES"):
    for k, v in sorted(_hip.prof_end().items()):
        print("  %-34s %8.3f ms  x%d" % (k, sum(v) / len(v), len(v) // (n - 4)))
feed.close()
