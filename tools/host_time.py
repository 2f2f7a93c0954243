#!/usr/bin/env python
"""Host-side enqueue time of one train step (no synchronisation inside the step) vs the GPU step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import bench
from helpers import synthetic_batch
from las import layers as L, variables as V
from las.las import LAS, Listener, Speller
dev = torch.device("cuda", 0)
L.set_cell("lstm"); L.set_precision("bf16")
V.reset_default_store(device=dev, seed=0)
args = bench.bench_args("lstm")
las = LAS(args, Listener, Speller, {})
las.build_variables()
xs, ys = synthetic_batch(48, 1274, 256, args.vocab_size, seed=0, min_frac=0.834)
xs = (torch.tensor(xs[0], device=dev), xs[1]); ys = (torch.tensor(ys[0], device=dev), ys[1])
for _ in range(3):
    las.train(xs, ys)
torch.cuda.synchronize()
import las.las as M
marks = {}
orig_listener = las.listener.__call__
hs = []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    las.train(xs, ys)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0, t2 - t0))
for h, g in hs:
    print("host enqueue %.2f ms   step (host+drain) %.2f ms" % (h * 1e3, g * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(3):
    las.train(xs, ys)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
