#!/usr/bin/env python
"""The train step of the bench geometry once scheduled sampling is ACTIVE (the bench itself runs at the start of the schedule, where every
coin says "teacher": las/las.py:177-183): the global step is set into the middle of the decay, so that ~half of the decode steps draw their
token on the device (in-loop logits + Gumbel arg-max in the forward loop).  ms per step at the schedule's start and in its middle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import bench
from helpers import synthetic_batch
from las import layers as L, variables as V, _hip
from las.las import LAS, Listener, Speller
dev = torch.device("cuda", 0)
L.set_cell("lstm"); L.set_precision("bf16")
st = V.reset_default_store(device=dev, seed=0)
args = bench.bench_args("lstm")
las = LAS(args, Listener, Speller, {})
las.build_variables()
xs, ys = synthetic_batch(48, 1274, 256, args.vocab_size, seed=0, min_frac=0.834)
xs = (torch.tensor(xs[0], device=dev), xs[1]); ys = (torch.tensor(ys[0], device=dev), ys[1])
for name, g0 in (("start of the schedule (teacher forcing)", 0), ("middle of the decay", (args.warmup_step + args.max_step) // 2)):
    st.global_step = g0
    for _ in range(4):
        las.train(xs, ys)
    torch.cuda.synchronize()
    st.global_step = g0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        las.train(xs, ys)
    e1.record(); torch.cuda.synchronize()
    print("%-42s teacher-forcing rate %.3f: %.3f ms per step" % (name, las.speller._scheduled_sampling() if hasattr(las, "speller") else float("nan"), e0.elapsed_time(e1) / 30))
