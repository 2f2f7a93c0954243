"""The `stacked` legs of bench.py alone (k bucket batches in one step), for A/B runs of the scheduling knobs."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for k in [int(x) for x in (sys.argv[1:] or ["2", "4"])]:
    r = bench.side_step_bench(dev, "lstm", "bf16", 1, 48, 1274, stack=k)
    print("k=%d  %.2f ms  %.0f utt/s  %s" % (k, r["ms_per_step"], r["value"], r["schedule"]), flush=True)
