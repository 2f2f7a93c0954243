"""Beam-search decode rate against the number of utterances per device-resident batch (bench.py's decode leg fixes 16 = 256 rows)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
for nutt in (16, 32, 48, 64):
    try:
        r = bench.decode_bench(dev, "lstm", "bf16", nutt=nutt)
        print(nutt, r["value"], r["timing"], r["us_per_decode_step"], r["step_parts_us"], flush=True)
    except Exception as e:
        print(nutt, "failed:", type(e).__name__, str(e)[:200], flush=True)
