#!/usr/bin/env python
"""Time the Speller decode loop (forward + gradient) alone at the bench configuration.
    python tools/bench_speller.py [--mode add|loc] [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="add")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--B", type=int, default=48)
ap.add_argument("--Tp", type=int, default=160)
ap.add_argument("--U", type=int, default=191)
ap.add_argument("--cell", default="lstm")
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()

from helpers import make_args
from las import _hip, layers as L, variables as V
from las.las import Speller

dev = torch.device("cuda", 0)
L.set_cell(a.cell)
L.set_precision(a.dtype)
V.reset_default_store(device=dev, seed=0)
args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
                 mode=a.mode, vocab_size=30, enc_type="pblstm", loc_kernel_size=201, loc_num_channels=10)
sp = Speller(args)
rng = np.random.RandomState(0)
enc = torch.tensor(rng.randn(a.B, a.Tp, 512).astype(np.float32) * 0.5, device=dev, requires_grad=True)
enc_len = rng.randint(int(a.Tp * 0.8), a.Tp + 1, size=a.B)
y = torch.tensor(rng.randint(3, 30, size=(a.B, a.U)), device=dev)


def step():
    logits, _, _ = sp(enc, enc_len, a.U, teacher=y, is_training=True)
    (logits * 1e-3).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
_hip.prof_begin()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
prof = _hip.prof_end()
for k, v in sorted(prof.items()):
    print("%-24s %8.3f ms  (%d calls)  %.2f us/step" % (k, sum(v) / len(v), len(v), sum(v) / len(v) * 1e3 / a.U))
