"""Speller forward + gradient at the bench geometry for B = 48 .. 192 rows: one call, or row chunks of 48 (one-launch loops)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from helpers import make_args
from las import _hip, layers as L, variables as V
from las.las import Speller

L.set_cell("lstm"); L.set_precision("bf16")
V.reset_default_store(device="cuda", seed=3)
args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=256, attention_size=128, mode="add",
                 vocab_size=30, enc_type="pblstm")
sp = Speller(args)

Tp, U = 160, 191
rng = np.random.RandomState(1)
for B in (48, 96, 144, 192):
    enc = torch.tensor(rng.randn(B, Tp, 512).astype(np.float32) * 0.5, device="cuda", requires_grad=True)
    enc_len = rng.randint(Tp * 3 // 4, Tp + 1, size=B)
    y = rng.randint(3, 30, size=(B, U))
    w = torch.tensor(rng.randn(B, U, 30).astype(np.float32)).cuda()

    def step():
        logits, _, _ = sp(enc, enc_len, U, teacher=y, is_training=True)
        (logits * w).sum().backward()
        _hip.run_deferred() if hasattr(_hip, "run_deferred") else None
        _hip.join_side_stream()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    print("B=%d  %.2f ms per forward + gradient  (split rows: %s)" % (B, (time.perf_counter() - t0) / 5 * 1e3, os.environ.get("LAS_SPELLER_ROW_CHUNK", "-")), flush=True)
