"""Hashes of one optimiser step (updated parameters + both Adam moments) on five wide-Speller-path geometries (two layers, K = 201 / C = 10
location-aware attention, both cells, both modes): run it under two builds of the library (LAS_LIB_PATH) to state that a kernel change is
bit-identical -- how the round-6 re-arrangements of csrc/speller_wide.h were checked against the commit before them."""
import os, sys, hashlib, warnings
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)
import torch, numpy as np
warnings.simplefilter("ignore")
from helpers import make_args, synthetic_batch
from las import _hip, layers as L, variables as V
from las.las import LAS, Listener, Speller
from oracle import las_oracle as O
for (cell, prec, T, B, mode, nl) in (("rnn", "bf16", 1274, 6, "loc", 2), ("lstm", "bf16", 1274, 6, "loc", 2), ("rnn", "f32", 640, 5, "loc", 2),
                                     ("lstm", "bf16", 400, 3, "loc", 1), ("lstm", "f32", 333, 4, "loc", 2)):
    args = make_args(enc_units=64, num_enc_layers=2, dec_units=128, num_dec_layers=nl, embedding_size=64, attention_size=64, mode=mode,
                     loc_kernel_size=201, loc_num_channels=10, lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=40,
                     )
    xs, ys = synthetic_batch(B, T, 40, 40, seed=11, min_frac=0.7)
    p0 = O.init_params(args, seed=2, cell=cell)
    L.set_cell(cell); L.set_precision(prec)
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, {})
    out = las.train(xs, ys)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (st.flat, st.adam_m, st.adam_v):
        h.update(t.detach().cpu().numpy().tobytes())
    print(cell, prec, T, B, nl, "wide" in _hip.speller_last_variant()["fwd"], h.hexdigest()[:16], float(st.adam_m.abs().sum()))
