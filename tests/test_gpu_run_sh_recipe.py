"""The reference's own training recipe (run.sh:59-76) AT ITS SIZES through the C ABI (VERDICT r5 "next round" item 1a):

    train.py --enc_units 512 --dec_units 1024 --embedding_size 256 --attention_size 128 --num_enc_layers 4 --num_dec_layers 2
             --mode loc  (+ the defaults it leaves alone: --enc_type cnn, num_enc_channels 32, loc K = 201 / C = 10, subword V = 5000)

i.e. CNNLayer (las/layers.py:118-163: two 3x3 stride-2 convolutions, T = 1274 -> T' = 319 frames of 4 x 32 features, four BLSTM-512 +
dense + relu(bn(.))), LocationAwareAttention (las/layers.py:281-311) and a MultiRNNCell of two 1024-unit cells whose concatenated
states are the attention query (las/las.py:185-199).  One train step at B = 4 / T = 1274 / U ~ 190 and one beam search, both cells
(the reference's tanh BasicRNNCell and the LSTM variant), parity mode (f32) and speed mode (bf16), against the oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import grad_errors, make_args, oracle_decode, synthetic_batch, train_step_pair

pytestmark = pytest.mark.gpu

V = 5000


def run_sh_args(**over):
    kw = dict(enc_type="cnn", enc_units=512, num_enc_layers=4, num_enc_channels=32, dec_units=1024, num_dec_layers=2, embedding_size=256,
              attention_size=128, mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=V, unit="subword", lr=1e-4,
              grad_clip=5.0, label_smoothing=True, scheduled_sampling=False, dropout_rate=0.0)
    kw.update(over)
    return make_args(**kw)


def _log(name, rec):
    path = os.environ.get("LAS_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **rec)) + "\n")


# f32: the tolerances of the full-T rows (tests/test_gpu_full_scale.py).  bf16: the oracle's bf16-emulating mode; the four BLSTM-512
# layers sit behind a batch normalisation over 4 x 319 frames each (las/layers.py:161), which renormalises whatever the rounding did.
TOL = {
    "f32": dict(logits=1e-3, alphas=1e-3, loss=1e-4, grad=5e-3),
    "bf16": dict(logits=2e-2, alphas=5e-3, loss=2e-3, grad=3e-2),
}


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("cell", ["rnn", "lstm"])
def test_run_sh_recipe_train_step_matches_oracle(cell, prec):
    args = run_sh_args()
    xs, ys = synthetic_batch(4, 1274, 256, V, seed=21, min_frac=0.834)
    U = int(ys[1].max())
    assert 150 < U <= 200
    r = train_step_pair(args, cell, prec, xs, ys, seed=17, enc_type="cnn")
    assert r["alphas"].shape[-1] == 319 and r["logits"].shape[-1] == V
    errs = dict(logits=(r["logits"] - r["logits_o"]).abs().max().item(), alphas=(r["alphas"] - r["alphas_o"]).abs().max().item(),
                loss=abs(r["loss"] - r["loss_o"]) / max(1.0, abs(r["loss_o"])))
    ge = grad_errors(r)
    worst = max(ge, key=ge.get)
    agree = (r["logits"].argmax(-1) == r["logits_o"].argmax(-1)).float().mean().item()
    _log("run_sh_train_step", dict(prec=prec, cell=cell, B=4, T=1274, Tp=319, U=U, worst_grad=worst, worst_grad_err=ge[worst],
                                   token_agreement=agree, **errs))
    print("run.sh %s/%s: logits %.2e alphas %.2e loss %.2e worst grad %s %.2e agree %.4f" % (
        cell, prec, errs["logits"], errs["alphas"], errs["loss"], worst, ge[worst], agree))
    tol = TOL[prec]
    for k, v in errs.items():
        assert v < tol[k], (k, v)
    for n, e in ge.items():
        assert e < tol["grad"], (n, e)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("cell", ["rnn", "lstm"])
def test_run_sh_recipe_beam_search_matches_oracle(cell, prec):
    """decode.py's search at the recipe's sizes: one T = 1274 utterance (T' = 319), beam 4, 24 steps (convert_rate cut so that the
    oracle's 2 x 1024 cells x 4 rows stay seconds) -- state packing over TWO layers (las/beam_search.py:211-224)."""
    from las import layers as L, variables as V_
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from oracle import las_oracle as O
    args = run_sh_args(beam_size=4, convert_rate=24.5 / 1274, apply_lm=False)
    p0 = O.init_params(args, seed=19, cell=cell, enc_type="cnn")
    p0["Speller/decode/dense/kernel"] = (p0["Speller/decode/dense/kernel"] * 6).astype(np.float32)   # spread the 5000 logits: no near ties
    p0["Speller/decode/dense/bias"][2] = 1.0
    L.set_cell(cell); L.set_precision(prec)
    st = V_.reset_default_store(device="cuda"); st.load(p0)
    tok = {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2}
    las = LAS(args, Listener, Speller, tok)
    bs = BeamSearch(args, las, tok, None)
    xs, _ = synthetic_batch(1, 1274, 8, 30, seed=23)
    res = bs.decode_batch(None, [xs])[0]
    ref = oracle_decode(xs, p0, args, cell, 4, prec=prec)
    assert len(res) == len(ref) > 0
    assert res[-1].att[-1].shape[-1] == 319
    if prec == "f32":
        assert [b.token_ids for b in res] == [b.token_ids for b in ref]
    else:
        assert res[-1].token_ids == ref[-1].token_ids
    err = abs(float(res[-1].log_prob) - float(ref[-1].log_prob))
    _log("run_sh_beam_search", dict(prec=prec, cell=cell, steps=len(ref[-1].token_ids) - 1, best_score_err=err))
    assert err <= (2e-3 if prec == "f32" else 5e-2), err
