"""Parity at the sizes that are actually timed (SURVEY 8(d) "at full T").

bench.py times BASELINE configs[1]: Listener BLSTM-256 + 3 x pBLSTM-256, Speller 1 x LSTM-512, A = E = 128, V = 30 on
the bucket T = 1274 (T' = 160 -> the dec_step_{fwd,bwd}_pf_kernel<1,10> instances, the P=4 cluster sweeps with the
K-split BPTT, the fast GEMM path).  Here the SAME architecture runs one full train step at B = 4, T = 1274, U ~ 190
against the oracle restatement (reference las/las.py:226-304): 3,504 dependent recurrent steps per direction and
~190 decoder steps, so error growth through the recurrences is part of what is checked.

Tolerances: f32 mode <= 1e-3 on logits / alignments (SURVEY 8(d) full-T figure), gradients 5e-3 of the largest
oracle entry; bf16 mode against the oracle's bf16-operand mode: logits 2e-2, alignments 1e-2, loss 2e-3, gradients 3e-2.
"""
import numpy as np
import pytest
import torch

from helpers import grad_errors, make_args, synthetic_batch, train_step_pair

pytestmark = pytest.mark.gpu


def bench_arch(**over):
    kw = dict(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
              mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30)
    kw.update(over)
    return make_args(**kw)


@pytest.mark.parametrize("prec,cell", [("f32", "lstm"), ("bf16", "lstm"), ("bf16", "rnn")])
def test_bench_architecture_full_T_train_step(prec, cell):
    args = bench_arch()
    xs, ys = synthetic_batch(4, 1274, 256, 30, seed=7, min_frac=0.834)
    U = int(ys[1].max())
    assert 150 < U <= 200
    r = train_step_pair(args, cell, prec, xs, ys, seed=3)
    assert r["alphas"].shape[-1] == 160                       # T' in (128, 160]: the <.,10> row-kernel instances
    tol = dict(logits=1e-3, alphas=1e-3, loss=1e-4, grad=5e-3) if prec == "f32" else \
        dict(logits=2e-2, alphas=1e-2, loss=2e-3, grad=3e-2)
    errs = dict(logits=(r["logits"] - r["logits_o"]).abs().max().item(),
                alphas=(r["alphas"] - r["alphas_o"]).abs().max().item(),
                loss=abs(r["loss"] - r["loss_o"]) / max(1.0, abs(r["loss_o"])))
    ge = grad_errors(r)
    worst = max(ge, key=ge.get)
    print("full-T %s/%s: logits %.2e alphas %.2e loss %.2e worst grad %s %.2e" % (cell, prec, errs["logits"], errs["alphas"],
                                                                                  errs["loss"], worst, ge[worst]))
    for k, v in errs.items():
        assert v < tol[k], (k, v)
    for n, e in ge.items():
        assert e < tol["grad"], (n, e)
    # greedy token agreement of the teacher-forced logits (the "token agreement" criterion of SURVEY 8(d) for bf16)
    agree = (r["logits"].argmax(-1) == r["logits_o"].argmax(-1)).float().mean().item()
    assert agree > (0.999 if prec == "f32" else 0.98), agree


def test_bench_architecture_full_T_with_scheduled_sampling_bf16():
    """configs[2] arithmetic at the bench size: label smoothing + scheduled sampling with ON-DEVICE draws (tokens_in = -2,
    in-kernel logits): the sampled tokens the kernel resolved are fed to the oracle, which must reproduce the step."""
    args = bench_arch(scheduled_sampling=True)
    xs, ys = synthetic_batch(4, 1274, 256, 30, seed=9, min_frac=0.834)
    U = int(ys[1].max())
    rng = np.random.RandomState(5)
    coins = rng.rand(U) < 0.6
    # pass 1: on-device sampling; recover the draws (tokens_in[t] is the token entering step t = sample of step t-1)
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    p0 = O.init_params(args, seed=3, cell="lstm")
    L.set_cell("lstm"); L.set_precision("bf16")
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, {})
    las.train(xs, ys, coins=coins)
    tok = las.speller.last_tokens_in.cpu().numpy()            # [U, B]
    assert (tok >= 0).all() and (tok < 30).all()
    sampled = np.zeros((4, U), np.int32)
    sampled[:, :U - 1] = tok[1:].T                            # sampled[:, t] enters step t+1
    teacher_steps = np.nonzero(coins[:U - 1])[0]
    assert (tok[1:][teacher_steps] == ys[0][:, :U - 1].T[teacher_steps]).all()
    # pass 2: same weights, the recovered draws injected on both sides
    r = train_step_pair(args, "lstm", "bf16", xs, ys, seed=3, coins=coins, sampled=sampled)
    assert (r["tokens_in"].numpy() == tok).all()
    assert (r["logits"] - r["logits_o"]).abs().max().item() < 2e-2
    assert (r["alphas"] - r["alphas_o"]).abs().max().item() < 1e-2
    for n, e in grad_errors(r).items():
        assert e < 3e-2, (n, e)
