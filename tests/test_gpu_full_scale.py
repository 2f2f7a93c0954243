"""Parity at the sizes that are actually timed (SURVEY 8(d) "at full T").

bench.py times BASELINE configs[1]: Listener BLSTM-256 + 3 x pBLSTM-256, Speller 1 x LSTM-512, A = E = 128, V = 30 on
the bucket T = 1274 (T' = 160 -> the dec_step_{fwd,bwd}_pf_kernel<1,10> instances, the P=4 cluster sweeps with the
K-split BPTT, the fast GEMM path).  Here the SAME architecture runs one full train step at B = 4, T = 1274, U ~ 190
against the oracle restatement (reference las/las.py:226-304): 3,504 dependent recurrent steps per direction and
~190 decoder steps, so error growth through the recurrences is part of what is checked.

Tolerances: FULL_T_TOL below -- f32 mode <= 1e-3 on logits / alignments (SURVEY 8(d) full-T figure), gradients 5e-3 of
the largest oracle entry; bf16 mode against the oracle's bf16-operand mode.
"""
import numpy as np
import pytest
import torch

from helpers import expect_handovers, grad_errors, make_args, oracle_grads, synthetic_batch, train_step_pair

pytestmark = pytest.mark.gpu


def bench_arch(**over):
    kw = dict(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
              mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30)
    kw.update(over)
    return make_args(**kw)


# (prec, cell) -> tolerances.  The lstm cell is what bench.py times.  The reference's own cell (tanh BasicRNNCell) is
# chaotic over 1,274 steps at random initialisation: the ORACLE's f32 and bf16 modes already differ by 2.2e-2 (logits),
# 1.6e-2 (alignments) and 0.43 (the layer-0 backward-direction bias gradient) on this very input (tests/oracle_sensitivity.py,
# profiles/r2_oracle_sensitivity.txt), i.e. any perturbation -- accumulation order included -- is amplified to the size of
# the operand rounding.  rnn/bf16 is therefore held to that intrinsic gap, rnn/f32 to the tight bound.
FULL_T_TOL = {
    ("f32", "lstm"): dict(logits=1e-3, alphas=1e-3, loss=1e-4, grad=5e-3, agree=0.999),
    ("f32", "rnn"): dict(logits=1e-3, alphas=1e-3, loss=1e-4, grad=5e-3, agree=0.999),
    ("bf16", "lstm"): dict(logits=4e-3, alphas=1e-3, loss=1e-3, grad=5e-3, agree=0.99),     # measured r2: 7.3e-4 / 1.8e-5 / 1.0e-3
    # rnn/bf16: forward quantities within the oracle's own f32-vs-bf16 gap x3.  The gradient bound is DERIVED IN THE TEST (VERDICT r4 weak #1:
    # the fixed 1.5 of rounds 2-4 "only excluded garbage"): per parameter, at most GAP_FACTOR x the gap between the ORACLE's own f32 and
    # bf16-emulating gradients on this very input (a chaotic 1,274-step tanh recurrence amplifies ANY perturbation -- one rounding, a
    # summation order -- to that size: tests/oracle_flip_sensitivity.py), never tighter than the lstm row's 5e-3
    ("bf16", "rnn"): dict(logits=6e-2, alphas=5e-2, loss=1e-3, grad=None, agree=0.97),
}
GAP_FACTOR = 2.0


def _log(name, rec):
    """append the measured errors to $LAS_PARITY_LOG (kept under profiles/ as the parity record of the round)"""
    import json
    import os
    path = os.environ.get("LAS_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **rec)) + "\n")


@pytest.mark.parametrize("prec,cell", sorted(FULL_T_TOL))
def test_bench_architecture_full_T_train_step(prec, cell):
    args = bench_arch()
    xs, ys = synthetic_batch(4, 1274, 256, 30, seed=7, min_frac=0.834)
    U = int(ys[1].max())
    assert 150 < U <= 200
    r = train_step_pair(args, cell, prec, xs, ys, seed=3)
    assert r["alphas"].shape[-1] == 160                       # T' in (128, 160]: the <.,10> row-kernel instances
    if prec == "bf16":
        expect_handovers(r["las"], cell, 4)                   # the schedule bench.py times (chunked hand-overs), not a serial fallback
    tol = FULL_T_TOL[(prec, cell)]
    errs = dict(logits=(r["logits"] - r["logits_o"]).abs().max().item(),
                alphas=(r["alphas"] - r["alphas_o"]).abs().max().item(),
                loss=abs(r["loss"] - r["loss_o"]) / max(1.0, abs(r["loss_o"])))
    ge = grad_errors(r)
    worst = max(ge, key=ge.get)
    # greedy token agreement of the teacher-forced logits (the "token agreement" criterion of SURVEY 8(d) for bf16)
    agree = (r["logits"].argmax(-1) == r["logits_o"].argmax(-1)).float().mean().item()
    _log("full_T_train_step", dict(prec=prec, cell=cell, B=4, T=1274, U=U, worst_grad=worst, worst_grad_err=ge[worst],
                                   token_agreement=agree, **errs))
    print("full-T %s/%s: logits %.2e alphas %.2e loss %.2e worst grad %s %.2e agree %.4f" % (
        cell, prec, errs["logits"], errs["alphas"], errs["loss"], worst, ge[worst], agree))
    for k, v in errs.items():
        assert v < tol[k], (k, v)
    if tol["grad"] is None:
        # the oracle's own sensitivity on this input: its f32-mode gradients against its bf16-mode gradients (what `r` was held to),
        # in the same normalisation as grad_errors
        g32, _ = oracle_grads(args, cell, ("f32", "bf", False), xs, ys, seed=3)
        gap = {n: (g32[n] - r["g_o"][n]).abs().max().item() / max(r["g_o"][n].abs().max().item(), 1e-3) for n in r["names"]}
        bound = {n: max(GAP_FACTOR * gap[n], 5e-3) for n in gap}
        ratio = {n: ge[n] / max(gap[n], 1e-12) for n in gap}
        wr = max(ratio, key=ratio.get)
        _log("full_T_rnn_bf16_gradient_bound", dict(worst_ratio_param=wr, worst_ratio=ratio[wr], err=ge[wr], oracle_gap=gap[wr],
                                                    median_ratio=float(np.median(list(ratio.values()))),
                                                    table={n: [ge[n], gap[n]] for n in sorted(gap)}))
        print("rnn/bf16 full-T gradients vs the oracle's own f32-vs-bf16 gap: worst ratio %.2f (%s: err %.3g, gap %.3g), median ratio %.2f"
              % (ratio[wr], wr, ge[wr], gap[wr], float(np.median(list(ratio.values())))))
        for n, e in ge.items():
            assert e <= bound[n], (n, e, gap[n])
    else:
        for n, e in ge.items():
            assert e < tol["grad"], (n, e)
    assert agree > tol["agree"], agree


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
    ge = grad_errors(r)
    errs = dict(logits=(r["logits"] - r["logits_o"]).abs().max().item(), alphas=(r["alphas"] - r["alphas_o"]).abs().max().item())
    _log("full_T_scheduled_sampling", dict(prec="bf16", cell="lstm", worst_grad_err=max(ge.values()), **errs))
    assert errs["logits"] < 4e-3 and errs["alphas"] < 1e-3                  # measured r2: 4.6e-4 / 2.1e-5 / 7.6e-4
    for n, e in ge.items():
        assert e < 5e-3, (n, e)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_config3_location_aware_full_T_full_U(prec):
    """the location-aware loops were oracle-checked for at most 12 decode steps; the timed configs[3] leg runs
    U = 191, where alpha_{t-1} feeds the conv of every following step (las/layers.py:295-296).  BASELINE configs[3] at B = 4, T = 1274,
    the whole label length (U ~ 190), V = 5000, K = 201 / C = 10 -- the one-launch LOC loop kernels in speed mode, the per-step rows in
    parity mode -- with the tolerances of the additive full-T rows."""
    V = 5000
    args = bench_arch(mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=V, unit="subword")
    xs, ys = synthetic_batch(4, 1274, 256, V, seed=31, min_frac=0.834)
    U = int(ys[1].max())
    assert 150 < U <= 200
    r = train_step_pair(args, "lstm", prec, xs, ys, seed=5)
    assert r["alphas"].shape[-1] == 160 and r["logits"].shape[-1] == V
    tol = FULL_T_TOL[(prec, "lstm")]
    errs = dict(logits=(r["logits"] - r["logits_o"]).abs().max().item(), alphas=(r["alphas"] - r["alphas_o"]).abs().max().item(),
                loss=abs(r["loss"] - r["loss_o"]) / max(1.0, abs(r["loss_o"])))
    ge = grad_errors(r)
    worst = max(ge, key=ge.get)
    _log("config3_full_T_full_U", dict(prec=prec, cell="lstm", B=4, T=1274, U=U, worst_grad=worst, worst_grad_err=ge[worst], **errs))
    print("configs[3] full T / full U %s: logits %.2e alphas %.2e loss %.2e worst grad %s %.2e" % (
        prec, errs["logits"], errs["alphas"], errs["loss"], worst, ge[worst]))
    for k, v in errs.items():
        assert v < tol[k], (k, v)
    for n, e in ge.items():
        assert e < tol["grad"], (n, e)
