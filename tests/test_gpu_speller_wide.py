"""The wide Speller path (csrc/speller_wide.h, round 6): the decode loop and its gradient as per-step launch chains that fill the machine --
query projection as ONE product over all rows, energies (+ location conv) and context on (slice, utterance) workgroups, every layer's cell on
pre-packed MFMA fragments -- for the geometries the one-launch loop kernels do not take: multi-layer decoders, S = D NL up to 2048 and more,
T' > 224, location-aware attention with any filter (reference las/las.py:72-160,185-199; las/layers.py:215-311).

 * PARITY MODE, forced (LAS_SPELLER_WIDE): fp32 throughout, against the oracle's fp32 Speller at 1e-4-class tolerances: this is the tight
   statement about the kernels' arithmetic (slice statistics of the softmax, the conv and its transpose, d alpha / d energy / dq / du /
   d f, the after-loop keys / Wf / filter contractions), for both cells, both attention modes, 1-3 layers, ragged sizes, sampled tokens.
 * SPEED MODE, as selected by default: the same launches on bf16 operands against the oracle's bf16-row mode.
 * wide == not wide: the same call through round 5's per-utterance row kernels (LAS_SPELLER_NO_WIDE) in parity mode.
 * fused == launch per phase: the fused attention launch (wide_attend_kernel: an in-kernel hand-over between an utterance's workgroups) and
   the tanh cells as epilogues of their products, forward and backward, against LAS_SPELLER_NO_FUSED_STEP -- bit for bit.
"""
import numpy as np
import pytest
import torch

from helpers import make_args, wide_eligible

pytestmark = pytest.mark.gpu


def _run(flags, prec, cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=None, V=30, dropout=0.0, add_vn=False, oracle=True):
    from las import _hip, layers as L, variables as Vs
    from las.las import Speller
    from oracle import las_oracle as O
    saved = _hip.speller_flags
    _hip.speller_flags = flags
    try:
        L.set_cell(cell)
        L.set_precision(prec)
        Vs.reset_default_store(device="cuda", seed=3)
        args = make_args(enc_units=Hd, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=E, attention_size=A, mode="add",
                         vocab_size=V, enc_type="cnn")                       # (enc_type cnn: the Speller's hidden_dim is enc_units)
        if loc is not None:
            args.mode, args.loc_kernel_size, args.loc_num_channels = "loc", loc[0], loc[1]
        args.dropout_rate, args.add_vn = dropout, add_vn
        torch.manual_seed(1234)                          # (the embedding dropout mask / variational noise come from torch's generator)
        sp = Speller(args)
        rng = np.random.RandomState(1)
        enc_np = rng.randn(B, Tp, Hd).astype(np.float32) * 0.5
        enc = torch.tensor(enc_np, device="cuda", requires_grad=True)
        enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
        enc_len[0] = Tp
        y = rng.randint(3, V, size=(B, U))
        coins, sampled = np.ones(U, bool), None
        if mixed:
            coins = rng.rand(U) < 0.5
            sampled = rng.randint(3, V, size=(B, U)).astype(np.int32)
        wgt = torch.tensor(rng.randn(B, U, V).astype(np.float32))
        logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True, coins=coins, sampled=sampled)
        fam = _hip.speller_last_variant()
        (logits * wgt.cuda()).sum().backward()
        _hip.join_side_stream()
        torch.cuda.synchronize()
        _hip.check_status()
        fam["bwd"] = _hip.speller_last_variant()["bwd"]
        st = Vs.default_store()
        grads = {n: st.vars[n].grad.detach().cpu().clone() for n in st.order}
        grads["enc"] = enc.grad.detach().cpu().clone()
        p0 = {n: st.vars[n].detach().cpu().numpy() for n in st.order}
        if not oracle:
            return dict(logits=logits.detach().cpu(), alphas=alphas.detach().cpu(), grads=grads, fam=fam)
        if prec == "bf16":
            wide = not (flags & _hip.SPELLER_NO_WIDE) and wide_eligible(args, U, bool(flags & _hip.SPELLER_WIDE))
            O.set_precision("bf16", "bf" if (wide or args.mode == "add") else "f32")
        try:
            po = O.to_torch(p0, requires_grad=True)
            enc_o = torch.tensor(enc_np, requires_grad=True)
            lo, ao = O.speller_forward(enc_o, enc_len.astype(np.float64), U, po, args, cell, teacher=torch.tensor(y), is_training=True,
                                       coins=coins, sampled=None if sampled is None else torch.tensor(sampled))
            (lo * wgt).sum().backward()
        finally:
            O.set_precision("f32")
        go = {n: po[n].grad for n in po if po[n].grad is not None}
        go["enc"] = enc_o.grad
        return dict(logits=logits.detach().cpu(), alphas=alphas.detach().cpu(), grads=grads, lo=lo.detach(), ao=ao.detach(), go=go, fam=fam)
    finally:
        _hip.speller_flags = saved


def _check(r, tl, ta, tg):
    assert (r["alphas"] - r["ao"]).abs().max().item() < ta, (r["alphas"] - r["ao"]).abs().max().item()
    assert (r["alphas"].sum(-1) - 1).abs().max().item() < 1e-4
    el = (r["logits"] - r["lo"]).abs().max().item() / max(1.0, r["lo"].abs().max().item())
    assert el < tl, el
    assert set(r["go"]) <= set(r["grads"])
    worst = ("", 0.0)
    for n in sorted(r["go"]):
        scale = max(r["go"][n].abs().max().item(), 1e-3)
        err = (r["grads"][n] - r["go"][n]).abs().max().item() / scale
        if err > worst[1]:
            worst = (n, err)
        assert err < tg, (n, err, scale)
    return el, worst


MORE_ROWS = [
    # more utterances than a quarter of the machine's CUs: fewer slices per utterance (B = 96: two frame slices; B = 300: one)
    ("lstm", 2, 64, 32, 32, 32, 96, 40, 4, False, (7, 3)),
    ("rnn", 2, 64, 32, 64, 32, 300, 23, 3, True, None),
]

SHAPES = [
    # cell, NL,  D,   A,  Hd,  E,  B, Tp,  U, mixed, loc
    ("lstm", 1, 64, 32, 32, 32, 3, 21, 5, False, None),            # the smallest: one layer, additive
    ("rnn", 2, 64, 32, 64, 32, 4, 21, 7, True, None),              # two layers (query = concat of the states), sampled tokens
    ("lstm", 2, 128, 64, 64, 64, 5, 70, 6, False, (7, 3)),         # location-aware, small filter, ragged sizes
    ("rnn", 2, 128, 128, 96, 64, 3, 131, 5, True, (201, 10)),      # the reference's filter K = 201, C = 10: both borders clipped
    ("lstm", 3, 64, 136, 40, 24, 2, 45, 4, False, (31, 16)),       # three layers, attention width > 128 (two chunks per lane), C = 16
    ("lstm", 2, 256, 128, 128, 64, 9, 319, 4, False, (201, 10)),   # T' = 319 (run.sh's frame count): eight frame slices
    ("rnn", 1, 512, 128, 512, 128, 48, 160, 4, False, (201, 10)),  # the bench geometry's rows at B = 48: five slices per utterance
    ("lstm", 2, 1024, 128, 512, 256, 4, 319, 4, True, (201, 10)),  # run.sh's Speller sizes (S = 2048, T' = 319)
]


@pytest.mark.parametrize("shape", SHAPES + MORE_ROWS)
def test_wide_path_parity_mode_matches_the_fp32_oracle(shape):
    from las import _hip
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = shape
    r = _run(_hip.SPELLER_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc)
    assert "wide" in r["fam"]["fwd"] and "wide" in r["fam"]["bwd"], r["fam"]
    el, worst = _check(r, 2e-5, 2e-5, 5e-4)
    print("wide f32 %s: logits %.1e worst grad %s %.1e" % (shape[:4], el, worst[0], worst[1]))


@pytest.mark.parametrize("shape", SHAPES)
def test_wide_path_speed_mode_matches_the_bf16_row_oracle(shape):
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = shape
    forced = 0
    if NL == 1 and loc is None:
        from las import _hip
        forced = _hip.SPELLER_WIDE                       # (one additive layer is the loop kernels' by default)
    r = _run(forced, "bf16", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc)
    if NL >= 2 or forced or (loc is not None and "loop" not in r["fam"]["fwd"]):
        assert "wide" in r["fam"]["fwd"] and "wide" in r["fam"]["bwd"], r["fam"]
    el, worst = _check(r, 5e-3, 2e-3, 2e-2)
    print("wide bf16 %s: %s logits %.1e worst grad %s %.1e" % (shape[:4], r["fam"]["fwd"], el, worst[0], worst[1]))


@pytest.mark.parametrize("prec", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [SHAPES[1], SHAPES[3], SHAPES[5], SHAPES[6], SHAPES[7], MORE_ROWS[0]])
def test_wide_path_fused_launches_equal_one_launch_per_phase_bit_for_bit(shape, prec):
    """wide_attend_kernel (energies -> granules -> alignment / context in ONE launch) and the tanh-cell epilogues of the skinny products
    (las_skinny_gemm_bf16_tanh / _tanh_bwd) re-arrange launches, not arithmetic: the same operands in the same order."""
    from las import _hip
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = shape
    a = _run(_hip.SPELLER_WIDE, prec, cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc, oracle=False)
    b = _run(_hip.SPELLER_WIDE | _hip.SPELLER_NO_FUSED_STEP, prec, cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc, oracle=False)
    assert "wide" in a["fam"]["fwd"] and "wide" in b["fam"]["fwd"] and "wide" in a["fam"]["bwd"] and "wide" in b["fam"]["bwd"]
    assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["alphas"], b["alphas"])
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), n


def test_wide_fused_attention_poll_timeout_is_reported_and_the_next_call_is_clean():
    """The in-kernel hand-over of wide_attend_kernel is bounded: with a poll budget of 2 (LAS_SPELLER_SPIN_LOG2(1)) some workgroup gives up on
    its partners, the status word says so (the host raises at its next check; LAS.train would re-run the step on the launch-per-phase form),
    every later launch of the call stops waiting, nothing hangs -- and the next call with the normal budget is clean and equals the
    launch-per-phase form again."""
    from las import _hip
    # (T' = 21: three frame slices but eight context slices per utterance -- five workgroups of an utterance have no energies of their own
    #  and poll at once, microseconds before the other three have staged the K = 201 filter, let alone published)
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = "rnn", 2, 64, 32, 64, 32, 4, 21, 7, True, (201, 10)
    with pytest.raises(RuntimeError, match="status 3"):
        _run(_hip.SPELLER_WIDE | _hip.speller_spin_log2(1), "bf16", cell, NL, D, A, Hd, E, B, Tp, 12, mixed, loc=loc, oracle=False)
    _hip.clear_status(torch.device("cuda"))
    a = _run(_hip.SPELLER_WIDE, "bf16", cell, NL, D, A, Hd, E, B, Tp, 12, mixed, loc=loc, oracle=False)
    b = _run(_hip.SPELLER_WIDE | _hip.SPELLER_NO_FUSED_STEP, "bf16", cell, NL, D, A, Hd, E, B, Tp, 12, mixed, loc=loc, oracle=False)
    assert torch.equal(a["logits"], b["logits"])
    for n in a["grads"]:
        assert torch.equal(a["grads"][n], b["grads"][n]), n


@pytest.mark.parametrize("shape", [SHAPES[1], SHAPES[3], SHAPES[5]])
def test_wide_path_equals_the_per_utterance_rows_in_parity_mode(shape):
    from las import _hip
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = shape
    a = _run(_hip.SPELLER_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc)
    b = _run(_hip.SPELLER_NO_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc)
    assert "wide" in a["fam"]["fwd"] and "wide" not in b["fam"]["fwd"]
    assert (a["logits"] - b["logits"]).abs().max().item() < 2e-5 and (a["alphas"] - b["alphas"]).abs().max().item() < 2e-6
    for n in a["grads"]:
        scale = max(b["grads"][n].abs().max().item(), 1e-3)
        assert (a["grads"][n] - b["grads"][n]).abs().max().item() / scale < 5e-4, n


def test_wide_path_greedy_inference_resolves_tokens_on_the_device():
    """step_logits: the state launch projects the top layer's state of step t-1 onto the vocabulary, takes the arg-max and feeds it to step t
    (las/las.py:101-105) -- LAS.inference through the wide path against the oracle's greedy decode, two layers, location-aware."""
    from las import _hip, layers as L, variables as V_
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    from helpers import synthetic_batch
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=2, embedding_size=32, attention_size=32, mode="loc",
                     loc_kernel_size=7, loc_num_channels=3, vocab_size=30, convert_rate=0.2)
    xs, _ = synthetic_batch(3, 40, 8, 30, seed=5)
    p0 = O.init_params(args, seed=5, cell="lstm")
    with torch.no_grad():
        lo, yo = O.greedy_inference((torch.tensor(xs[0]), xs[1]), O.to_torch(p0), args, "lstm")
    saved = _hip.speller_flags
    _hip.speller_flags = _hip.SPELLER_WIDE
    try:
        L.set_cell("lstm"); L.set_precision("f32")
        st = V_.reset_default_store(device="cuda"); st.load(p0)
        las = LAS(args, Listener, Speller, {})
        logits, y_hat = las.inference(xs)
        assert "wide" in _hip.speller_last_variant()["fwd"]
    finally:
        _hip.speller_flags = saved
    assert (logits.cpu() - lo).abs().max().item() < 5e-4
    assert torch.equal(y_hat.cpu(), yo)


@pytest.mark.parametrize("dropout,add_vn", [(0.5, False), (0.0, True), (0.3, True)])
def test_wide_path_with_embedding_dropout_and_variational_noise_equals_the_per_utterance_rows(dropout, add_vn):
    """tf.layers.dropout on the embedded input token (las/las.py:107-108) and --add_vn's per-look-up noise on the embedding matrix
    (las/las.py:164-166) reach the wide path's cell input rows as they reach the row kernels': the same masks / noise (torch generator seeded
    alike) through both families, parity mode, two layers, location-aware."""
    from las import _hip
    shape = ("lstm", 2, 128, 64, 64, 64, 5, 70, 6, False, (7, 3))
    cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc = shape
    a = _run(_hip.SPELLER_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc, dropout=dropout, add_vn=add_vn, oracle=False)
    b = _run(_hip.SPELLER_NO_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc, dropout=dropout, add_vn=add_vn, oracle=False)
    c = _run(_hip.SPELLER_NO_WIDE, "f32", cell, NL, D, A, Hd, E, B, Tp, U, mixed, loc=loc, oracle=False)
    assert "wide" in a["fam"]["fwd"] and "wide" not in b["fam"]["fwd"]
    assert (a["logits"] - b["logits"]).abs().max().item() < 2e-5 and (a["alphas"] - b["alphas"]).abs().max().item() < 2e-6
    assert (b["logits"] - c["logits"]).abs().max().item() > 1e-3           # (the masks / the noise really were applied)
    for n in a["grads"]:
        scale = max(b["grads"][n].abs().max().item(), 1e-3)
        assert (a["grads"][n] - b["grads"][n]).abs().max().item() / scale < 5e-4, n
