"""On-device scheduled sampling (reference las/las.py:101-105,170-175: tf.distributions.Categorical(logits).sample()).

The row kernels draw Categorical(logits) as arg-max(logits + Gumbel noise) (tokens_in = -2).  Checked here:
  * the draws follow softmax(logits): chi-square goodness of fit over >= 20k draws,
  * a fixed (seed, rank) reproduces the draws, another seed / rank changes them,
  * every row-kernel family resolves tokens in place to valid ids and agrees with the oracle on the step that
    consumed them (gradients do not flow through the draws).
"""
import numpy as np
import pytest
import torch

from helpers import make_args

pytestmark = pytest.mark.gpu


def _speller(prec, D=64, A=32, H=32, NL=1, V=30, flags=0):
    from las import _hip, layers as L, variables as Vs
    from las.las import Speller
    L.set_cell("lstm"); L.set_precision(prec)
    Vs.reset_default_store(device="cuda", seed=3)
    _hip.speller_flags = flags
    args = make_args(enc_units=H, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=32, attention_size=A,
                     mode="add", vocab_size=V, enc_type="pblstm")
    return Speller(args), args


@pytest.mark.parametrize("prec,flags", [("f32", 0), ("bf16", 0), ("bf16", 1), ("bf16", 2), ("bf16", 4)])
def test_gumbel_draws_follow_softmax_and_are_reproducible(prec, flags):
    from las import _hip, variables as Vs
    V, Bn, Tp = 30, 2048, 12
    sp, args = _speller(prec, V=V, flags=flags)
    try:
        sp._params()
        with torch.no_grad():                        # a peaked distribution (a random-init vocabulary layer is nearly flat)
            Vs.default_store().vars["Speller/decode/dense/kernel"].mul_(12.0)
        rng = np.random.RandomState(0)
        enc1 = rng.randn(1, Tp, 64).astype(np.float32)
        enc = torch.tensor(np.repeat(enc1, Bn, 0), device="cuda")         # identical rows -> identical step-0 logits
        enc_len = np.full(Bn, Tp)
        y = np.full((Bn, 2), 5)
        coins = np.array([False, False])                                   # the token entering step 1 is a draw
        counts = np.zeros(V)
        st = Vs.default_store()
        draws = {}
        for gs in range(10):
            st.global_step = gs
            with torch.no_grad():
                logits, _, _ = sp(enc, enc_len, 2, teacher=y, is_training=True, coins=coins)
            tok = sp.last_tokens_in[1].cpu().numpy()
            assert tok.min() >= 0 and tok.max() < V
            draws[gs] = tok
            counts += np.bincount(tok, minlength=V)
            l0 = logits[:, 0].cpu().double()
            assert (l0 - l0[0]).abs().max().item() < 1e-5                  # identical rows
        p = torch.softmax(l0[0], -1).numpy()
        n = counts.sum()
        assert n == 10 * Bn
        keep = p * n >= 5                                                  # pool rare classes (chi-square validity)
        exp = np.append(p[keep] * n, p[~keep].sum() * n)
        obs = np.append(counts[keep], counts[~keep].sum())
        if exp[-1] == 0:
            exp, obs = exp[:-1], obs[:-1]
        chi2 = ((obs - exp) ** 2 / exp).sum()
        dof = len(exp) - 1
        # 99.99 % quantile of chi-square(dof) via Wilson-Hilferty; a wrong sampler (e.g. arg-max, uniform) is off by 100s
        z = 3.72
        crit = dof * (1 - 2 / (9 * dof) + z * (2 / (9 * dof)) ** 0.5) ** 3
        assert chi2 < crit, (chi2, crit, dof)
        # power of the test: the same counts are far from a uniform draw and from an arg-max "sampler"
        uni = np.full(V, n / V)
        assert ((counts - uni) ** 2 / uni).sum() > 10 * crit
        assert (counts > 0).sum() >= 5 and counts.max() < 0.9 * n
        # determinism: same (step, rank) -> same draws; other step / rank -> different draws
        st.global_step = 3
        with torch.no_grad():
            sp(enc, enc_len, 2, teacher=y, is_training=True, coins=coins)
        assert (sp.last_tokens_in[1].cpu().numpy() == draws[3]).all()
        sp.rank = 1
        with torch.no_grad():
            sp(enc, enc_len, 2, teacher=y, is_training=True, coins=coins)
        assert (sp.last_tokens_in[1].cpu().numpy() != draws[3]).mean() > 0.3
    finally:
        _hip.speller_flags = 0


@pytest.mark.parametrize("prec,NL,flags", [("f32", 1, 0), ("f32", 2, 0), ("bf16", 1, 0), ("bf16", 1, 1), ("bf16", 2, 0), ("bf16", 1, 4)])
def test_on_device_sampling_step_matches_oracle_given_the_draws(prec, NL, flags):
    """training with in-kernel logits (step_logits) followed by las_speller_bwd: feed the oracle the tokens the kernel
    drew and compare logits, alignments and every gradient."""
    from las import _hip, variables as Vs
    from oracle import las_oracle as O
    B, Tp, U, V = 5, 37, 9, 30
    sp, args = _speller(prec, D=64, A=32, H=32, NL=NL, V=V, flags=flags)
    try:
        st = Vs.default_store()
        rng = np.random.RandomState(2)
        enc_np = (rng.randn(B, Tp, 64) * 0.5).astype(np.float32)
        enc = torch.tensor(enc_np, device="cuda", requires_grad=True)
        enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
        y = rng.randint(3, V, size=(B, U))
        coins = rng.rand(U) < 0.5
        coins[0] = False
        w = torch.tensor(rng.randn(B, U, V).astype(np.float32))
        logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True, coins=coins)
        (logits * w.cuda()).sum().backward()
        _hip.join_side_stream()
        torch.cuda.synchronize()
        tok = sp.last_tokens_in.cpu().numpy()                     # [U, B] resolved in place
        assert (tok >= 0).all() and (tok < V).all()
        sampled = np.zeros((B, U), np.int64)
        sampled[:, :U - 1] = tok[1:].T
        p0 = {n: st.vars[n].detach().cpu().numpy() for n in st.order}
        O.set_precision("bf16" if prec == "bf16" else "f32", "bf")
        try:
            po = O.to_torch(p0, requires_grad=True)
            enc_o = torch.tensor(enc_np, requires_grad=True)
            lo, ao = O.speller_forward(enc_o, enc_len.astype(np.float64), U, po, args, "lstm", teacher=torch.tensor(y),
                                       is_training=True, coins=coins, sampled=torch.tensor(sampled))
            (lo * w).sum().backward()
        finally:
            O.set_precision("f32")
        tl, ta, tg = (2e-4, 1e-4, 2e-3) if prec == "f32" else (4e-3, 2e-3, 2e-2)
        assert (logits.detach().cpu() - lo.detach()).abs().max().item() < tl
        assert (alphas.detach().cpu() - ao.detach()).abs().max().item() < ta
        ge = (enc.grad.cpu() - enc_o.grad).abs().max().item() / max(enc_o.grad.abs().max().item(), 1e-3)
        assert ge < tg, ("enc", ge)
        for n in st.order:
            go = po[n].grad
            if go is None:
                continue
            e = (st.vars[n].grad.cpu() - go).abs().max().item() / max(go.abs().max().item(), 1e-3)
            assert e < tg, (n, e)
    finally:
        _hip.speller_flags = 0


@pytest.mark.parametrize("prec,NL,flags", [("f32", 1, 0), ("bf16", 1, 0), ("bf16", 2, 0), ("bf16", 1, 4)])
def test_logits_only_at_the_sampled_steps_draw_the_same_tokens_as_logits_at_every_step(prec, NL, flags):
    """step_logits = 2 (round 6: in-loop logits and draws only where the entering token is device-resolved, the loss's logits from the batched
    product behind the loop) against step_logits = 1 (in the loop at every step): the SAME tokens are drawn -- the draw sees the same in-loop
    values -- the alignments are the same bits, and logits / gradients agree to the summation order of the projection."""
    from las import _hip, variables as Vs
    B, Tp, U, V = 5, 37, 12, 30
    res = []
    for every in (True, False):
        sp, args = _speller(prec, D=64, A=32, H=32, NL=NL, V=V, flags=flags)
        try:
            sp.logits_every_step = every
            st = Vs.default_store()
            rng = np.random.RandomState(2)
            enc = torch.tensor((rng.randn(B, Tp, 64) * 0.5).astype(np.float32), device="cuda", requires_grad=True)
            enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
            y = rng.randint(3, V, size=(B, U))
            coins = rng.rand(U) < 0.6
            coins[0], coins[3] = False, True
            w = torch.tensor(rng.randn(B, U, V).astype(np.float32))
            logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True, coins=coins)
            (logits * w.cuda()).sum().backward()
            _hip.join_side_stream()
            torch.cuda.synchronize()
            _hip.check_status()
            res.append((sp.last_tokens_in.cpu().clone(), logits.detach().cpu(), alphas.detach().cpu(), enc.grad.cpu().clone(),
                        {n: st.vars[n].grad.detach().cpu().clone() for n in st.order}))
        finally:
            _hip.speller_flags = 0
    (tok1, l1, a1, g1, p1), (tok2, l2, a2, g2, p2) = res
    assert (tok1 >= 0).all() and torch.equal(tok1, tok2)
    assert torch.equal(a1, a2)
    tol = 1e-5 if prec == "f32" else 1e-3
    assert (l1 - l2).abs().max().item() < tol * max(1.0, l1.abs().max().item())
    assert (g1 - g2).abs().max().item() <= tol * max(g1.abs().max().item(), 1e-3)
    for n in p1:
        assert (p1[n] - p2[n]).abs().max().item() <= tol * max(p1[n].abs().max().item(), 1e-3), n


@pytest.mark.parametrize("prec,flags", [("f32", 0), ("bf16", 0), ("bf16", 1)])
def test_variational_noise_on_the_embedding_matrix(prec, flags):
    """--add_vn (reference las/las.py:164-166): every look-up adds a fresh N(0, 0.075) matrix to the WHOLE embedding matrix.
    With the noise matrices injected on both sides the Speller must match the oracle; without injection two calls differ."""
    from las import _hip, variables as Vs
    from oracle import las_oracle as O
    B, Tp, U, V = 4, 23, 6, 30
    sp, args = _speller(prec, D=64, A=32, H=32, NL=1, V=V, flags=flags)
    args.add_vn = True
    try:
        st = Vs.default_store()
        rng = np.random.RandomState(4)
        enc_np = (rng.randn(B, Tp, 64) * 0.5).astype(np.float32)
        enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
        y = rng.randint(3, V, size=(B, U))
        y[1] = y[0]                                                  # two rows looking up the same tokens share the noise
        noise = (rng.randn(U, V, 32) * 0.075).astype(np.float32)
        sp.vn_noise = torch.tensor(noise, device="cuda")
        enc = torch.tensor(enc_np, device="cuda", requires_grad=True)
        w = torch.tensor(rng.randn(B, U, V).astype(np.float32))
        logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True)
        (logits * w.cuda()).sum().backward()
        _hip.join_side_stream()
        torch.cuda.synchronize()
        p0 = {n: st.vars[n].detach().cpu().numpy() for n in st.order}
        O.set_precision("bf16" if prec == "bf16" else "f32", "bf")
        try:
            po = O.to_torch(p0, requires_grad=True)
            lo, ao = O.speller_forward(torch.tensor(enc_np), enc_len.astype(np.float64), U, po, args, "lstm", teacher=torch.tensor(y),
                                       is_training=True, emb_noise=torch.tensor(noise))
            (lo * w).sum().backward()
        finally:
            O.set_precision("f32")
        tl, tg = (2e-4, 2e-3) if prec == "f32" else (4e-3, 2e-2)
        assert (logits.detach().cpu() - lo.detach()).abs().max().item() < tl
        ge = po["embedding/embedding_matrix"].grad
        assert (st.vars["embedding/embedding_matrix"].grad.cpu() - ge).abs().max().item() / max(ge.abs().max().item(), 1e-3) < tg
        # the noise matters, and without an injected matrix it is drawn afresh on every call
        sp.vn_noise = None
        with torch.no_grad():
            l1, _, _ = sp(enc, enc_len, U, teacher=y, is_training=True)
            l2, _, _ = sp(enc, enc_len, U, teacher=y, is_training=True)
        assert (l1 - l2).abs().max().item() > 1e-4 and (l1 - logits.detach()).abs().max().item() > 1e-4
    finally:
        _hip.speller_flags = 0
