"""End-to-end parity of one LAS.train step (C ABI path) against the oracle restatement
(oracle/las_oracle.py train_step; reference las/las.py:226-304): logits, alignments, loss, every
parameter gradient and the Adam-updated parameters, on identical seeded inputs and weights.

Tolerances: see TOL below (fp32 mode: SURVEY 8(d); bf16 mode: against the oracle's bf16-operand mode)."""
import numpy as np
import pytest
import torch

from helpers import grad_errors, make_args, synthetic_batch, train_step_pair

pytestmark = pytest.mark.gpu

CONFIGS = [
    # cell, mode, NL, H, D, prec, mixed_sampling
    ("rnn", "add", 1, 48, 64, "f32", False),
    ("lstm", "add", 1, 64, 64, "f32", False),
    ("lstm", "add", 2, 48, 48, "f32", True),
    ("rnn", "loc", 2, 64, 64, "f32", False),
    ("lstm", "loc", 1, 64, 96, "f32", True),
    ("lstm", "add", 1, 64, 64, "bf16", False),
    ("rnn", "add", 1, 128, 64, "bf16", False),
    ("lstm", "add", 2, 64, 64, "bf16", True),
    ("lstm", "loc", 1, 64, 64, "bf16", False),
    ("rnn", "loc", 2, 64, 64, "bf16", True),
]


# Stated tolerances.  f32 mode: SURVEY 8(d) (<= 1e-4 on logits / alignments at these sizes, 2e-3 relative on gradients).
# bf16 mode is held against the oracle's bf16-OPERAND mode (oracle.set_precision('bf16'): same rounding points, fp32
# accumulation), so what remains is accumulation order, fast transcendentals and the few values that land on the other
# side of a bf16 rounding boundary.  How far ONE such flip moves the outputs depends on the cell: perturbing the inputs of
# these very configurations by 1e-5 (a handful of flipped bf16 inputs) moves the ORACLE's own logits by 3.8e-4 with lstm
# cells and by 8.2e-3 with the reference's tanh BasicRNNCell, which has no gates to damp it (tests/oracle_flip_sensitivity.py
# ; profiles/r2_oracle_sensitivity.txt).  Hence two bf16 rows: lstm <= 4e-3 logits / 2e-3 alignments / 2e-2
# gradients; rnn 3e-2 / 1.5e-2 / 0.1 (a few flips), with the f32 rows pinning the rnn kernels' arithmetic at 1e-4.
TOL = {("f32", "lstm"): dict(logits=5e-4, alphas=1e-4, loss=1e-4, grad=2e-3), ("f32", "rnn"): dict(logits=5e-4, alphas=1e-4, loss=1e-4, grad=2e-3),
       ("bf16", "lstm"): dict(logits=4e-3, alphas=2e-3, loss=2e-3, grad=2e-2), ("bf16", "rnn"): dict(logits=3e-2, alphas=1.5e-2, loss=5e-3, grad=0.1)}


def _run_pair(cfg, B=5, T=37, U_max=9):
    cell, mode, NL, H, D, prec, mixed = cfg
    args = make_args(enc_units=H, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=32,
                     attention_size=32, mode=mode, loc_kernel_size=11, loc_num_channels=3, lr=1e-3, grad_clip=5.0)
    xs, ys = synthetic_batch(B, T, U_max, args.vocab_size, seed=H + D)
    U = int(ys[1].max())
    rng = np.random.RandomState(3)
    coins = np.ones(U, bool)
    sampled = None
    if mixed:
        coins = rng.rand(U) < 0.5
        sampled = rng.randint(3, args.vocab_size, size=(B, U)).astype(np.int32)
    return train_step_pair(args, cell, prec, xs, ys, seed=11, coins=coins, sampled=sampled)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_train_step_matches_oracle(cfg):
    r = _run_pair(cfg)
    prec = cfg[5]
    tol = TOL[(prec, cfg[0])]
    assert r["gs"] == 1
    assert set(r["grads"]) == set(r["names"]), set(r["names"]) ^ set(r["grads"])
    assert (r["logits"] - r["logits_o"]).abs().max().item() < tol["logits"]
    assert (r["alphas"] - r["alphas_o"]).abs().max().item() < tol["alphas"]
    assert abs(r["loss"] - r["loss_o"]) < tol["loss"] * max(1.0, abs(r["loss_o"]))
    for n, err in grad_errors(r).items():
        assert err < tol["grad"], (n, err)
    if prec == "f32":
        for n in r["names"]:
            # Adam's first step moves every weight by ~lr*sign(g): compare the update, not the weight
            assert (r["params"][n] - r["newp"][n]).abs().max().item() < 2e-4, n


def test_greedy_inference_matches_oracle():
    from las import layers as L
    from las import variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=1, embedding_size=32,
                     attention_size=32, convert_rate=0.3)
    xs, _ = synthetic_batch(4, 30, 8, args.vocab_size, seed=5)
    for cell in ("rnn", "lstm"):
        p0 = O.init_params(args, seed=5, cell=cell)
        with torch.no_grad():
            lo, yo = O.greedy_inference((torch.tensor(xs[0]), xs[1]), O.to_torch(p0), args, cell)
        L.set_cell(cell)
        L.set_precision("f32")
        st = V.reset_default_store(device="cuda")
        st.load(p0)
        las = LAS(args, Listener, Speller, {})
        logits, y_hat = las.inference(xs)
        assert logits.shape == lo.shape
        assert (logits.cpu() - lo).abs().max().item() < 5e-4
        assert torch.equal(y_hat.cpu(), yo)


@pytest.mark.parametrize("apply_bn,cell", [(False, "rnn"), (True, "lstm")])
def test_cnn_listener_train_step_matches_oracle(apply_bn, cell):
    """L4: the reference's default encoder (enc_type='cnn', las/layers.py:118-163) end to end."""
    from las import layers as L
    from las import variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_type="cnn", enc_units=64, num_enc_layers=2, num_enc_channels=8, dec_units=64, num_dec_layers=1,
                     embedding_size=32, attention_size=32, apply_bn=apply_bn, lr=1e-3)
    xs, ys = synthetic_batch(4, 45, 8, 30, seed=3)
    p0 = O.init_params(args, seed=13, cell=cell, enc_type="cnn")
    po = O.to_torch(p0, requires_grad=True)
    z = {k: torch.zeros_like(v) for k, v in po.items()}
    loss_o, logits_o, alphas_o, g_o, newp, _, _ = O.train_step(po, z, {k: torch.zeros_like(v) for k, v in po.items()}, 0,
                                                                (torch.tensor(xs[0]), xs[1]), (torch.tensor(ys[0]), ys[1]), args, cell)
    L.set_cell(cell); L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, {})
    loss, _, gs, logits, alphas, _, _ = las.train(xs, ys)
    assert set(st.order) == set(p0), set(st.order) ^ set(p0)
    assert logits.shape == logits_o.shape and alphas.shape[-1] == 12          # T 45 -> 23 -> 12
    assert (logits.cpu() - logits_o).abs().max().item() < 1e-3
    assert abs(float(loss) - float(loss_o)) < 1e-4
    for n in sorted(p0):
        go, g = g_o[n], st.vars[n].grad.cpu()
        scale = max(go.abs().max().item(), 1e-3)
        assert (g - go).abs().max().item() / scale < 5e-3, n
    assert float(st.buffers["Listener/blstm_0/batch_normalization/moving_mean"].abs().max()) > 0    # UPDATE_OPS ran


def test_dropout_changes_activations_only_in_training():
    from las import layers as L, variables as V
    L.set_cell("rnn"); L.set_precision("f32")
    V.reset_default_store(device="cuda", seed=1)
    x = torch.randn(2, 6, 10, device="cuda")
    (f0, _), _ = L.blstm(x, 16, 0.5, False)
    (f1, _), _ = L.blstm(x, 16, 0.5, False)
    (f2, _), _ = L.blstm(x, 16, 0.5, True)
    assert torch.equal(f0, f1) and not torch.allclose(f0, f2)


def test_training_with_dropout_runs_and_is_masked():
    """dropout_rate > 0 (the reference default 0.5): BLSTM input dropout + embedding dropout are active in training,
    the step stays finite and two steps with different masks differ."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    args = make_args(enc_units=64, num_enc_layers=1, dec_units=64, num_dec_layers=1, embedding_size=32, attention_size=32,
                     dropout_rate=0.3, lr=0.0)
    xs, ys = synthetic_batch(4, 30, 8, 30, seed=2)
    L.set_cell("lstm"); L.set_precision("f32")
    V.reset_default_store(device="cuda", seed=5)
    las = LAS(args, Listener, Speller, {})
    l1 = float(las.train(xs, ys)[0]); l2 = float(las.train(xs, ys)[0])
    assert np.isfinite(l1) and np.isfinite(l2) and l1 != l2          # lr = 0: only the masks changed
    logits, _ = las.inference(xs)
    logits2, _ = las.inference(xs)
    assert torch.equal(logits, logits2)                              # no dropout at inference


def test_weight_shadows_are_rebuilt_by_one_launch_and_match_the_torch_build():
    """las.layers._shadow: the first request builds a bf16 operand copy with torch and registers its recipe; after the
    parameters change (store.shadows cleared, as Adam does) ONE las_build_shadows launch rebuilds every registered copy.
    Both must give the same bits for every recipe shape the Listener uses (transposed / concatenated / padded / fp32 bias)."""
    from las import _hip, layers as L, variables as V
    st = V.reset_default_store(device="cuda", seed=0)
    kfw = st.get("a/kfw", (39 + 64, 256)); kbw = st.get("a/kbw", (39 + 64, 256))
    bfw = st.get("a/bfw", (256,), init="uniform1"); bbw = st.get("a/bbw", (256,), init="uniform1")
    W = st.get("a/dense", (200, 96))
    st.flatten()

    def all_shadows():
        return [L._shadow("ihT", (kfw, kbw), 39, True, 512, 64), L._shadow("ihb", (bfw, bbw), 1, False, 1, 512, bf16=False),
                L._shadow("ih", (kfw, kbw), 39, False, 64, 512), L._shadow("ihT0", (kfw,), 39, True, 256, 64),
                L._shadow("denseT", (W,), 200, True, 96, 256), L._shadow("dense", (W,), 200, False, 256, 128)]

    first = [t.clone() for t in all_shadows()]                       # torch builds
    assert len(st.shadow_recipes) == 6
    with torch.no_grad():
        want = torch.cat((kfw[:39], kbw[:39]), 1).t().to(torch.bfloat16)
    assert torch.equal(first[0][:, :39], want) and first[0][:, 39:].abs().max().item() == 0
    with torch.no_grad():
        st.flat.mul_(-1.5).add_(0.01)                                # "optimiser step"
    st.shadows.clear()
    second = all_shadows()                                           # one fused launch
    st.shadows.clear(); st.shadow_recipes.clear(); st.shadow_table = None
    third = all_shadows()                                            # torch builds of the NEW values
    torch.cuda.synchronize()
    for a, b, c in zip(first, second, third):
        assert a.shape == b.shape == c.shape and b.dtype == c.dtype
        assert torch.equal(b, c)
        assert not torch.equal(a, b)


def test_backward_hand_over_in_chunks_gives_the_same_gradients():
    """Speed-mode Listener backward: between two BPTT sweeps the dX products (recurrent layer's dX with the fused Tanh gradient,
    then the dense layer's dX) run in time chunks from both ends of the sequence, the lower sweep starting after the first chunk
    (las.layers DOUT_CHUNK_ROWS, las_rnn_seq_bwd_db_chunked).  The parameter gradients must equal those of the whole-GEMM
    schedule: the chain values bit for bit (same tiles per row), the weight gradients up to split-K summation order."""
    from las import _hip, layers as L, variables as V
    B, T, F, H, layers = 8, 1100, 39, 256, 2
    assert _hip.rnn_seq_bwd_chunks_ok(1, 1, B, H), "the chunk-aware BPTT kernel must serve the bench's listener configuration"
    assert _hip.streams_overlap(torch.device("cuda", 0))        # (raises by itself when the auxiliary streams cannot overlap)
    used = {}
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, T, F, generator=g).cuda()
    dy = None
    got = {}
    L.set_cell("lstm"); L.set_precision("bf16")
    old = L.DOUT_CHUNK_ROWS
    try:
        for rows in (0, 64):
            L.DOUT_CHUNK_ROWS = rows
            st = V.reset_default_store(device="cuda", seed=3)
            for _ in range(2):                        # first pass creates the variables; the second runs on the flattened store
                y, _, _ = L.pBLSTMLayer(x, [T] * B, layers, H, 0.0, True)
                if st.flat_grad is None:
                    st.flatten()
                    continue
                if dy is None:
                    dy = torch.randn(y.shape, generator=g).cuda() * 0.1
                st.flat_grad.zero_()
                for k in L.VARIANTS:
                    L.VARIANTS[k] = 0
                y.backward(dy)
                _hip.join_side_stream()
            torch.cuda.synchronize()
            _hip.check_status()
            got[rows] = st.flat_grad.clone()
            used[rows] = dict(L.VARIANTS)
    finally:
        L.DOUT_CHUNK_ROWS = old
        L.set_cell("rnn"); L.set_precision("f32")
    a, b = got[0], got[64]
    # the two runs really were two schedules: whole GEMMs between the sweeps vs the CH = true BPTT instance behind chunked products
    assert used[0]["dout_chunks"] == 0 and (used[64]["sweeps_bwd"], used[64]["dout_chunks"]) == (layers + 1, layers), used   # (every BPTT sweep but the top one)
    assert torch.isfinite(a).all() and a.abs().max().item() > 0
    assert (a - b).abs().max().item() <= 1e-3 * max(1.0, a.abs().max().item())
