"""Speed-mode (bf16) Speller row kernels against the ORACLE's Speller (oracle.speller_forward in its bf16-operand
mode; reference las/las.py:72-160, las/layers.py:199-257): logits, alignments and every gradient, for each row-kernel
family (fully prefetching `pf`, generic bf16 `bf`, fp32-operand rows) and each frames-per-wave instantiation,
T' in (128, 160] -- what bench.py times -- included.

Tolerance: same rounding points on both sides (bf16 operands, fp32 accumulation), so what is left is accumulation
order / fast transcendentals / boundary flips of the bf16 rounding: logits 5e-3 (of max|logit|), alignments 2e-3,
gradients 2e-2 of the largest oracle entry."""
import numpy as np
import pytest
import torch

from helpers import make_args

pytestmark = pytest.mark.gpu


def _run(flags, NL, D, A, Hd2, B, Tp, U, mixed, loc=None):
    from las import _hip, layers as L, variables as V
    from las.las import Speller
    from oracle import las_oracle as O
    _hip.speller_flags = flags
    try:
        L.set_cell("lstm")
        L.set_precision("bf16")
        V.reset_default_store(device="cuda", seed=3)
        args = make_args(enc_units=Hd2, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=64,
                         attention_size=A, mode="add", vocab_size=30, enc_type="pblstm")
        if loc is not None:
            args.mode, args.loc_kernel_size, args.loc_num_channels = "loc", loc[0], loc[1]
        sp = Speller(args)
        rng = np.random.RandomState(1)
        enc_np = rng.randn(B, Tp, 2 * Hd2).astype(np.float32) * 0.5
        enc = torch.tensor(enc_np, device="cuda", requires_grad=True)
        enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
        y = rng.randint(3, 30, size=(B, U))
        coins = np.ones(U, bool)
        sampled = None
        if mixed:
            coins = rng.rand(U) < 0.5
            sampled = rng.randint(3, 30, size=(B, U)).astype(np.int32)
        w = torch.tensor(rng.randn(B, U, 30).astype(np.float32))
        logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True, coins=coins, sampled=sampled)
        (logits * w.cuda()).sum().backward()
        _hip.join_side_stream()
        torch.cuda.synchronize()
        st = V.default_store()
        grads = {n: st.vars[n].grad.detach().cpu().clone() for n in st.order}
        grads["enc"] = enc.grad.detach().cpu().clone()
        # oracle, bf16-operand mode; fp32-operand rows (flag 2) keep query / keys / context in fp32
        p0 = {n: st.vars[n].detach().cpu().numpy() for n in st.order}
        from helpers import loc_loop_eligible, wide_eligible
        bf_rows = not (flags & 2) and (loc is None or loc_loop_eligible(args, B, Tp, U) or wide_eligible(args, U))
        O.set_precision("bf16", "bf" if bf_rows else "f32")
        try:
            po = O.to_torch(p0, requires_grad=True)
            enc_o = torch.tensor(enc_np, requires_grad=True)
            lo, ao = O.speller_forward(enc_o, enc_len.astype(np.float64), U, po, args, "lstm", teacher=torch.tensor(y),
                                       is_training=True, coins=coins, sampled=None if sampled is None else torch.tensor(sampled))
            (lo * w).sum().backward()
        finally:
            O.set_precision("f32")
        go = {n: po[n].grad for n in po if po[n].grad is not None}
        go["enc"] = enc_o.grad
        return logits.detach().cpu(), alphas.detach().cpu(), grads, lo.detach(), ao.detach(), go
    finally:
        _hip.speller_flags = 0


SHAPES = [
    # NL, D,  A,   H,  B, Tp, U, mixed sampling
    (1, 512, 128, 256, 5, 37, 9, False),      # the bench geometry at small B / T' / U: pf<.,8>
    (1, 512, 128, 256, 4, 160, 6, False),     # the bench geometry at the bench T' = 160: pf<.,10>
    (1, 512, 128, 256, 3, 131, 5, True),      # T' in (128, 160], ragged, sampled tokens
    (2, 64, 32, 64, 4, 21, 7, True),          # multi-layer state (generic bf rows)
    (1, 96, 136, 36, 3, 70, 5, False),        # attention width > 128 (two 16-byte chunks per lane), ragged sizes
    (1, 128, 64, 64, 3, 181, 4, False),       # T' in (160, 192]: pf<.,12>
    (1, 128, 64, 64, 2, 214, 4, True),        # T' in (192, 224]: pf<.,14>
    (1, 64, 32, 32, 2, 230, 3, False),        # T' > 224: generic bf16 row kernels
    # the geometry bench.py times (D = 512, A = 128, Hd = 512, T' = 160) with MORE THAN ONE utterance per XCD group of the
    # one-launch loop kernels: utterance b = 8 r + x is tile row r of group x, so B = 9 / 17 / 48 exercise Rx = 2 / 3 / 6 row
    # workgroups per group (ragged: some groups one row short), the `r16 < Rx` masking and the (j - pn) * 8 + x indexing
    (1, 512, 128, 256, 9, 160, 6, False),
    (1, 512, 128, 256, 17, 160, 6, True),
    (1, 512, 128, 256, 48, 160, 6, False),
    (1, 512, 128, 256, 48, 160, 5, True),
    # the B = 96 / T = 638 bucket (T' = 80): R = 12 row workgroups per group
    (1, 512, 128, 256, 96, 80, 5, False),
]


@pytest.mark.parametrize("flags", [0, 1, 2, 4])         # default (one launch per step where eligible) / no pf rows / fp32-operand rows / pf rows, two launches per step
@pytest.mark.parametrize("shape", SHAPES)
def test_bf16_row_kernels_match_oracle(shape, flags):
    NL, D, A, H, B, Tp, U, mixed = shape
    ln, an, gn, lo, ao, go = _run(flags, NL, D, A, H, B, Tp, U, mixed)
    assert (an - ao).abs().max().item() < 2e-3
    assert (an.sum(-1) - 1).abs().max().item() < 1e-4
    assert (ln - lo).abs().max().item() < 5e-3 * max(1.0, lo.abs().max().item())
    assert set(go) <= set(gn)
    for n in sorted(go):
        scale = max(go[n].abs().max().item(), 1e-3)
        err = (gn[n] - go[n]).abs().max().item() / scale
        assert err < 2e-2, (n, err, scale)


def test_embedding_gradient_buckets_with_more_than_32k_positions():
    """The embedding gradient's token buckets (emb_hist / emb_place / emb_reduce) keep the token list in LDS: at B U = 35,200
    positions that is more than the 64 KB a launch gets without asking (a stacked B = 192 step has 36,672)."""
    ln, an, gn, lo, ao, go = _run(0, 1, 64, 32, 32, 176, 20, 200, False)
    n = "embedding/embedding_matrix"
    scale = max(go[n].abs().max().item(), 1e-3)
    assert (gn[n] - go[n]).abs().max().item() / scale < 2e-2
    assert (ln - lo).abs().max().item() < 5e-3 * max(1.0, lo.abs().max().item())


LOC_SHAPES = [
    # D,  A,   H,  B, Tp, U, mixed, (Kc, C)      -- location-aware attention in the one-launch loop kernels (round 3)
    (512, 128, 256, 5, 37, 9, False, (201, 10)),     # reference defaults K = 201, C = 10: both borders of the filter clipped
    (512, 128, 256, 48, 160, 6, False, (201, 10)),   # the bench geometry, 6 row workgroups per XCD group
    (512, 128, 256, 17, 131, 5, True, (201, 10)),    # ragged T', sampled tokens
    (128, 64, 64, 9, 181, 4, True, (7, 3)),          # small filter, T' in (160, 192]
    (128, 128, 64, 3, 214, 4, False, (31, 10)),      # T' in (192, 224]
]


@pytest.mark.parametrize("shape", LOC_SHAPES)
def test_location_aware_loop_kernels_match_oracle(shape):
    """dec_loop_{fwd,bwd}_kernel<., ., LOC = true> (conv1d over the previous alignment from LDS, f . Wf in the energies, d f and
    d alpha_{t-1} in the gradient loop; keys / Wf / filter gradients contracted after the loop by dkeys_loc_kernel / dlocw_kernel)
    against the oracle's LocationAwareAttention (reference las/layers.py:259-311) in its bf16-row mode: logits, alignments, every
    gradient including conv1d/kernel, conv1d/bias and dense_2/kernel."""
    from helpers import loc_loop_eligible
    D, A, H, B, Tp, U, mixed, loc = shape
    ln, an, gn, lo, ao, go = _run(0, 1, D, A, H, B, Tp, U, mixed, loc=loc)
    assert (an - ao).abs().max().item() < 2e-3
    assert (ln - lo).abs().max().item() < 5e-3 * max(1.0, lo.abs().max().item())
    for n in sorted(go):
        scale = max(go[n].abs().max().item(), 1e-3)
        err = (gn[n] - go[n]).abs().max().item() / scale
        assert err < 2e-2, (n, err, scale)
    for n in ("Speller/decode/attention/conv1d/kernel", "Speller/decode/attention/conv1d/bias", "Speller/decode/attention/dense_2/kernel"):
        assert float(go[n].abs().max()) > 0 and n in gn


@pytest.mark.parametrize("shape", [LOC_SHAPES[0], LOC_SHAPES[2], LOC_SHAPES[3]])
def test_location_aware_gradient_loop_with_and_without_the_saved_activations(shape):
    """The location-aware forward rows hand tanh(keys + q + f . Wf) (fp16) and the conv outputs f to the gradient loop
    (las_speller_fwd_args.act_save); without the buffer (LAS_SPELLER_SAVE_ACT=0, or a caller of the C ABI that passes NULL) the gradient rows
    recompute both and run the transposed conv at the end of the iteration.  Same forward bit for bit, gradients equal up to the fp16 rounding of
    the activations (|error| <= 2^-12 on values in (-1, 1)) -- and both are held against the oracle above / here."""
    from las import las as LL
    D, A, H, B, Tp, U, mixed, loc = shape
    assert LL.SAVE_ACTIVATIONS, "the default is to save"
    ln, an, gn, lo, ao, go = _run(0, 1, D, A, H, B, Tp, U, mixed, loc=loc)
    LL.SAVE_ACTIVATIONS = False
    try:
        ln2, an2, gn2, _, _, _ = _run(0, 1, D, A, H, B, Tp, U, mixed, loc=loc)
    finally:
        LL.SAVE_ACTIVATIONS = True
    assert torch.equal(ln, ln2) and torch.equal(an, an2)
    differ = 0
    for n in sorted(go):
        scale = max(go[n].abs().max().item(), 1e-3)
        assert (gn[n] - gn2[n]).abs().max().item() / scale < 1e-2, n        # (measured: up to 4.4e-3 on conv1d/bias)
        assert (gn2[n] - go[n]).abs().max().item() / scale < 2e-2, n
        differ += int(not torch.equal(gn[n], gn2[n]))
    assert differ > 0, "the two gradient loops are different kernels paths: identical bits mean the switch did nothing"


def test_backward_reuses_the_forward_operand_copies_only_when_nobody_else_used_the_workspace():
    """_SpellerLoop.backward passes LAS_SPELLER_REUSE_PREP when no other Speller call asked for the workspace since its own
    forward (the bf16 copies of enc / keys / Ws are still there).  An interleaved call on OTHER inputs must switch the reuse
    off: the gradients of the first call have to be the same either way."""
    from las import _hip, layers as L, variables as V
    from las.las import Speller
    L.set_cell("lstm"); L.set_precision("bf16")
    grads = []
    for interleave in (False, True):
        V.reset_default_store(device="cuda", seed=3)
        args = make_args(enc_units=64, num_enc_layers=2, dec_units=128, num_dec_layers=1, embedding_size=64, attention_size=64,
                         mode="add", vocab_size=30, enc_type="pblstm")
        sp = Speller(args)
        rng = np.random.RandomState(5)
        enc = torch.tensor(rng.randn(6, 40, 128).astype(np.float32) * 0.5, device="cuda", requires_grad=True)
        enc2 = torch.tensor(rng.randn(6, 40, 128).astype(np.float32) * 0.5, device="cuda")
        enc_len = rng.randint(20, 41, size=6)
        y = rng.randint(3, 30, size=(6, 9))
        w = torch.tensor(rng.randn(6, 9, 30).astype(np.float32)).cuda()
        logits, _, _ = sp(enc, enc_len, 9, teacher=y, is_training=True)
        if interleave:
            with torch.no_grad():
                sp(enc2, enc_len, 9, teacher=y, is_training=True)        # overwrites the workspace's operand copies
        (logits * w).sum().backward()
        _hip.join_side_stream()
        torch.cuda.synchronize()
        st = V.default_store()
        grads.append([enc.grad.cpu().clone()] + [st.vars[n].grad.detach().cpu().clone() for n in st.order])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_loop_kernel_poll_timeout_is_reported_and_the_launch_drains():
    """the one-launch loop kernels used to __builtin_trap() (abort the context) when a partner workgroup was not
    seen within the poll bound.  Now the wave stores LAS_SPELLER_STATUS_TIMEOUT in the status word and ends, the launch drains in
    bounded time, and the host raises at its next status check -- forced here with a poll budget of 2 (LAS_SPELLER_SPIN_LOG2(1)),
    in a child process so that a regression can only cost the child."""
    import subprocess, sys, os
    from helpers import PKG, ROOT
    script = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch
from helpers import make_args
from las import _hip, layers as L, variables as V
from las.las import Speller
L.set_cell("lstm"); L.set_precision("bf16")
V.reset_default_store(device="cuda", seed=3)
args = make_args(enc_units=256, num_enc_layers=2, dec_units=512, num_dec_layers=1, embedding_size=64, attention_size=128, mode="add", vocab_size=30)
sp = Speller(args)
rng = np.random.RandomState(1)
enc = torch.tensor(rng.randn(48, 160, 512).astype(np.float32) * 0.5, device="cuda", requires_grad=True)
y = rng.randint(3, 30, size=(48, 12))
_hip.speller_flags = _hip.speller_spin_log2(1)
logits, _, _ = sp(enc, np.full(48, 160), 12, teacher=y, is_training=True)
logits.sum().backward()
_hip.join_side_stream()
torch.cuda.synchronize()
try:
    _hip.check_status()
    print("NO_ERROR")
except RuntimeError as e:
    print("RAISED", "status 3" in str(e))
_hip.speller_flags = 0
V.default_store().zero_grad()
logits, _, _ = sp(enc, np.full(48, 160), 12, teacher=y, is_training=True)      # and the normal budget works right after
logits.sum().backward()
_hip.join_side_stream()
torch.cuda.synchronize()
_hip.check_status()
print("CLEAN", bool(torch.isfinite(logits).all()))
''' % (PKG, ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "RAISED True" in r.stdout and "CLEAN True" in r.stdout, r.stdout[-2000:]
