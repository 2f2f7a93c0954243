"""Speed-mode (bf16) Speller row kernels against the fp32-operand row kernels of the same precision mode.

Both paths contract in bf16 (MFMA step products); the bf16 row kernels additionally read Ws / keys / encoder
rows from bf16 copies and contract the keys gradient after the loop.  The two must agree to bf16 rounding of
those operands (relative 2^-9 per element, averaged down by the contractions)."""
import os

import numpy as np
import pytest
import torch

from helpers import make_args

pytestmark = pytest.mark.gpu


def _run(no_bf_rows, NL, D, A, Hd2, B, Tp, U, mixed):
    from las import layers as L
    from las import variables as V
    from las.las import Speller
    os.environ["LAS_NO_BF_ROWS"] = "1" if no_bf_rows else "0"
    try:
        L.set_cell("lstm")
        L.set_precision("bf16")
        V.reset_default_store(device="cuda", seed=3)
        args = make_args(enc_units=Hd2, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=64,
                         attention_size=A, mode="add", vocab_size=30, enc_type="pblstm")
        sp = Speller(args)
        rng = np.random.RandomState(1)
        enc = torch.tensor(rng.randn(B, Tp, 2 * Hd2).astype(np.float32) * 0.5, device="cuda", requires_grad=True)
        enc_len = rng.randint(Tp // 2, Tp + 1, size=B)
        y = rng.randint(3, 30, size=(B, U))
        coins = np.ones(U, bool)
        sampled = None
        if mixed:
            coins = rng.rand(U) < 0.5
            sampled = rng.randint(3, 30, size=(B, U)).astype(np.int32)
        w = torch.tensor(rng.randn(B, U, 30).astype(np.float32), device="cuda")
        logits, _, alphas = sp(enc, enc_len, U, teacher=y, is_training=True, coins=coins, sampled=sampled)
        (logits * w).sum().backward()
        torch.cuda.synchronize()
        st = V.default_store()
        grads = {n: st.vars[n].grad.detach().cpu().clone() for n in st.order}
        grads["enc"] = enc.grad.detach().cpu().clone()
        return logits.detach().cpu(), alphas.detach().cpu(), grads
    finally:
        os.environ.pop("LAS_NO_BF_ROWS", None)


@pytest.mark.parametrize("shape", [
    # NL, D,  A,   H,  B, Tp, U, mixed sampling
    (1, 512, 128, 256, 5, 37, 9, False),      # the bench geometry at small B / T' / U
    (2, 64, 32, 64, 4, 21, 7, True),          # multi-layer state, sampled tokens (in-loop logits)
    (1, 96, 136, 36, 3, 70, 5, False),        # attention width > 128 (two 16-byte chunks per lane), ragged sizes
    (1, 128, 64, 64, 3, 181, 4, False),       # T' in (160, 192]: 12 frames per wave
    (1, 128, 64, 64, 2, 214, 3, True),        # T' in (192, 224]: 14 frames per wave, 4 frames per 16-lane group
    (1, 64, 32, 32, 2, 230, 3, False),        # T' > 224: generic bf16 row kernels
])
def test_bf16_row_kernels_match_fp32_operand_rows(shape):
    NL, D, A, H, B, Tp, U, mixed = shape
    lo, ao, go = _run(True, NL, D, A, H, B, Tp, U, mixed)
    ln, an, gn = _run(False, NL, D, A, H, B, Tp, U, mixed)
    assert (an - ao).abs().max().item() < 2e-2
    assert (an.sum(-1) - 1).abs().max().item() < 1e-4
    assert (ln - lo).abs().max().item() < 5e-2 * max(1.0, lo.abs().max().item())
    assert set(go) == set(gn)
    for n in sorted(go):
        scale = max(go[n].abs().max().item(), 1e-3)
        err = (gn[n] - go[n]).abs().max().item() / scale
        assert err < 6e-2, (n, err, scale)
