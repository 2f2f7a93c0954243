"""Input dropout of a bidirectional layer in one launch (las_dropout_pair_fwd / _bwd; reference las/layers.py:37-47: fw_cell and bw_cell in
separate DropoutWrappers, input_keep_prob = 1 - dropout_rate, default --dropout_rate 0.5, las/arguments.py:76-78).  tf.nn.dropout's Philox
stream is not contractual (the reference fixes no seed): what is checked is the distribution, the scaling, the independence of the two
directions' masks, the zero padding, reproducibility under a seed, that backward applies EXACTLY the forward's masks, and a train step."""
import numpy as np
import pytest
import torch

from helpers import make_args, synthetic_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,out_bf16,K,ld", [(torch.float32, True, 39, 64), (torch.bfloat16, True, 512, 512), (torch.float32, False, 39, 40),
                                                 (torch.bfloat16, True, 100, 128), (torch.float32, False, 128, 128)])
@pytest.mark.parametrize("keep", [0.5, 0.9])
def test_dropout_pair_distribution_padding_and_backward(dtype, out_bf16, K, ld, keep):
    from las.layers import _DropoutPair
    g = torch.Generator().manual_seed(K + int(keep * 10))
    x = (torch.randn(7, 301, K, generator=g) + 3.0).to(dtype).cuda().requires_grad_(True)        # (no zeros in the input: a zero output IS a dropped element)
    yf, yb = _DropoutPair.apply(x, keep, 12345, out_bf16, ld)
    assert yf.shape == (7, 301, ld) and yf.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    if ld > K:
        assert float(yf[..., K:].abs().max()) == 0 and float(yb[..., K:].abs().max()) == 0
    mf, mb = (yf[..., :K] != 0), (yb[..., :K] != 0)
    n = mf.numel()
    for m in (mf, mb):
        frac = m.float().mean().item()
        assert abs(frac - keep) < 5 * np.sqrt(keep * (1 - keep) / n) + 2e-5, frac          # (+ the 16-bit threshold's resolution)
    both = (mf & mb).float().mean().item()
    assert abs(both - keep * keep) < 5 * np.sqrt(0.25 / n) + 1e-4, both                     # the two directions' masks are independent
    # kept elements are x / keep (in the output's precision)
    ref = (x.detach().float() / keep)
    ref = ref.to(torch.bfloat16).float() if out_bf16 else ref
    assert torch.equal(torch.where(mf, yf[..., :K].float(), ref), ref) and torch.equal(torch.where(mb, yb[..., :K].float(), ref), ref)
    # same seed -> same masks; another seed -> other masks
    yf2, yb2 = _DropoutPair.apply(x, keep, 12345, out_bf16, ld)
    assert torch.equal(yf, yf2) and torch.equal(yb, yb2)
    yf3, _ = _DropoutPair.apply(x, keep, 12346, out_bf16, ld)
    assert not torch.equal(yf, yf3)
    # backward: dx = (m_fw g_fw + m_bw g_bw) / keep with the forward's masks
    gf = torch.randn(7, 301, ld, generator=g).to(yf.dtype).cuda()
    gb = torch.randn(7, 301, ld, generator=g).to(yf.dtype).cuda()
    (dx,) = torch.autograd.grad((yf, yb), x, (gf, gb))
    want = (mf.float() * gf[..., :K].float() + mb.float() * gb[..., :K].float()) / keep
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-6
    assert dx.dtype == dtype and (dx.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_train_step_at_the_reference_default_dropout_rate(prec):
    """--dropout_rate 0.5 (the reference's default): the listener's recurrent layers draw their two masks in one launch per layer; the step
    trains (finite loss, every gradient finite and non-zero), is reproducible under torch.manual_seed, and differs from the no-dropout step."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_units=128, num_enc_layers=2, dec_units=128, num_dec_layers=1, embedding_size=64, attention_size=64, dropout_rate=0.5, lr=1e-3)
    xs, ys = synthetic_batch(6, 64, 12, 30, seed=2, min_frac=0.8)
    p0 = O.init_params(args, seed=1, cell="lstm")

    def step(rate, seed):
        args.dropout_rate = rate
        L.set_cell("lstm"); L.set_precision(prec)
        st = V.reset_default_store(device="cuda"); st.load(p0)
        torch.manual_seed(seed)
        las = LAS(args, Listener, Speller, {})
        loss = float(las.train(xs, ys)[0])
        torch.cuda.synchronize()
        las.check_status()
        return loss, st.flat_grad.clone()

    l1, g1 = step(0.5, 7)
    l2, g2 = step(0.5, 7)
    l3, g3 = step(0.5, 8)
    l0, g0 = step(0.0, 7)
    assert np.isfinite(l1) and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    assert l1 == l2 and torch.equal(g1, g2)
    assert l1 != l3 and l1 != l0
