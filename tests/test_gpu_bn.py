"""tf.layers.batch_normalization (+ the ReLU behind it) in training mode through las_bn_relu_fwd / las_bn_relu_bwd (csrc/bn.hip; reference
las/layers.py:114-116,155-161) against torch's batch_norm + relu on the same tensors: output, input / gamma / beta gradients, the moving
statistics' update, at the run.sh recipe's size ([48 x 319, 512]), at ragged sizes and on a 4-D NHWC block; and the layer-level switch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,relu", [((15312, 512), True), ((37, 64), True), ((1276, 512), False), ((5, 9, 7, 8), True), ((300, 132), True)])
def test_bn_relu_kernels_match_torch(shape, relu):
    from las.layers import _BNReLU
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[-1]
    x = (torch.randn(*shape, generator=g) * 0.7 + 0.3).cuda()
    x2 = x.reshape(-1, C).contiguous().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).cuda().requires_grad_(True)
    mm, mv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y = _BNReLU.apply(x2, gamma, beta, mm, mv, relu)
    w = torch.randn(x2.shape, generator=g).cuda()
    dx, dg, db = torch.autograd.grad((y * w).sum(), (x2, gamma, beta))
    xr = x2.detach().clone().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    yr = torch.nn.functional.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.01, eps=1e-3)
    if relu:
        yr = torch.relu(yr)
    dxr, dgr, dbr = torch.autograd.grad((yr * w).sum(), (xr, gr, br))
    assert (y - yr).abs().max().item() < 2e-5
    if relu:
        assert float(y.min()) >= 0.0
    # (a handful of outputs within rounding of the ReLU's kink may fall on the other side: compare the gradients where both masks agree)
    same = ((y > 0) == (yr > 0)) if relu else torch.ones_like(y, dtype=torch.bool)
    assert float((~same).float().mean()) < 1e-5
    scale = dxr.abs().max().item()
    assert ((dx - dxr) * same).abs().max().item() < 2e-4 * scale + 1e-6
    assert (dg - dgr).abs().max().item() < 2e-4 * max(1.0, dgr.abs().max().item())
    assert (db - dbr).abs().max().item() < 2e-4 * max(1.0, dbr.abs().max().item())
    assert (mm - rm).abs().max().item() < 1e-6 and (mv - rv).abs().max().item() < 1e-5


def test_cnn_listener_step_equals_the_torch_batch_norm_path():
    """the layer-level switch: one train step of the CNN listener (apply_bn on: conv2d bn + two bn per recurrent layer) with the kernels ==
    the same step through torch's batch norm (LAS_NO_BN_KERNEL), parity mode"""
    from helpers import make_args, synthetic_batch
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_type="cnn", enc_units=64, num_enc_layers=2, num_enc_channels=8, dec_units=64, num_dec_layers=1, embedding_size=32,
                     attention_size=32, apply_bn=True, lr=1e-3)
    xs, ys = synthetic_batch(4, 45, 8, 30, seed=3)
    p0 = O.init_params(args, seed=13, cell="lstm", enc_type="cnn")
    out = {}
    for on in (True, False):
        saved = L.BN_KERNEL
        L.BN_KERNEL = on
        try:
            L.set_cell("lstm"); L.set_precision("f32")
            st = V.reset_default_store(device="cuda"); st.load(p0)
            las = LAS(args, Listener, Speller, {})
            loss = float(las.train(xs, ys)[0])
            torch.cuda.synchronize()
            out[on] = (loss, st.flat_grad.clone(), {k: v.clone() for k, v in st.buffers.items()})
        finally:
            L.BN_KERNEL = saved
    assert abs(out[True][0] - out[False][0]) < 1e-5
    g1, g0 = out[True][1], out[False][1]
    assert (g1 - g0).abs().max().item() < 5e-3 * g0.abs().max().item()             # (ReLU-kink flips: see tests/test_gpu_run_sh_recipe.py)
    for k in out[False][2]:
        assert (out[True][2][k] - out[False][2][k]).abs().max().item() < 1e-5, k
