"""R1 gate kernel with the one-hot input's row look-up folded in (las_lstm_pointwise_rows): bitwise the same as adding the
looked-up input rows to z first (what CharRNN.step_fused did with index_select + add before the decode trace of round 3)."""
import pytest
import torch

import helpers  # noqa: F401  (sys.path)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,V,shift", [(256, 512, 28, 2), (5, 24, 7, 0), (33, 64, 30, 2)])
def test_pointwise_rows_equals_lookup_add_then_pointwise(N, H, V, shift):
    from las import _hip
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(N + H)
    z = torch.randn(N, 4 * H, generator=g).to(dev)
    xrows = torch.randn(V, 4 * H, generator=g).to(dev)
    c_prev = torch.randn(N, H, generator=g).to(dev)
    ids = torch.randint(0, V + shift, (N,), generator=g).to(torch.int32).to(dev)       # ids < shift clamp to row 0 (SOS -> LM id 0)
    lib = _hip.lib()
    c0, h0, c1, h1 = (torch.empty(N, H, device=dev) for _ in range(4))
    z_ref = z + xrows.index_select(0, (ids.long() - shift).clamp_min(0))
    _hip.check(lib.las_lstm_pointwise(_hip.p(z_ref), _hip.p(c_prev), N, H, 0.0, _hip.p(c0), _hip.p(h0), _hip.stream()), "las_lstm_pointwise")
    _hip.check(lib.las_lstm_pointwise_rows(_hip.p(z), _hip.p(xrows), _hip.p(ids), shift, _hip.p(c_prev), N, H, 0.0, _hip.p(c1), _hip.p(h1),
                                           _hip.stream()), "las_lstm_pointwise_rows")
    assert torch.equal(c0, c1) and torch.equal(h0, h1)
    # and against plain torch fp32 (BasicLSTMCell gate order i, j, f, o; forget_bias 0 -- lang/char_rnn_model.py:57-66)
    i, j, f, o = z_ref.split(H, 1)
    c_t = c_prev * torch.sigmoid(f) + torch.sigmoid(i) * torch.tanh(j)
    assert (c1 - c_t).abs().max() < 1e-5 and (h1 - torch.tanh(c_t) * torch.sigmoid(o)).abs().max() < 1e-5


# (M >= 384: the 128-row workgroups of round 5 -- lstm_cell_rows_big_body -- incl. a ragged last row block and K chunks that end inside a part)
@pytest.mark.parametrize("M,I,H,onehot", [(256, 512, 512, False), (256, 0, 512, True), (5, 64, 32, False), (33, 0, 64, True), (70, 1024, 96, False),
                                          (1024, 512, 512, False), (1024, 0, 512, True), (390, 1024, 96, False), (700, 160, 64, False)])
def test_lstm_cell_rows_matches_bf16_operand_reference(M, I, H, onehot):
    """las_lstm_cell_rows (the LM step of the beam search in one launch per layer) against torch: operands rounded to bf16, fp32
    accumulation, then the BasicLSTMCell gate math; and against the two-product + gate-kernel path it replaces."""
    from las import _hip
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(M * 7 + H)
    V, shift = 28, 2
    kern = (torch.randn((V if onehot else I) + H, 4 * H, generator=g) * 0.08).to(dev)
    bias = (torch.randn(4 * H, generator=g) * 0.1).to(dev)
    h = torch.randn(M, H, generator=g).to(dev)
    c_prev = torch.randn(M, H, generator=g).to(dev)
    Ix = V if onehot else I
    hh = _hip.skinny_pack(kern, H, 4 * H, row0=Ix)
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    z_ref = rnd(h).double() @ rnd(kern[Ix:]).double() + bias.double()
    lib = _hip.lib()
    c1, h1 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    if onehot:
        ids = torch.randint(0, V + shift, (M,), generator=g).to(torch.int32).to(dev)
        wx = rnd(kern[:V]).contiguous()                                                # CharRNN.fusion_plan rounds the rows once
        z_ref = z_ref + wx.index_select(0, (ids.long() - shift).clamp_min(0)).double()
        _hip.check(lib.las_lstm_cell_rows(None, 0, 0, _hip.p(ids), shift, _hip.p(wx), _hip.p(h), H, None, _hip.p(hh), _hip.p(bias), _hip.p(c_prev),
                                          M, H, 0.0, _hip.p(c1), _hip.p(h1), _hip.stream()), "las_lstm_cell_rows")
    else:
        x = torch.randn(M, I, generator=g).to(dev)
        ih = _hip.skinny_pack(kern, I, 4 * H)
        z_ref = z_ref + rnd(x).double() @ rnd(kern[:I]).double()
        _hip.check(lib.las_lstm_cell_rows(_hip.p(x), I, I, None, 0, None, _hip.p(h), H, _hip.p(ih), _hip.p(hh), _hip.p(bias), _hip.p(c_prev),
                                          M, H, 0.0, _hip.p(c1), _hip.p(h1), _hip.stream()), "las_lstm_cell_rows")
    i, j, f, o = z_ref.split(H, 1)
    c_t = c_prev.double() * torch.sigmoid(f) + torch.sigmoid(i) * torch.tanh(j)
    h_t = torch.tanh(c_t) * torch.sigmoid(o)
    assert (c1.double() - c_t).abs().max() < 2e-5 and (h1.double() - h_t).abs().max() < 2e-5
    # bad arguments are refused before any launch
    assert lib.las_lstm_cell_rows(None, 0, 0, None, 0, None, _hip.p(h), H, None, _hip.p(hh), _hip.p(bias), _hip.p(c_prev), M, H, 0.0, _hip.p(c1),
                                  _hip.p(h1), _hip.stream()) < 0


@pytest.mark.parametrize("M,I,H", [(256, 1152, 512), (40, 160, 64), (1024, 1152, 512), (500, 160, 64)])
def test_lstm_cell_rows_struct_entry_bf16_rows_fast_gates(M, I, H):
    """las_lstm_cell_rows_args with the options the beam search's Speller step uses: x already bf16, no separate h part, the Speller's
    approximated transcendentals (fast), activated gates written -- against the bf16-operand reference (tolerance of the fast
    sigmoid / tanh: 2e-3)."""
    import ctypes
    from las import _hip
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(M + I)
    kern = (torch.randn(I, 4 * H, generator=g) * 0.05).to(dev)
    bias = (torch.randn(4 * H, generator=g) * 0.1).to(dev)
    x = torch.randn(M, I, generator=g).to(dev).to(torch.bfloat16).contiguous()
    c_prev = torch.randn(M, H, generator=g).to(dev)
    packed = _hip.skinny_pack(kern, I, 4 * H)
    c1, h1, gt = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev), torch.empty(M, 4 * H, device=dev)
    a = _hip.LstmCellArgs()
    a.x, a.x_bf16, a.ldx, a.I = x.data_ptr(), 1, I, I
    a.h, a.Wx, a.Wh, a.bias, a.c_prev, a.fb = None, packed.data_ptr(), None, bias.data_ptr(), c_prev.data_ptr(), 1.0
    a.c_out, a.h_out, a.gates_out, a.M, a.H, a.fast = c1.data_ptr(), h1.data_ptr(), gt.data_ptr(), M, H, 1
    _hip.check(_hip.lib().las_lstm_cell_rows_args(ctypes.byref(a), _hip.stream()), "las_lstm_cell_rows_args")
    z = x.double() @ kern.to(torch.bfloat16).double() + bias.double()
    i, j, f, o = z.split(H, 1)
    gi, gj, gf, go = torch.sigmoid(i), torch.tanh(j), torch.sigmoid(f + 1.0), torch.sigmoid(o)
    c_t = c_prev.double() * gf + gi * gj
    assert (c1.double() - c_t).abs().max() < 2e-3 and (h1.double() - torch.tanh(c_t) * go).abs().max() < 2e-3
    assert (gt.double() - torch.cat([gi, gj, gf, go], 1)).abs().max() < 2e-3


@pytest.mark.parametrize("M,onehot", [(384, False), (1024, False), (1024, True), (500, False)])
def test_lstm_cell_rows_bf16_state_copies_are_bit_identical(M, onehot):
    """Round 5: from 384 rows on the beam search's LM cells read their operand rows from bf16 copies (x = the layer below's h_out_bf16,
    h = the gathered copy of the step before) instead of converting the fp32 rows while staging them.  The copy is the staging's own
    rounding, so c' and h' must be BIT-identical to the fp32-row launch, and h_out_bf16 must be h' rounded to nearest even.  Below
    384 rows (32-row workgroups) the options are refused."""
    import ctypes
    from las import _hip
    dev, I, H, V, shift = "cuda", 512, 512, 28, 2
    g = torch.Generator(device="cpu").manual_seed(M + 3)
    kern = (torch.randn((V if onehot else I) + H, 4 * H, generator=g) * 0.06).to(dev)
    bias = (torch.randn(4 * H, generator=g) * 0.1).to(dev)
    x = torch.randn(M, I, generator=g).to(dev)
    h = torch.randn(M, H, generator=g).to(dev)
    c_prev = torch.randn(M, H, generator=g).to(dev)
    ids = torch.randint(0, V + shift, (M,), generator=g).to(torch.int32).to(dev)
    Ix = V if onehot else I
    hh = _hip.skinny_pack(kern, H, 4 * H, row0=Ix)
    ih = None if onehot else _hip.skinny_pack(kern, I, 4 * H)
    wx = kern[:V].to(torch.bfloat16).to(torch.float32).contiguous()

    def run(copies):
        c1, h1 = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
        hb = torch.zeros(M, H, dtype=torch.bfloat16, device=dev)
        a = _hip.LstmCellArgs()
        xs, hs_ = (x.to(torch.bfloat16).contiguous(), h.to(torch.bfloat16).contiguous()) if copies else (x, h)
        if onehot:
            a.x, a.x_bf16, a.ldx, a.I, a.Wx = None, 0, 0, 0, None
            a.ids, a.id_shift, a.xrows = ids.data_ptr(), shift, wx.data_ptr()
        else:
            a.x, a.x_bf16, a.ldx, a.I, a.Wx = xs.data_ptr(), int(copies), I, I, ih.data_ptr()
            a.ids, a.id_shift, a.xrows = None, 0, None
        a.h, a.h_bf16, a.ldh, a.Wh = hs_.data_ptr(), int(copies), H, hh.data_ptr()
        a.bias, a.c_prev, a.fb = bias.data_ptr(), c_prev.data_ptr(), 0.0
        a.c_out, a.h_out, a.gates_out, a.h_out_bf16, a.M, a.H, a.fast = c1.data_ptr(), h1.data_ptr(), None, hb.data_ptr(), M, H, 0
        rc = _hip.lib().las_lstm_cell_rows_args(ctypes.byref(a), _hip.stream())
        torch.cuda.synchronize()
        return rc, c1, h1, hb, (xs, hs_)

    rc0, c0, h0, hb0, _ = run(False)
    rc1, c1, h1, hb1, _ = run(True)
    assert rc0 == 0 and rc1 == 0
    assert torch.equal(c0, c1) and torch.equal(h0, h1)
    assert torch.equal(hb0, h0.to(torch.bfloat16)) and torch.equal(hb1, hb0)
    # 32-row workgroups do not take the options
    a = _hip.LstmCellArgs()
    hs_ = h[:64].to(torch.bfloat16).contiguous()
    c2, h2 = torch.empty(64, H, device=dev), torch.empty(64, H, device=dev)
    a.x, a.x_bf16, a.ldx, a.I, a.Wx = None, 0, 0, 0, None
    a.ids, a.id_shift, a.xrows = ids.data_ptr(), shift, wx.data_ptr()
    a.h, a.h_bf16, a.ldh, a.Wh = hs_.data_ptr(), 1, H, hh.data_ptr()
    a.bias, a.c_prev, a.fb = bias.data_ptr(), c_prev.data_ptr(), 0.0
    a.c_out, a.h_out, a.gates_out, a.h_out_bf16, a.M, a.H, a.fast = c2.data_ptr(), h2.data_ptr(), None, None, 64, H, 0
    assert _hip.lib().las_lstm_cell_rows_args(ctypes.byref(a), _hip.stream()) < 0
