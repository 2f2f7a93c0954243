"""The neural part of the oracle is parity-unpinned (TensorFlow 1.13 is unavailable offline); these checks
anchor it to independent implementations of the same published definitions."""
import math

import numpy as np
import torch

import helpers
from oracle import las_oracle as O


def test_rnn_and_lstm_cells_match_torch_nn():
    torch.manual_seed(0)
    B, T, I, H = 3, 6, 5, 7
    x = torch.randn(B, T, I)
    # rnn: kernel [(I+H),H] -> W_ih^T, W_hh^T
    k = torch.randn(I + H, H) * 0.3
    b = torch.randn(H) * 0.1
    ref = torch.nn.RNN(I, H, batch_first=True)
    with torch.no_grad():
        ref.weight_ih_l0.copy_(k[:I].t()); ref.weight_hh_l0.copy_(k[I:].t())
        ref.bias_ih_l0.copy_(b); ref.bias_hh_l0.zero_()
    assert torch.allclose(O._run_dir(x, k, b, "rnn", False), ref(x)[0], atol=1e-6)
    # lstm: TF gate order i,j,f,o with forget_bias -> torch order i,f,g,o
    k = torch.randn(I + H, 4 * H) * 0.3
    b = torch.randn(4 * H) * 0.1
    ref = torch.nn.LSTM(I, H, batch_first=True)
    i_, j_, f_, o_ = torch.chunk(k, 4, 1)
    bi, bj, bf, bo = torch.chunk(b, 4)
    kt = torch.cat([i_, f_, j_, o_], 1)
    bt = torch.cat([bi, bf + 1.0, bj, bo])
    with torch.no_grad():
        ref.weight_ih_l0.copy_(kt[:I].t()); ref.weight_hh_l0.copy_(kt[I:].t())
        ref.bias_ih_l0.copy_(bt); ref.bias_hh_l0.zero_()
    assert torch.allclose(O._run_dir(x, k, b, "lstm", False), ref(x)[0], atol=1e-6)
    # backward direction = forward on the time-reversed padded block, re-reversed (App. A.4)
    assert torch.allclose(O._run_dir(x, k, b, "lstm", True), O._run_dir(x.flip(1), k, b, "lstm", False).flip(1), atol=1e-6)


def test_adam_clip_and_schedules_closed_forms():
    th, g = torch.tensor([1.0, -2.0]), torch.tensor([0.5, -0.25])
    th1, m1, v1 = O.adam_tf(th, g, torch.zeros(2), torch.zeros(2), 1, 1e-3)
    # first step: m/(sqrt(v)+eps) with bias correction == g/|g| up to eps-hat
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp = th - lr_t * (0.1 * g) / (torch.sqrt(0.001 * g * g) + 1e-8)
    assert torch.allclose(th1, exp, atol=1e-9)
    gs, n = O.clip_by_global_norm([torch.tensor([3.0]), torch.tensor([4.0])], 2.5)
    assert n == 5.0 and torch.allclose(gs[0], torch.tensor([1.5])) and torch.allclose(gs[1], torch.tensor([2.0]))
    gs, _ = O.clip_by_global_norm([torch.tensor([0.3])], 5.0)
    assert torch.allclose(gs[0], torch.tensor([0.3]))
    assert O.scheduled_learning_rate(1e-3, 0) == 1e-3
    assert O.scheduled_learning_rate(1e-3, 150000) == 1e-3 * 0.5
    assert O.scheduled_learning_rate(1e-3, 10 ** 7) == 1e-5
    assert O.scheduled_sampling_rate(0, 100000, 500000, 0.4) == 1.0
    assert abs(O.scheduled_sampling_rate(300000, 100000, 500000, 0.4) - 0.7) < 1e-6
    assert abs(O.scheduled_sampling_rate(900000, 100000, 500000, 0.4) - 0.4) < 1e-6


def test_mask_attend_and_lengths():
    m = O.attention_mask(np.array([2.0, 3.0, 1.0, 0.0]), 3)
    assert m.int().tolist() == [[1, 1, 0], [1, 1, 1], [1, 0, 0], [0, 0, 0]]
    h = torch.randn(2, 4, 3)
    e = torch.randn(2, 4)
    ctx, al = O.attend(h, e, np.array([2, 4]))
    assert torch.all(al[0, 2:] == 0) and abs(float(al[0].sum()) - 1) < 1e-6
    assert torch.allclose(ctx[0], (h[0, :2] * torch.softmax(e[0, :2], 0)[:, None]).sum(0), atol=1e-6)
    # (x + x%2)/2 == ceil(x/2)   (App. A.7)
    for x in range(0, 40):
        assert (x + x % 2) / 2 == math.ceil(x / 2)


def test_hoisted_key_projection_is_exact_restatement():
    """Hoisting dense(hidden) out of the decode loop and preallocating the outputs must
    not change a single bit of the oracle's results."""
    args = helpers.make_args(enc_units=8, num_enc_layers=1, dec_units=12, num_dec_layers=2, embedding_size=6,
                             attention_size=8, mode="loc", loc_kernel_size=5, loc_num_channels=2)
    xs, ys = helpers.synthetic_batch(3, 12, 6, 30, seed=1)
    p = O.to_torch(O.init_params(args, seed=4, cell="lstm"))
    x = torch.tensor(xs[0]).reshape(3, 12, 39)
    with torch.no_grad():
        h, el = O.pblstm_listener(x, xs[1], p, 1, "lstm")
        a1 = O.speller_forward(h, el, 5, p, args, "lstm", teacher=torch.tensor(ys[0]), hoist=True)
        a2 = O.speller_forward(h, el, 5, p, args, "lstm", teacher=torch.tensor(ys[0]), hoist=False)
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])
    assert h.shape == (3, 6, 16) and el.tolist() == [math.ceil(v / 2) for v in xs[1]]


def test_loss_matches_manual_label_smoothing():
    logits = torch.randn(2, 3, 5)
    y = torch.tensor([[1, 4, 0], [2, 0, 0]])
    loss = O.las_loss(logits, y, 5, smooth=True)
    lp = torch.log_softmax(logits, -1)
    tot, n = 0.0, 0
    for b in range(2):
        for t in range(3):
            if y[b, t] != 0:
                soft = torch.full((5,), 0.01 / 5)
                soft[y[b, t]] += 0.99
                tot += float(-(soft * lp[b, t]).sum()); n += 1
    assert abs(float(loss) - tot / n) < 1e-6


def test_bf16_operand_mode_of_the_oracle():
    """oracle.set_precision('bf16'): operands rounded to bf16 (RNE), fp32 accumulation, bf16-operand gradient products,
    straight-through rounding; 'f32' restores exact arithmetic."""
    import numpy as np
    import torch
    from oracle import las_oracle as O
    g = torch.Generator().manual_seed(0)
    a = torch.randn(3, 5, 8, generator=g, requires_grad=True)
    w = torch.randn(8, 4, generator=g, requires_grad=True)
    r = torch.randn(3, 5, 4, generator=g)
    O.set_precision("bf16")
    try:
        y = O._mm(a, w)
        (y * r).sum().backward()
    finally:
        O.set_precision("f32")
    bf = lambda t: t.detach().to(torch.bfloat16).to(torch.float32)
    assert torch.equal(y.detach(), bf(a) @ bf(w))
    assert torch.allclose(a.grad, bf(r) @ bf(w).t(), atol=1e-6)
    assert torch.allclose(w.grad, bf(a).reshape(-1, 8).t() @ bf(r).reshape(-1, 4), atol=1e-5)
    assert torch.equal(O._mm(a, w).detach(), a.detach() @ w.detach())          # back in f32 mode
    # end to end: the bf16-mode train step stays close to the f32 one (operand rounding only), and differs from it
    import helpers
    args = helpers.make_args(enc_units=16, num_enc_layers=1, dec_units=16, num_dec_layers=1, embedding_size=8, attention_size=8)
    xs, ys = helpers.synthetic_batch(3, 12, 5, 30, seed=1)
    outs = {}
    for mode in ("f32", "bf16"):
        O.set_precision(mode)
        try:
            p = O.to_torch(O.init_params(args, seed=2, cell="lstm"), requires_grad=True)
            z = {k: torch.zeros_like(v) for k, v in p.items()}
            outs[mode] = O.train_step(p, z, {k: torch.zeros_like(v) for k, v in p.items()}, 0,
                                      (torch.tensor(xs[0]), xs[1]), (torch.tensor(ys[0]), ys[1]), args, "lstm")
        finally:
            O.set_precision("f32")
    d = (outs["f32"][1] - outs["bf16"][1]).abs().max().item()
    assert 1e-6 < d < 5e-2
