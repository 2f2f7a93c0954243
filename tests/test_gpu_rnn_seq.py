"""K2/K2b parity: the persistent recurrent sweep (C ABI las_rnn_seq_fwd/bwd) vs the oracle's
bidirectional_dynamic_rnn restatement (oracle.las_oracle._run_dir, reference las/layers.py:28-54)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (cell, prec, B, T, H)
CASES = [
    ("rnn", 0, 5, 7, 48), ("lstm", 0, 5, 7, 48), ("rnn", 0, 9, 12, 100), ("lstm", 0, 17, 9, 64),
    ("rnn", 1, 5, 7, 64), ("lstm", 1, 5, 7, 64), ("rnn", 1, 20, 33, 128), ("lstm", 1, 20, 33, 128),
    ("rnn", 1, 48, 40, 256), ("lstm", 1, 48, 40, 256), ("lstm", 0, 48, 20, 256), ("rnn", 1, 3, 5, 512), ("lstm", 1, 37, 23, 256), ("lstm", 1, 3, 1, 256),
    # parity mode on the clustered exact-fp32 MFMA sweeps (csrc/rnn_seq_f32.hip: H in {64, 128, 256, 512}; P = G H / 64 members per
    # (direction, 16-row tile): 1 ... 32), ragged last tiles, T = 1, and enough steps to go round the two exchange slots many times
    ("rnn", 0, 7, 9, 64), ("rnn", 0, 6, 10, 128), ("rnn", 0, 20, 15, 256), ("rnn", 0, 19, 8, 512),
    ("lstm", 0, 33, 11, 128), ("lstm", 0, 5, 6, 512), ("lstm", 0, 3, 1, 256), ("lstm", 0, 48, 300, 256), ("rnn", 0, 48, 300, 256),
]


def _oracle_sweep(xp, whh, cell, double=True):
    """xp [B,T,2,GH] pre-activations; whh [2][H,GH].  Returns out [B,T,2H] and grads hook inputs."""
    from oracle import las_oracle as O
    dt = torch.float64 if double else torch.float32
    GH = xp.shape[-1]
    outs = []
    for d in range(2):
        kernel = torch.cat([torch.eye(GH, dtype=dt), whh[d].to(dt)], 0)
        outs.append(O._run_dir(xp[:, :, d].to(dt), kernel, torch.zeros(GH, dtype=dt), cell, reverse=bool(d)))
    return torch.cat(outs, -1)


@pytest.mark.parametrize("case", CASES)
def test_rnn_seq_fwd_bwd(case):
    from las import _hip
    cell, prec, B, T, H = case
    G = 4 if cell == "lstm" else 1
    GH = G * H
    g = torch.Generator().manual_seed(B * 131 + T * 7 + H)
    xp = torch.randn(B, T, 2, GH, generator=g) * 0.8
    lim = (6.0 / (H + GH)) ** 0.5 * 1.5
    whh = [(torch.rand(H, GH, generator=g) * 2 - 1) * lim for _ in range(2)]
    R = torch.randn(B, T, 2 * H, generator=g)

    xpl = xp.clone().double().requires_grad_(True)
    wl = [w.clone().double().requires_grad_(True) for w in whh]
    ref = _oracle_sweep(xpl, wl, cell)
    (ref * R.double()).sum().backward()

    dev = "cuda"
    c = _hip.CELL_LSTM if cell == "lstm" else _hip.CELL_RNN
    io = _hip.rnn_seq_io_dtype(c, prec, H)    # bf16 storage when the speed-mode MFMA sweeps serve the shape
    if io == torch.bfloat16:                 # the oracle sees the same (bf16-rounded) x-projection and upstream gradient
        xp = xp.to(io).float()
        R = R.to(io).float()
        xpl = xp.clone().double().requires_grad_(True)
        wl = [w.clone().double().requires_grad_(True) for w in whh]
        ref = _oracle_sweep(xpl, wl, cell)
        (ref * R.double()).sum().backward()
    gates = xp.to(dev).to(io).contiguous()
    w0, w1 = whh[0].to(dev), whh[1].to(dev)
    Tpad = T + (T % 2)                       # exercise the pad-frame batch stride
    out = torch.zeros(B, Tpad, 2 * H, device=dev, dtype=io)
    cst = torch.zeros(B, T, 2, H, device=dev, dtype=io) if cell == "lstm" else None
    _hip.rnn_seq_fwd(c, prec, B, T, H, gates, w0, w1, GH, out, 2 * H, Tpad * 2 * H, cst)
    got = out[:, :T].float().cpu().double()
    tol = 2e-5 if prec == 0 else 4e-2
    err = (got - ref.detach()).abs().max().item()
    assert err < tol, ("fwd", case, err)
    if T % 2:
        assert out[:, T:].float().abs().max().item() == 0.0   # pad frame untouched

    dout = torch.zeros(B, Tpad, 2 * H, device=dev, dtype=io)
    dout[:, :T] = R.to(dev).to(io)
    # the sweep also accumulates (+=) the bias gradients = column sums of dZ per direction
    db_fw = torch.full((GH,), 0.5, device=dev)
    db_bw = torch.full((GH,), -0.25, device=dev)
    _hip.rnn_seq_bwd(c, prec, B, T, H, gates, w0, w1, GH, out, 2 * H, Tpad * 2 * H, cst, dout, 2 * H, Tpad * 2 * H,
                     db_fw=db_fw, db_bw=db_bw)
    dg = gates.float().cpu().double()
    for d, (db, init) in enumerate(((db_fw, 0.5), (db_bw, -0.25))):
        want = dg[:, :, d].sum((0, 1)) + init
        errb = (db.cpu().double() - want).abs().max().item()
        # bf16 storage: the sweep sums dZ in fp32 BEFORE rounding the stored copy, `want` sums the rounded copy
        assert errb < (1e-3 if io == torch.float32 else 1e-2) * max(1.0, want.abs().max().item()), ("db", case, d, errb)
    refg = xpl.grad
    err = (dg - refg).abs().max().item()
    scale = refg.abs().max().item()
    tolb = (5e-5 if prec == 0 else 6e-2) * max(1.0, scale)
    assert err < tolb, ("bwd", case, err, scale)


@pytest.mark.parametrize("cell,B,T,H", [("lstm", 21, 40, 256), ("rnn", 21, 40, 256), ("lstm", 9, 17, 64)])
def test_fp32_cluster_sweeps_equal_the_valu_kernels(cell, B, T, H):
    """The parity mode's two kernel families on the same inputs: clustered exact-fp32 MFMA (default) vs the round-1 VALU kernels
    (LAS_SEQ_F32_VALU).  Both are fp32 fma chains; they differ in summation order only (the MFMA form splits K into quarters)."""
    from las import _hip
    c = _hip.CELL_LSTM if cell == "lstm" else _hip.CELL_RNN
    G = 4 if cell == "lstm" else 1
    GH = G * H
    g = torch.Generator().manual_seed(5 + B + H)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda()
    lim = (6.0 / (H + GH)) ** 0.5 * 1.5
    w0, w1 = [((torch.rand(H, GH, generator=g) * 2 - 1) * lim).cuda() for _ in range(2)]
    dout = torch.randn(B, T, 2 * H, generator=g).cuda()
    res = []
    for flags in (0, _hip.SEQ_F32_VALU):
        gates = xp.clone()
        out = torch.zeros(B, T, 2 * H, device="cuda")
        cst = torch.zeros(B, T, 2, H, device="cuda") if cell == "lstm" else None
        _hip.rnn_seq_fwd(c, 0, B, T, H, gates, w0, w1, GH, out, 2 * H, T * 2 * H, cst, flags=flags)
        act = gates.clone()
        _hip.rnn_seq_bwd(c, 0, B, T, H, gates, w0, w1, GH, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H, flags=flags)
        torch.cuda.synchronize()
        _hip.check_status()
        res.append((out, act, gates) + ((cst,) if cst is not None else ()))
    for x, y in zip(*res):
        assert (x - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())


def test_fp32_cluster_sweep_reports_an_exchange_timeout():
    from las import _hip
    B, T, H = 48, 256, 256
    gates = torch.randn(B, T, 2, 4 * H, device="cuda")
    w0 = torch.randn(H, 4 * H, device="cuda") * 0.05
    out = torch.zeros(B, T, 2 * H, device="cuda")
    cst = torch.zeros(B, T, 2, H, device="cuda")
    _hip.check_status()
    _hip.rnn_seq_fwd(1, 0, B, T, H, gates, w0, w0, 4 * H, out, 2 * H, T * 2 * H, cst, flags=_hip.seq_spin_log2(1))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="did not publish"):
        _hip.check_status()
    _hip.check_status()


def test_rnn_seq_rejects_bad_args():
    from las import _hip
    x = torch.zeros(4, device="cuda")
    with pytest.raises(RuntimeError):
        _hip.rnn_seq_fwd(0, 0, 0, 4, 8, x, x, x, 8, x, 16, 0, None)     # B = 0
    with pytest.raises(RuntimeError):
        _hip.rnn_seq_fwd(1, 0, 1, 1, 1, x, x, x, 4, x, 2, 0, None)      # lstm without cstate


def test_exchange_timeout_is_reported_not_silent():
    """A cluster member that never sees its partners' granules (forced here by a spin bound of 2 polls) must surface as
    an error: the sweep stores LAS_SEQ_STATUS_* in the status word and the host raises at its next check."""
    from las import _hip
    B, T, H = 48, 512, 256          # 512 exchanges x 24 workgroups: some gather needs more than 2 polls
    GH = 4 * H
    gates = torch.randn(B, T, 2, GH, device="cuda").to(torch.bfloat16)
    w0 = torch.randn(H, GH, device="cuda") * 0.05
    out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
    cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
    _hip.check_status()                                   # clean before
    _hip.rnn_seq_fwd(1, 1, B, T, H, gates, w0, w0, GH, out, 2 * H, T * 2 * H, cst, flags=_hip.seq_spin_log2(1))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="did not publish"):
        _hip.check_status()
    _hip.check_status()                                   # the word was cleared by the raise
    # and the normal bound works on the same shape
    _hip.rnn_seq_fwd(1, 1, B, T, H, gates, w0, w0, GH, out, 2 * H, T * 2 * H, cst)
    torch.cuda.synchronize()
    _hip.check_status()
    dout = torch.randn(B, T, 2 * H, device="cuda").to(torch.bfloat16)
    _hip.rnn_seq_bwd(1, 1, B, T, H, gates, w0, w0, GH, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H, flags=_hip.seq_spin_log2(1))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="status 2"):
        _hip.check_status()


def test_a_reported_timeout_keeps_the_optimiser_from_using_the_step():
    """a sweep time-out must not reach the parameters.  las_clip_adam checks the status word (and the all-reduced
    guard slot of the gradient bucket) ON THE DEVICE and leaves theta / m / v untouched -- no host synchronisation involved;
    LAS.train wires both in, and the next check_status() still raises."""
    from helpers import make_args, synthetic_batch
    from las import _hip, layers as L, variables as V
    from las.las import LAS, Listener, Speller
    n = 1000
    th = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda")
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    th0 = th.clone()
    sumsq = (g * g).sum().reshape(1)
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    guard = torch.zeros(1, device="cuda")

    applied = torch.zeros(1, dtype=torch.int32, device="cuda")

    def adam():
        _hip.check(_hip.lib().las_clip_adam(_hip.p(th), _hip.p(g), _hip.p(m), _hip.p(v), n, _hip.p(sumsq), 5.0, 1e-3, 0.9, 0.999, 1e-8,
                                            _hip.p(status), _hip.p(guard), _hip.p(applied), _hip.stream()), "las_clip_adam")
        torch.cuda.synchronize()

    status[0] = 2
    adam()
    assert torch.equal(th, th0) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0 and int(applied) == 0
    status[0] = 0; guard[0] = 1.0                                   # another rank's time-out (summed guard slot)
    adam()
    assert torch.equal(th, th0) and float(m.abs().max()) == 0 and int(applied) == 0
    guard[0] = 0.0
    adam()
    assert not torch.equal(th, th0) and float(m.abs().max()) > 0 and int(applied) == 1      # (round 6: the device counts the updates it applied)
    # end to end: a step whose status word is set leaves the store untouched.  With the step recovery off (rounds 1-5) the next check
    # raises and the step after the raise trains again; with it on (round 6, the default) the check RE-RUNS the lost step
    from las import las as LL
    args = make_args(enc_units=64, num_enc_layers=1, dec_units=64, num_dec_layers=1, embedding_size=32, attention_size=32, lr=1e-3)
    L.set_cell("lstm"); L.set_precision("bf16")
    st = V.reset_default_store(device="cuda", seed=4)
    las = LAS(args, Listener, Speller, {})
    xs, ys = synthetic_batch(4, 40, 8, 30, seed=3)
    las.build_variables()
    _hip.check_status()
    before = st.flat.clone()
    saved = LL.RECOVER_STEPS
    LL.RECOVER_STEPS = False
    try:
        _hip.status_word(st.flat.device)[0] = 1                     # as a timed-out forward sweep would leave it
        las.train(xs, ys)
        torch.cuda.synchronize()
        assert torch.equal(st.flat, before) and float(st.adam_m.abs().max()) == 0
        with pytest.raises(RuntimeError, match="status 1"):
            las.check_status()
    finally:
        LL.RECOVER_STEPS = saved
    st = V.reset_default_store(device="cuda", seed=4)
    las = LAS(args, Listener, Speller, {})
    las.build_variables()
    _hip.status_word(st.flat.device)[0] = 1
    las.train(xs, ys)
    torch.cuda.synchronize()
    assert torch.equal(st.flat, before) and float(st.adam_m.abs().max()) == 0 and st.global_step == 1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        las.check_status()                                          # finds the word, re-runs the lost step on the fall-back schedule
    assert las.recovered_steps == 1 and st.global_step == 1 and not torch.equal(st.flat, before) and float(st.adam_m.abs().max()) > 0
    las.train(xs, ys)
    torch.cuda.synchronize()
    las.check_status()
    assert not torch.equal(st.flat, before)


def test_batches_larger_than_the_cu_budget_are_swept_in_row_chunks():
    """every cluster member must be resident (one workgroup per CU): B = 560 rows x 2 directions x P = 4 needs 280 CUs,
    so the sweep runs as two launches over row chunks; results must equal the oracle as for a small batch."""
    from las import _hip
    B, T, H = 560, 5, 256
    GH = 4 * H
    g = torch.Generator().manual_seed(1)
    xp = torch.randn(B, T, 2, GH, generator=g) * 0.8
    whh = [(torch.rand(H, GH, generator=g) * 2 - 1) * 0.05 for _ in range(2)]
    ref = _oracle_sweep(xp, whh, "lstm", double=False)
    gates = xp.cuda().to(torch.bfloat16)
    out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
    cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
    _hip.rnn_seq_fwd(1, 1, B, T, H, gates, whh[0].cuda(), whh[1].cuda(), GH, out, 2 * H, T * 2 * H, cst)
    torch.cuda.synchronize()
    _hip.check_status()
    assert (out.float().cpu() - ref).abs().max().item() < 4e-2


@pytest.mark.parametrize("case", [("lstm", 1, 48, 40, 256), ("lstm", 1, 37, 23, 256), ("lstm", 1, 5, 9, 256), ("lstm", 1, 20, 33, 128)])
def test_rows16_flag_gives_the_same_sweep(case):
    """The clustered sweeps default to 8-row batch tiles (duplicated MFMA rows, half the per-lane work of a dependent step);
    LAS_SEQ_ROWS16 selects the 16-row tiling.  Both are checked against the oracle by test_rnn_seq_fwd_bwd (default) and here
    (flag); the two tilings must also agree with EACH OTHER to rounding noise of the bf16 exchange."""
    from las import _hip
    cell, prec, B, T, H = case
    GH = 4 * H
    g = torch.Generator().manual_seed(7)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).to(torch.bfloat16)
    whh = [((torch.rand(H, GH, generator=g) * 2 - 1) * 0.06).cuda() for _ in range(2)]
    R = torch.randn(B, T, 2 * H, generator=g).to(torch.bfloat16).cuda()
    res = []
    for flags in (0, _hip.SEQ_ROWS16):
        gates = xp.cuda().clone()
        out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
        cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
        _hip.rnn_seq_fwd(1, 1, B, T, H, gates, whh[0], whh[1], GH, out, 2 * H, T * 2 * H, cst, flags=flags)
        db = [torch.zeros(GH, device="cuda") for _ in range(2)]
        _hip.rnn_seq_bwd(1, 1, B, T, H, gates, whh[0], whh[1], GH, out, 2 * H, T * 2 * H, cst, R, 2 * H, T * 2 * H,
                         db_fw=db[0], db_bw=db[1], flags=flags)
        torch.cuda.synchronize()
        _hip.check_status()
        res.append((out.float().cpu(), cst.float().cpu(), gates.float().cpu(), db[0].cpu(), db[1].cpu()))
    names = ("h", "c", "dZ", "db_fw", "db_bw")
    for n, a, b in zip(names, res[0], res[1]):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 2e-2 * scale, (n, case, (a - b).abs().max().item(), scale)


def test_chunked_x_projection_is_waited_for():
    """las_rnn_seq_fwd_chunked: the forward sweep may be launched while its x-projection is still being written in time chunks
    (both ends of the sequence first) by kernels of ANOTHER stream; it must read a frame only after the chunk's flag.  Here the
    chunks are copied in late and slowly (sleep kernels between them) from a second stream; the result has to equal the sweep
    over the complete x-projection bit for bit.  A flag that never arrives must surface as a status error, not as a hang."""
    from las import _hip
    B, T, H, cs = 24, 300, 256, 32
    GH = 4 * H
    assert _hip.rnn_seq_fwd_chunks_ok(1, 1, B, H)
    g = torch.Generator().manual_seed(11)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda().to(torch.bfloat16)
    w = [((torch.rand(H, GH, generator=g) * 2 - 1) * 0.06).cuda() for _ in range(2)]

    def sweep(gates, **kw):
        out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
        cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
        _hip.rnn_seq_fwd(1, 1, B, T, H, gates, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst, **kw)
        return out, cst

    ref_out, ref_c = sweep(xp.clone())
    torch.cuda.synchronize()
    th = (T + 1) // 2
    nch = (th + cs - 1) // cs
    gates = torch.full_like(xp, float("nan"))                   # frames of incomplete chunks are poison
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()

    def put(k):
        lo0, lo1 = k * cs, min((k + 1) * cs, th)
        hi0, hi1 = max(T - lo1, lo1), T - lo0
        gates[:, lo0:lo1] = xp[:, lo0:lo1]
        gates[:, hi0:hi1] = xp[:, hi0:hi1]
        if k:                                                   # chunk 0 precedes the sweep in stream order: never flagged, never waited for
            _hip.set_word(flag, k + 1)

    put(0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(1, nch):
            torch.cuda._sleep(400000)                           # ~0.17 ms: the sweep reaches the chunk boundary first and has to wait
            put(k)
    out, cst = sweep(gates, chunk_flag=flag, chunk_steps=cs)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _hip.check_status()
    assert torch.equal(out, ref_out) and torch.equal(cst, ref_c)
    # the flag never moves past chunk 0: bounded wait, then the status word says so
    gates = xp.clone()
    flag.fill_(1)
    sweep(gates, chunk_flag=flag, chunk_steps=cs, flags=_hip.seq_spin_log2(6))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        _hip.check_status()


@pytest.mark.parametrize("pairs", [False, True])
def test_chunked_upstream_gradient_is_waited_for(pairs):
    """las_rnn_seq_bwd_db_chunked: the BPTT sweep may start while its upstream gradient `dout` is still being written in chunks of
    producer rows (frames, or frame PAIRS under a pyramid dense layer) from both ends of the sequence by another stream.  Late,
    slow chunks (poison in the frames not yet delivered) must give the same dZ / bias gradients as the sweep over the complete
    dout, bit for bit; a flag that never arrives surfaces as a status error."""
    from las import _hip
    B, T, H, c = 24, 301 if pairs else 300, 256, 32
    GH = 4 * H
    assert _hip.rnn_seq_bwd_chunks_ok(1, 1, B, H)
    Tp = T + (T % 2)
    Tq = Tp // 2 if pairs else T
    g = torch.Generator().manual_seed(12)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda().to(torch.bfloat16)
    w = [((torch.rand(H, GH, generator=g) * 2 - 1) * 0.06).cuda() for _ in range(2)]
    out = torch.zeros(B, Tp, 2 * H, device="cuda", dtype=torch.bfloat16)
    cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
    act = xp.clone()
    _hip.rnn_seq_fwd(1, 1, B, T, H, act, w[0], w[1], GH, out, 2 * H, Tp * 2 * H, cst)
    dfull = (torch.randn(B, Tp, 2 * H, generator=g) * 0.1).cuda().to(torch.bfloat16)

    def bptt(dout, **kw):
        gz = act.clone()
        db = [torch.zeros(GH, device="cuda") for _ in range(2)]
        _hip.rnn_seq_bwd(1, 1, B, T, H, gz, w[0], w[1], GH, out, 2 * H, Tp * 2 * H, cst, dout, 2 * H, Tp * 2 * H,
                         db_fw=db[0], db_bw=db[1], **kw)
        return gz, db

    ref_z, ref_db = bptt(dfull)
    torch.cuda.synchronize()
    rows = dfull.view(B, Tq, -1)
    dout = torch.full_like(dfull, float("nan"))
    drows = dout.view(B, Tq, -1)
    th = (Tq + 1) // 2
    nch = (th + c - 1) // c
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()

    def put(k):
        lo0, lo1 = k * c, min((k + 1) * c, th)
        hi0, hi1 = max(Tq - lo1, lo1), Tq - lo0
        drows[:, lo0:lo1] = rows[:, lo0:lo1]
        drows[:, hi0:hi1] = rows[:, hi0:hi1]
        if k:
            _hip.set_word(flag, k + 1)

    put(0)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(1, nch):
            torch.cuda._sleep(400000)
            put(k)
    gz, db = bptt(dout, chunk_flag=flag, chunk_rows=c, n_rows=Tq)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _hip.check_status()
    assert torch.equal(gz, ref_z)
    for a, b in zip(db, ref_db):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
    flag.fill_(1)
    bptt(dfull, chunk_flag=flag, chunk_rows=c, n_rows=Tq, flags=_hip.seq_spin_log2(6))
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        _hip.check_status()


@pytest.mark.parametrize("cell,H,B,Bprep", [(1, 256, 48, 48), (1, 256, 20, 48), (1, 128, 9, 16), (0, 256, 24, 24), (1, 512, 16, 16)])
def test_prepared_workspace_sweeps_equal_self_packing_sweeps(cell, H, B, Bprep):
    """las_rnn_seq_prepare (round 5): weight packs + exchange-state clears of several sweeps in one launch; a sweep that gets such a
    workspace with LAS_SEQ_PREPARED launches only its persistent kernel.  Forward and BPTT through prepared workspaces (prepared for a
    batch >= the one swept, both passes' jobs in ONE prepare call) must equal the self-packing launches bit for bit -- and a workspace
    used twice without a new prepare must NOT be relied on (the host layer consumes it once): here only the documented use is checked."""
    from las import _hip
    T = 77
    G = 4 if cell == 1 else 1
    GH = G * H
    g = torch.Generator().manual_seed(100 + H + B)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda().to(torch.bfloat16)
    w = [((torch.rand(H + 8, GH, generator=g) * 2 - 1) * 0.06).cuda() for _ in range(2)]        # W_hh = rows 8.. of a larger kernel (an offset, as in the layers)
    off = 8 * GH
    dfull = (torch.randn(B, T, 2 * H, generator=g) * 0.1).cuda().to(torch.bfloat16)

    def run(ws_f=None, ws_b=None):
        act = xp.clone()
        out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
        cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16) if cell == 1 else None
        _hip.rnn_seq_fwd(cell, 1, B, T, H, act, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst, wf_off=off, wb_off=off, prepared_ws=ws_f)
        gz = act.clone()
        db = [torch.zeros(GH, device="cuda") for _ in range(2)]
        _hip.rnn_seq_bwd(cell, 1, B, T, H, gz, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst, dfull, 2 * H, T * 2 * H,
                         wf_off=off, wb_off=off, db_fw=db[0], db_bw=db[1], prepared_ws=ws_b)
        torch.cuda.synchronize()
        _hip.check_status()
        return act, out, cst, gz, db

    ref = run()
    nb = int(_hip.lib().las_rnn_seq_workspace_bytes(cell, 1, H, Bprep))
    ws = [torch.full((nb,), 0x5a, dtype=torch.uint8, device="cuda") for _ in range(2)]                 # poison: the prepare must clear what matters
    for rep in range(2):                                                                                 # (second round: a DIRTY workspace is prepared again)
        _hip.rnn_seq_prepare([(cell, H, Bprep, False, w[0], w[1], GH, off, off, ws[0]), (cell, H, Bprep, True, w[0], w[1], GH, off, off, ws[1])])
        got = run(ws[0], ws[1])
        for a, b in zip(ref[:4], got[:4]):
            assert (a is None and b is None) or torch.equal(a, b)
        for a, b in zip(ref[4], got[4]):
            assert torch.equal(a, b)


def test_bptt_sweep_that_publishes_its_progress_is_followed_window_by_window():
    """las_rnn_seq_bwd_db_progress + las_wait_words_min (round 5): the sweep stores d(pre-activation) with agent-scope stores and publishes,
    every `ps` steps, how many steps have reached memory.  (1) dZ and the bias gradients are bit-identical to the chunked sweep without
    progress; (2) a FOLLOWER on another stream, gated by las_wait_words_min, copies the frames each window vouches for WHILE the sweep is
    still running -- forward direction [T - s, T), backward direction [0, s) after s steps: every copy must equal the final dZ (stale or
    unwritten frames would not); (3) the words end at T."""
    from las import _hip
    B, T, H, c, ps = 24, 640, 256, 32, 64
    GH = 4 * H
    nw = _hip.rnn_seq_bwd_progress_words(1, 1, B, H)
    assert nw == 2 * ((B + 7) // 8) * 4
    g = torch.Generator().manual_seed(21)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda().to(torch.bfloat16)
    w = [((torch.rand(H, GH, generator=g) * 2 - 1) * 0.06).cuda() for _ in range(2)]
    out = torch.zeros(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
    cst = torch.zeros(B, T, 2, H, device="cuda", dtype=torch.bfloat16)
    act = xp.clone()
    _hip.rnn_seq_fwd(1, 1, B, T, H, act, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst)
    dfull = (torch.randn(B, T, 2 * H, generator=g) * 0.1).cuda().to(torch.bfloat16)
    flag = torch.full((1,), 1000, dtype=torch.int32, device="cuda")             # every chunk of dout is there

    def bptt(progress=None):
        gz = act.clone()
        db = [torch.zeros(GH, device="cuda") for _ in range(2)]
        _hip.rnn_seq_bwd(1, 1, B, T, H, gz, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst, dfull, 2 * H, T * 2 * H, db_fw=db[0], db_bw=db[1],
                         chunk_flag=flag, chunk_rows=c, n_rows=T, progress=progress, progress_steps=ps)
        return gz, db

    ref_z, ref_db = bptt()
    torch.cuda.synchronize()
    prog = torch.zeros(nw, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    nwin = T // ps
    copies = torch.zeros(nwin, B, ps, 2, GH, device="cuda", dtype=torch.bfloat16)
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    gz = act.clone()
    db = [torch.zeros(GH, device="cuda") for _ in range(2)]
    ev.record()
    _hip.rnn_seq_bwd(1, 1, B, T, H, gz, w[0], w[1], GH, out, 2 * H, T * 2 * H, cst, dfull, 2 * H, T * 2 * H, db_fw=db[0], db_bw=db[1],
                     chunk_flag=flag, chunk_rows=c, n_rows=T, progress=prog, progress_steps=ps)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        for k in range(nwin):
            _hip.wait_words_min(prog, nw, (k + 1) * ps)
            copies[k, :, :, 0] = gz[:, T - (k + 1) * ps:T - k * ps, 0]           # forward direction: frames from the end
            copies[k, :, :, 1] = gz[:, k * ps:(k + 1) * ps, 1]                  # backward direction: frames from the start
    main.wait_stream(side)
    torch.cuda.synchronize()
    _hip.check_status()
    assert torch.equal(gz, ref_z)
    for a, b in zip(db, ref_db):
        assert torch.equal(a, b)
    assert int(prog.min()) == T and int(prog.max()) == T
    for k in range(nwin):
        assert torch.equal(copies[k, :, :, 0], ref_z[:, T - (k + 1) * ps:T - k * ps, 0]), ("fw", k)
        assert torch.equal(copies[k, :, :, 1], ref_z[:, k * ps:(k + 1) * ps, 1]), ("bw", k)
    # a follower whose words never arrive marks the step invalid instead of reading on
    _hip.check(_hip.lib().las_wait_words_min(_hip.p(torch.zeros(4, dtype=torch.int32, device="cuda")), 4, 1, 200, _hip.p(_hip.status_word("cuda")), 2,
                                             _hip.stream()), "las_wait_words_min")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        _hip.check_status()


def test_wait_announce_passes_for_numbers_that_have_come_and_gone():
    """las_wait_announce: the hold of side-stream work on a sweep's announcement passes at once when that sweep -- or a later one -- has
    announced itself (cyclic numbers 1..1023), and sits out its bound otherwise (las_wait_word waits for equality only)."""
    import time
    from las import _hip
    lib = _hip.lib()
    word = torch.zeros(1, dtype=torch.int32, device="cuda")

    def timed(n, bound_us):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _hip.check(lib.las_wait_announce(_hip.p(word), n, bound_us, _hip.stream()), "las_wait_announce")
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    timed(1, 10)                                          # warm-up
    bound = 20000
    word.fill_(7)
    assert timed(7, bound) < bound / 4                    # reached exactly
    assert timed(5, bound) < bound / 4                    # two later sweeps have announced themselves meanwhile
    assert timed(8, bound) > bound * 0.9                  # not yet
    word.fill_(3)
    assert timed(1020, bound) < bound / 4                 # wrapped: 1020, 1021, 1022, 1023, 1, 2, 3
    word.zero_()
    assert timed(1, bound) > bound * 0.9                  # nothing announced yet
    assert lib.las_wait_announce(_hip.p(word), 0, 10, _hip.stream()) < 0


@pytest.mark.parametrize("cell", [1, 0])
def test_forward_sweep_over_rows_of_different_lengths_equals_each_row_alone(cell):
    """las_rnn_seq_fwd_rows: every real frame of a row (both directions, h and c) is bit-identical to sweeping that row alone at its own
    length; behind a row's last frame h and c are zero."""
    from las import _hip
    dev = "cuda"
    B, T, H = 11, 77, 256
    G = 4 if cell else 1
    g = torch.Generator().manual_seed(5)
    io = _hip.rnn_seq_io_dtype(cell, 1, H)
    xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).to(dev).to(io)
    w0 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
    w1 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
    lens = [77, 1, 40, 76, 13, 77, 2, 55, 31, 64, 9]
    row_T = torch.tensor(lens, dtype=torch.int32, device=dev)
    out = torch.full((B, T, 2 * H), 7.0, device=dev, dtype=io)
    cst = torch.full((B, T, 2, H), 7.0, device=dev, dtype=io) if cell else None
    if not _hip.rnn_seq_fwd_rows_ok(cell, 1, B, H):
        # (the tanh cell: no 8-row helper-wave kernel) the contract is "ask las_rnn_seq_fwd_rows_ok": a configuration it does not serve
        # must be REFUSED, not swept as if the rows were of equal length
        with pytest.raises(RuntimeError, match="rows of different lengths"):
            _hip.rnn_seq_fwd(cell, 1, B, T, H, xp.clone(), w0, w1, G * H, out, 2 * H, T * 2 * H, cst, row_T=row_T)
        return
    _hip.rnn_seq_fwd(cell, 1, B, T, H, xp.clone(), w0, w1, G * H, out, 2 * H, T * 2 * H, cst, row_T=row_T)
    for b, n in enumerate(lens):
        o1 = torch.empty(1, n, 2 * H, device=dev, dtype=io)
        c1 = torch.empty(1, n, 2, H, device=dev, dtype=io) if cell else None
        _hip.rnn_seq_fwd(cell, 1, 1, n, H, xp[b:b + 1, :n].clone().contiguous(), w0, w1, G * H, o1, 2 * H, n * 2 * H, c1)
        assert torch.equal(out[b, :n], o1[0]), (b, n)
        assert bool((out[b, n:] == 0).all())
        if cell:
            assert torch.equal(cst[b, :n], c1[0]) and bool((cst[b, n:] == 0).all())
    _hip.check_status(dev)
