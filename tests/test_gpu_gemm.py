"""K1/K3/K4 parity: las_gemm (C ABI) vs an fp64 torch-CPU contraction of the same operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B, tA, tB, alpha, beta, C0, bias, act, mask_period=0, mask_skip=0):
    a = A.double().cpu()
    b = B.double().cpu()
    a = a.transpose(-1, -2) if tA else a
    b = b.transpose(-1, -2) if tB else b
    if mask_period:
        K = a.shape[-1]
        keep = torch.tensor([(k % mask_period) != mask_skip for k in range(K)], dtype=torch.float64)
        a = a * keep
    r = alpha * (a @ b)
    if beta:
        r = r + beta * C0.double().cpu()
    if bias is not None:
        r = r + bias.double().cpu()
    if act:
        r = torch.tanh(r)
    return r


CASES = [
    # M, N, K, tA, tB
    (48, 64, 39, 0, 0), (130, 70, 100, 0, 0), (256, 256, 512, 0, 0), (300, 512, 64, 0, 1),
    (512, 1024, 3000, 1, 0), (128, 30, 512, 0, 0), (48, 2048, 1152, 0, 0), (48, 1152, 2048, 0, 1),
    (39, 256, 2500, 1, 0), (7, 5, 3, 0, 0), (1, 128, 64, 0, 0), (200, 200, 33, 1, 1),
    (33, 100, 300, 0, 0), (64, 96, 130, 1, 0), (17, 40, 260, 0, 1), (48, 30, 512, 0, 0),       # M <= 64: the parity mode's skinny exact-fp32 MFMA kernel
]


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_gemm_matches_fp64(prec, case):
    from las import _hip
    M, N, K, tA, tB = case
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g).cuda()
    B = torch.randn((N, K) if tB else (K, N), generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    C0 = torch.randn(M, N, generator=g).cuda()
    C = C0.clone()
    act = 1 if (M + N) % 2 else 0
    scale = 1.0 / (K ** 0.5) if act else 1.0
    _hip.gemm(prec, A, B, C, bool(tA), bool(tB), M, N, K, A.shape[1], B.shape[1], N,
              alpha=scale, beta=0.5, bias=bias, act=act)
    ref = _ref(A, B, tA, tB, scale, 0.5, C0, bias, act)
    err = (C.double().cpu() - ref).abs().max().item()
    mag = ref.abs().max().item() + 1e-9
    # tolerance: fp32 FMA chains ~1e-6*sqrt(K); bf16 operands ~ 2^-9 relative per product
    tol = (2e-5 if prec == 0 else 2e-2) * max(1.0, mag)
    assert err < tol, (case, prec, err, mag)


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_batched_strided_and_mask(prec):
    from las import _hip
    g = torch.Generator().manual_seed(5)
    # batched TN:  C[b] = A[b]^T . B[b]   A [bt, K, M], B [bt, K, N]
    bt, K, M, N = 5, 70, 96, 40
    A = torch.randn(bt, K, M, generator=g).cuda()
    B = torch.randn(bt, K, N, generator=g).cuda()
    C = torch.zeros(bt, M, N).cuda()
    _hip.gemm(prec, A, B, C, True, False, M, N, K, M, N, N, batch=bt, strideA=K * M, strideB=K * N, strideC=M * N)
    ref = A.double().cpu().transpose(1, 2) @ B.double().cpu()
    assert (C.double().cpu() - ref).abs().max().item() < (1e-4 if prec == 0 else 0.3)
    # masked TN with element offsets (the dW_hh form): rows k with k % 10 == 0 dropped
    K, M, N = 4000, 64, 128
    A = torch.randn(K + 1, M, generator=g).cuda()
    B = torch.randn(K, N, generator=g).cuda()
    C = torch.zeros(M, N).cuda()
    _hip.gemm(prec, A, B, C, True, False, M, N, K, M, N, N, mask_period=10, mask_skip=0, a_off=0)
    ref = _ref(A[:K], B, 1, 0, 1.0, 0.0, None, None, 0, 10, 0)
    assert (C.double().cpu() - ref).abs().max().item() < (1e-3 if prec == 0 else 2.0)


def test_colsum_and_tanh_bwd():
    from las import _hip
    g = torch.Generator().manual_seed(9)
    X = torch.randn(5000, 300, generator=g).cuda()
    out = torch.ones(300).cuda()
    _hip.colsum(X, 5000, 300, 300, out, beta=2.0)
    ref = 2.0 + X.double().cpu().sum(0)
    assert (out.double().cpu() - ref).abs().max().item() < 1e-3
    Y = torch.tanh(torch.randn(77, 130, generator=g)).cuda()
    dY = torch.randn(77, 130, generator=g).cuda()
    dX = torch.empty_like(Y)
    _hip.tanh_bwd(Y, 130, dY, 130, dX, 130, 77, 130)
    assert torch.allclose(dX.cpu(), (dY * (1 - Y * Y)).cpu(), atol=1e-6)


@pytest.mark.parametrize("M,N,K,act,out", [(128, 128, 64, 0, "bf16"), (300, 2048, 512, 0, "bf16"), (1000, 512, 1024, 1, "bf16"),
                                           (257, 132, 128, 1, "f32"), (61, 512, 2048, 0, "f32"), (4096, 2048, 64, 0, "bf16"),
                                           # 256 x 256 tiles (>= 192 of them, N % 256 == 0): ragged last row block, odd k-tile count
                                           (6200, 2048, 192, 0, "bf16"), (24500, 512, 256, 1, "bf16"), (12300, 1024, 128, 0, "f32")])
def test_gemm_kk_bf16_operands_k_contiguous(M, N, K, act, out):
    """las_gemm_kk (LDS-DMA staged 128x128x64 or 256x256x64 MFMA tiles): C = act(A . B^T + bias) with bf16 operands vs a float64 product
    of the SAME bf16 values (exact operands, so the only error is fp32 accumulation + the output rounding)."""
    from las import _hip
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K + 8, generator=g) * 0.5).to(torch.bfloat16).cuda()          # padded pitch
    B = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    C = torch.full((M, N + 4), 7.0, dtype=torch.bfloat16 if out == "bf16" else torch.float32, device="cuda")
    _hip.gemm_kk(A, B, C, M, N, K, K + 8, K, N + 4, bias=bias, act=act)
    ref = A[:, :K].double().cpu() @ B.double().cpu().t() + bias.double().cpu()
    if act:
        ref = torch.tanh(ref)
    got = C[:, :N].double().cpu()
    tol = 2e-5 * K ** 0.5 + (4e-3 * ref.abs().max().item() if out == "bf16" else 0)       # bf16 output: half an ulp of the result
    assert (got - ref).abs().max().item() < tol + 1e-4
    assert (C[:, N:].float() == 7.0).all()                                                 # nothing written past N


@pytest.mark.parametrize("case", [(3, 37, 0, 5, 30, 7, 128, 64), (2, 64, 16, 16, 32, 16, 256, 128), (4, 21, 0, 11, 11, 10, 64, 64),
                                  (2, 50, 10, 8, 40, 0, 192, 64), (48, 300, 0, 70, 230, 70, 2048, 128)])     # last: 256 x 256 tiles
def test_gemm_kk_frames_touches_exactly_the_selected_frames(case):
    """las_gemm_kk_frames: C[b, t, :] = A[b, t, :] . B^T + bias for t in [lo0, lo0+nlo) u [hi0, hi0+nhi) only (the time chunks
    of a layer's x-projection); every other frame of C keeps its old contents."""
    from las import _hip
    nb, T, lo0, nlo, hi0, nhi, N, K = case
    g = torch.Generator().manual_seed(sum(case))
    A = (torch.randn(nb, T, K, generator=g) * 0.5).cuda().to(torch.bfloat16)
    Bm = (torch.randn(N, K, generator=g) * 0.1).cuda().to(torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    C = torch.full((nb, T, N), 7.0, device="cuda", dtype=torch.bfloat16)
    _hip.gemm_kk_frames(A, Bm, C, nb, T, lo0, nlo, hi0, nhi, N, K, K, K, N, bias=bias)
    torch.cuda.synchronize()
    ref = (A.float() @ Bm.float().t() + bias).to(torch.bfloat16).float()
    sel = torch.zeros(T, dtype=torch.bool)
    sel[lo0:lo0 + nlo] = True
    sel[hi0:hi0 + nhi] = True
    got = C.float().cpu()
    assert (got[:, sel] - ref.cpu()[:, sel]).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    assert (got[:, ~sel] == 7.0).all()


@pytest.mark.parametrize("rows,cols,dt", [(48, 4096, "f32"), (48, 1000, "bf16"), (7000, 512, "bf16"), (7000, 520, "bf16"), (600, 64, "bf16")])
def test_colsum_paths(rows, cols, dt):
    """las_colsum_dt: few rows -> one direct pass; bf16 with 16-byte rows -> 8 columns per thread; else the generic two-stage kernel."""
    from las import _hip
    g = torch.Generator().manual_seed(rows + cols)
    X = torch.randn(rows, cols, generator=g)
    Xd = X.cuda().to(torch.bfloat16) if dt == "bf16" else X.cuda()
    out = torch.full((cols,), 0.5, device="cuda")
    _hip.colsum(Xd, rows, cols, cols, out, beta=2.0)
    want = Xd.float().cpu().double().sum(0) + 1.0
    assert (out.cpu().double() - want).abs().max().item() < 1e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,K,batch", [(256, 384, 5000, 1), (512, 1024, 20011, 1), (128, 128, 31, 1), (256, 256, 300, 5),
                                         (40, 256, 9001, 1), (64, 1024, 4100, 1)])      # last two: the 64-row variant (first-layer dW_ih)
def test_weight_gradient_form_bf16_tn(M, N, K, batch):
    """las_gemm with bf16 operands that are both k-strided (dW = X^T . dZ): served by gemm_tn_tr_kernel ([k][m] LDS tiles read
    with ds_read_b64_tr_b16) when M, N are multiples of 128 (or M <= 64, a multiple of 8) -- split-K, a contraction length that is no multiple of 32, element
    offsets into wider tensors, batched with strides, beta = 1 accumulation; against float64 on the same bf16 values, and against
    the register-transposing kernel it replaced when the library was built with the development switches (`make prof`:
    las_dev_gemm_tn_tr(0); the shipping library does not export them)."""
    from las import _hip
    g = torch.Generator().manual_seed(M + N + K + batch)
    X = (torch.randn(batch, K, M + 64, generator=g) * 0.5).cuda().to(torch.bfloat16)           # operands are column slices of wider tensors
    Z = (torch.randn(batch, K, 2 * N, generator=g) * 0.5).cuda().to(torch.bfloat16)
    C0 = torch.randn(batch, M, N, generator=g).cuda()
    res = []
    dev_switch = getattr(_hip.lib(), "las_dev_gemm_tn_tr", None) if hasattr(_hip.lib(), "las_dev_gemm_tn_tr") else None
    for on in (1, 0) if dev_switch is not None else (1,):
        if dev_switch is not None:
            dev_switch(on)
        C = C0.clone()
        _hip.gemm(_hip.PREC_BF16, X, Z, C, True, False, M, N, K, M + 64, 2 * N, N, beta=1.0, batch=batch,
                  strideA=K * (M + 64), strideB=K * 2 * N, strideC=M * N, a_off=64, b_off=N)
        res.append(C.double().cpu())
    if dev_switch is not None:
        dev_switch(1)
    ref = C0.double().cpu() + X[:, :, 64:].double().cpu().transpose(1, 2) @ Z[:, :, N:].double().cpu()
    tol = 3e-6 * K ** 0.5 * max(1.0, ref.abs().max().item())                                  # fp32 accumulation of exact bf16 products
    assert (res[0] - ref).abs().max().item() < tol
    assert len(res) == 1 or (res[0] - res[1]).abs().max().item() < tol


@pytest.mark.parametrize("M,K,N", [(256, 512, 2048), (17, 1024, 2048), (256, 512, 28), (1, 40, 16), (1000, 64, 100)])
def test_skinny_gemm_matches_float64_on_the_rounded_operands(M, K, N):
    """las_gemm_skinny_pack + las_gemm_skinny (the LM step's products in beam search; reference lang/char_rnn_model.py:57-66 driven
    from las/beam_search.py:226-236): C = bf16(A) . bf16(W[row0:row0+K]) + bias, and the accumulate form into a column window of a
    wider C -- against float64 on the same bf16-rounded operands (fp32 accumulation: 3e-6 sqrt(K))."""
    from las import _hip
    g = torch.Generator().manual_seed(M + K + N)
    row0 = 24
    W = (torch.randn(row0 + K, N + 8, generator=g) * 0.2).cuda()
    A = (torch.randn(M, K, generator=g) * 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    packed = _hip.skinny_pack(W, K, N, row0=row0)
    Ar, Wr = A.to(torch.bfloat16).double().cpu(), W[row0:, :N].to(torch.bfloat16).double().cpu()
    ref = Ar @ Wr + bias.double().cpu()
    tol = 3e-6 * K ** 0.5 * max(1.0, ref.abs().max().item())
    C = torch.full((M, N), 7.0, device="cuda")
    _hip.skinny_gemm(A, packed, C, M, K, N, K, N, bias=bias)
    assert (C.double().cpu() - ref).abs().max().item() < tol
    wide = torch.randn(M, N + 5, generator=g).cuda()
    want = wide.double().cpu().clone()
    want[:, 2:2 + N] += Ar @ Wr
    _hip.skinny_gemm(A, packed, wide, M, K, N, K, N + 5, accumulate=True, c_off=2)
    assert (wide.double().cpu() - want).abs().max().item() < tol
    l = _hip.lib()
    assert l.las_gemm_skinny(_hip.p(A), K, 2000, K, _hip.p(packed), N, _hip.p(C), N, None, 0, None) < 0      # M > 1024 is refused


@pytest.mark.parametrize("I,H,G,B,T", [(40, 128, 4, 3, 50), (240, 256, 4, 5, 333), (512, 256, 4, 48, 160), (1024, 256, 4, 4, 77),
                                       (512, 256, 1, 7, 129), (100, 128, 1, 2, 2)])
def test_both_weight_gradients_in_one_pass_over_dz(I, H, G, B, T):
    """las_wgrad_ih_hh: dW[0:I] += X^T dZ_d and dW[I:I+H] += sum_b sum_t h_prev(b,t)^T dZ_d(b,t) with h_prev read from the layer output one
    frame back (direction 0) / ahead (direction 1), against float64 on the same bf16 values and against the two las_gemm products +
    column sum it replaces (the matmul gradient of the cell kernel under bidirectional_dynamic_rnn, reference las/layers.py:49-53)."""
    from las import _hip
    GH = G * H
    Ik = (I + 63) // 64 * 64
    Tp = T + 3                                                      # the layer output is stored with padded frames
    g = torch.Generator().manual_seed(I + H + B + T)
    X = torch.zeros(B, T, Ik)
    X[:, :, :I] = torch.randn(B, T, I, generator=g) * 0.5
    X = X.cuda().to(torch.bfloat16)
    out = (torch.randn(B, Tp, 2 * H, generator=g) * 0.5).cuda().to(torch.bfloat16)
    dZ = (torch.randn(B, T, 2 * GH, generator=g) * 0.5).cuda().to(torch.bfloat16)
    for d in (0, 1):
        dW0 = torch.randn(I + H, GH, generator=g).cuda()
        dW = dW0.clone()
        _hip.wgrad_ih_hh(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, d, dW)
        Zd = dZ[:, :, d * GH:(d + 1) * GH].double().cpu()
        Od = out[:, :T, d * H:(d + 1) * H].double().cpu()
        hp = torch.zeros(B, T, H, dtype=torch.float64)
        if d == 0:
            hp[:, 1:] = Od[:, :-1]
        else:
            hp[:, :-1] = Od[:, 1:]
        ref = dW0.double().cpu()
        ref[:I] += X[:, :, :I].double().cpu().reshape(B * T, I).t() @ Zd.reshape(B * T, GH)
        ref[I:] += hp.reshape(B * T, H).t() @ Zd.reshape(B * T, GH)
        tol = 3e-6 * (B * T) ** 0.5 * max(1.0, ref.abs().max().item())
        assert (dW.double().cpu() - ref).abs().max().item() < tol, (d,)
        # the two products it replaces
        old = dW0.clone()
        Ig = (I + 3) // 4 * 4
        if Ig == I:
            _hip.gemm(_hip.PREC_BF16, X, dZ, old, True, False, Ig, GH, B * T, Ik, 2 * GH, GH, beta=1.0, b_off=d * GH)
            part = torch.empty(B, H, GH, device="cuda")
            a_off = d * H + (0 if d == 0 else 2 * H)
            b_off = d * GH + (2 * GH if d == 0 else 0)
            _hip.gemm(_hip.PREC_BF16, out, dZ, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                      strideA=Tp * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=a_off, b_off=b_off)
            _hip.colsum(part, B, H * GH, H * GH, old[I:].reshape(-1), beta=1.0)
            assert (dW - old).abs().max().item() < tol
    # run to run: bit-identical (fixed-order split-K)
    a, b = torch.zeros(I + H, GH, device="cuda"), torch.zeros(I + H, GH, device="cuda")
    _hip.wgrad_ih_hh(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 0, a)
    _hip.wgrad_ih_hh(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 0, b)
    assert torch.equal(a, b)
    # both directions in one launch (the backward direction with its own input copy, as under input dropout)
    X2 = torch.zeros(B, T, Ik)
    X2[:, :, :I] = torch.randn(B, T, I, generator=g) * 0.5
    X2 = X2.cuda().to(torch.bfloat16)
    f2, b2 = torch.zeros(I + H, GH, device="cuda"), torch.zeros(I + H, GH, device="cuda")
    _hip.wgrad_ih_hh(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 2, f2, b2, X2)
    b1 = torch.zeros(I + H, GH, device="cuda")
    _hip.wgrad_ih_hh(X2, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 1, b1)
    tol2 = 3e-6 * (B * T) ** 0.5 * max(1.0, a.abs().max().item(), b1.abs().max().item())     # (the pair splits the frames differently)
    assert (f2 - a).abs().max().item() < tol2 and (b2 - b1).abs().max().item() < tol2


@pytest.mark.parametrize("I,H,G,B,T,S,cap", [(40, 256, 4, 6, 500, 160, 192), (240, 256, 4, 5, 333, 100, 0), (512, 256, 1, 7, 129, 40, 48), (64, 128, 4, 48, 330, 160, 192)])
def test_weight_gradients_in_frame_windows_sum_to_the_whole_sequence(I, H, G, B, T, S, cap):
    """las_wgrad_ih_hh_window (round 5): the windows a BPTT sweep completes one after the other -- the forward direction's frames from the
    END of the sequence, the backward direction's from its START -- contracted one launch per window, both directions per launch, some with
    a workgroup cap (few long k-chunks beside a running sweep): accumulated they are the whole-sequence gradient (float64 reference on the
    same bf16 values), bit-identical run to run; h_prev crosses window borders (frame t - 1 / t + 1 of the NEIGHBOURING window)."""
    from las import _hip
    GH = G * H
    Ik = (I + 63) // 64 * 64
    Tp = T + (T % 2)
    g = torch.Generator().manual_seed(I + H + B + T)
    X = torch.zeros(B, T, Ik)
    X[:, :, :I] = torch.randn(B, T, I, generator=g) * 0.5
    X = X.cuda().to(torch.bfloat16)
    out = (torch.randn(B, Tp, 2 * H, generator=g) * 0.5).cuda().to(torch.bfloat16)
    dZ = (torch.randn(B, T, 2 * GH, generator=g) * 0.5).cuda().to(torch.bfloat16)

    def run():
        dW = [torch.zeros(I + H, GH, device="cuda") for _ in range(2)]
        nwin = (T + S - 1) // S
        for c in range(nwin):
            lo, n = c * S, min(S, T - c * S)
            _hip.wgrad_ih_hh_window(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 2, T - lo - n, lo, n, cap if c + 1 < nwin else 0,
                                    dW[0], dW[1])
        return dW

    got, again = run(), run()
    for d in (0, 1):
        assert torch.equal(got[d], again[d])
        Zd = dZ[:, :, d * GH:(d + 1) * GH].double().cpu()
        Od = out[:, :T, d * H:(d + 1) * H].double().cpu()
        hp = torch.zeros(B, T, H, dtype=torch.float64)
        if d == 0:
            hp[:, 1:] = Od[:, :-1]
        else:
            hp[:, :-1] = Od[:, 1:]
        ref = torch.zeros(I + H, GH, dtype=torch.float64)
        ref[:I] = X[:, :, :I].double().cpu().reshape(B * T, I).t() @ Zd.reshape(B * T, GH)
        ref[I:] = hp.reshape(B * T, H).t() @ Zd.reshape(B * T, GH)
        tol = 3e-6 * (B * T) ** 0.5 * max(1.0, ref.abs().max().item())
        assert (got[d].double().cpu() - ref).abs().max().item() < tol, d
    # one direction alone (dir = 0 / 1) over the same windows
    one = torch.zeros(I + H, GH, device="cuda")
    for c in range((T + S - 1) // S):
        lo, n = c * S, min(S, T - c * S)
        _hip.wgrad_ih_hh_window(X, Ik, I, out, 2 * H, Tp * 2 * H, dZ, 2 * GH, B, T, H, GH, 1, 0, lo, n, 0, one)
    assert (one - got[1]).abs().max().item() < 3e-6 * (B * T) ** 0.5 * max(1.0, got[1].abs().max().item())


@pytest.mark.parametrize("case", [(48, 2048, 1152, 0, 0), (48, 1152, 2048, 0, 1), (17, 96, 1024, 0, 0), (64, 40, 4100, 1, 0)])
def test_skinny_f32_k_slices_are_reproducible(case):
    """The parity mode's per-step cell products (M <= 64, K >= 512): K slices over blockIdx.z, partial tiles to the stream's scratch,
    added in slice order (gemm.hip gemm_mf32_skinny_kernel + gemm_splitk_reduce_kernel).  40 repetitions, alone and with a second
    stream running the same product, are bit-identical, inside a captured graph as well -- and right (float64 of the same operands)."""
    from las import _hip
    M, N, K, tA, tB = case
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g).cuda()
    B = torch.randn((N, K) if tB else (K, N), generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    C0 = torch.randn(M, N, generator=g).cuda()

    def product(out):
        out.copy_(C0)
        _hip.gemm(0, A, B, out, bool(tA), bool(tB), M, N, K, A.shape[1], B.shape[1], N, alpha=0.25, beta=0.5, bias=bias)

    first = torch.empty_like(C0)
    product(first)
    ref = _ref(A, B, tA, tB, 0.25, 0.5, C0, bias, 0)
    assert (first.double().cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    side = torch.cuda.Stream()
    other = torch.empty_like(C0)
    outs = [torch.empty_like(C0) for _ in range(40)]
    for i, o in enumerate(outs):
        if i % 2:
            with torch.cuda.stream(side):
                product(other)
        product(o)
    torch.cuda.synchronize()
    assert all(torch.equal(o, first) for o in outs) and torch.equal(other, first)
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        product(other)                       # eager first use on the capture stream
        graph = torch.cuda.CUDAGraph()
        cap = torch.empty_like(C0)
        with torch.cuda.graph(graph, stream=cs):
            product(cap)
        for _ in range(3):
            graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap, first)
