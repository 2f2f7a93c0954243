"""las.layers -- the reference's layer library (reference las/layers.py) on the MI355X engine.

Same names and call signatures as the reference (`blstm`, `pBLSTMLayer`, `BaseAttention`,
`AdditiveAttention`, `LocationAwareAttention`, ...), eager execution on ROCm tensors; every
contraction / recurrence / attention step runs in liblas_hip.so through `las._hip` (no CPU
fallback).  TF-1.13 variable scoping becomes names in `las.variables`.

Two module-level knobs select what the reference cannot express:
    set_cell('rnn'|'lstm')   'rnn' = tf.contrib.rnn.BasicRNNCell, what the reference actually builds
                             (reference las/layers.py:31); 'lstm' = BasicLSTMCell (north-star).
    set_precision('f32'|'bf16')  arithmetic of the contractions (see include/las_hip.h).
"""
import os

import torch

from las import _hip
from las import variables as V

_CFG = {"cell": "rnn", "prec": "f32"}


def set_cell(cell):
    assert cell in ("rnn", "lstm")
    _CFG["cell"] = cell


def set_precision(prec):
    assert prec in ("f32", "bf16")
    _CFG["prec"] = prec


def get_cell():
    return _CFG["cell"]


def _prec():
    return _hip.PREC_BF16 if _CFG["prec"] == "bf16" else _hip.PREC_F32


WGRAD_ONE_PASS = os.environ.get("LAS_WGRAD_ONE_PASS", "1") != "0"


def _cellid(cell):
    return _hip.CELL_LSTM if cell == "lstm" else _hip.CELL_RNN


def cell_scope(cell):
    return "basic_lstm_cell" if cell == "lstm" else "basic_rnn_cell"


# ------------------------------------------------------------------------------------------------
# autograd nodes (coarse: one node per layer, not per time step)
# ------------------------------------------------------------------------------------------------
class _Dense(torch.autograd.Function):
    """tf.layers.dense on the last axis (+ optional tanh): y = act(x.W + b)."""

    @staticmethod
    def forward(ctx, x2d, W, b, act, prec):
        M, K = x2d.shape
        N = W.shape[1]
        y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
        _hip.gemm(prec, x2d, W, y, False, False, M, N, K, K, N, N, bias=b, act=_hip.ACT_TANH if act else _hip.ACT_NONE)
        ctx.save_for_backward(x2d, W, y)
        ctx.act, ctx.prec, ctx.has_b = act, prec, b is not None
        ctx.params = (_PARAMS.get("W"), _PARAMS.get("b"))
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, W, y = ctx.saved_tensors
        M, K = x2d.shape
        N = W.shape[1]
        dy = dy.contiguous()
        if ctx.act:
            dpre = torch.empty_like(dy)
            _hip.tanh_bwd(y, N, dy, N, dpre, N, M, N)
        else:
            dpre = dy
        dx = None
        if ctx.needs_input_grad[0]:                    # on the dependency chain: main stream, first
            dx = torch.empty_like(x2d)
            _hip.gemm(ctx.prec, dpre, W, dx, False, True, M, K, N, N, N, K)
        Wp, bp = ctx.params
        _hip.run_deferred()          # side-stream work queued by the previous node: its host cost lands here, off the chain
        if _direct_ok(Wp) and (bp is None or _direct_ok(bp)):
            # off the chain: accumulate straight into the flat gradient bucket on the side stream, overlapping
            # with the next layer's BPTT sweep (which occupies only a few dozen CUs)
            with _hip.on_side_stream():
                for t in (x2d, dpre):
                    t.record_stream(_hip.side_stream())
                _hip.gemm(ctx.prec, x2d, dpre, Wp.grad, True, False, K, N, M, K, N, N, beta=1.0)
                if bp is not None:
                    _hip.colsum(dpre, M, N, N, bp.grad, beta=1.0)
            return dx, None, None, None, None
        dW = torch.empty_like(W)
        _hip.gemm(ctx.prec, x2d, dpre, dW, True, False, K, N, M, K, N, N)
        db = None
        if ctx.has_b:
            db = torch.empty(N, device=dy.device, dtype=torch.float32)
            _hip.colsum(dpre, M, N, N, db)
        return dx, dW, db, None, None


_TANH_OUT = {}        # data_ptr -> bf16 output of a dense+tanh layer (speed mode): a recurrent layer that reads exactly this tensor
                      # fuses the Tanh gradient into its dX product (las_gemm_kk_tanhgrad) and hands dPre, not dY, to the dense node
_DPRE = set()         # data_ptr of gradients that already ARE d(pre-activation) of the dense layer they flow into
_EXPECT_DPRE = set()  # data_ptr of dense+tanh outputs whose consumer promised to hand back dPre (checked in _Dense16.backward: a
                      # gradient that arrives there by any other route -- accumulation from a second consumer, a hook, retain_grad --
                      # would get 1 - y^2 applied twice without this check)
_DCHUNK = {}          # data_ptr of a dPre whose producer (a recurrent layer's dX product) has only run its first time chunk:
                      # (chunk rows, rows per utterance, chunks, fn(k), holder) -- the dense node below runs fn(k) interleaved with its
                      # own chunks and puts the event behind the last one into `holder`
_DOUT_CHUNKS = {}     # data_ptr of a dense layer's dX that is still being produced in chunks on the chain stream: (flag, chunk rows, rows)
_XCHUNK = {}          # data_ptr of a dense + tanh output (pBLSTMLayer, forward) of which only the first time chunk exists: (chunk steps,
                      # frames per utterance, chunks, fn(k), tensors to keep alive on the side stream) -- the recurrent layer that consumes it runs fn(k) in front of chunk k of its
                      # own x-projection (round 4: the dense product between two sweeps was 65 us of the forward chain per pyramid level)
_PARAMS = {}          # hand-over of the leaf parameter objects to the autograd node being built (same thread, immediate)
import os
ROW_T = [None]        # inference over a batch of utterances of DIFFERENT lengths: int32 [B] device tensor of the current layer's frames per row
                      # (set around the listener call by BeamSearch.decode_batch; the pyramid layers halve it) -- las_rnn_seq_fwd_rows
XPROJ_CHUNK_STEPS = int(os.environ.get("LAS_XPROJ_CHUNK", "64"))     # 0: the whole x-projection before the sweep
DENSE_CHUNK_MAX_ROWS = 64
DENSE_CHUNKS = os.environ.get("LAS_DENSE_CHUNK", "1") != "0"         # the dense + tanh in front of a chunked x-projection follows the same chunks
DOUT_CHUNK_ROWS = int(os.environ.get("LAS_DOUT_CHUNK", "64"))        # backward hand-over in chunks of this many rows (a power of two); 0 = off
FUSE_TANH_GRAD = not os.environ.get("LAS_NO_FUSE_TANH_GRAD")
TAIL_ONE_LAUNCH = os.environ.get("LAS_TAIL_ONE_LAUNCH", "1") != "0"   # bottom layer's weight gradients: both directions in one launch on the launch stream
# Round 5, measured and NOT on by default (profiles/r5_tail_follow.txt): the bottom layer's weight gradients in windows that follow the last BPTT
# sweep on the side stream hide the 240 us tail launch -- and cost the sweep as much as they hide (14.04-14.15 ms per step with 3 ... 8
# windows against 14.05-14.09 with the one launch behind the sweep; with 4-byte agent-scope dZ stores instead of one L2 write-back per
# publication the sweep itself ran 1.77 -> 2.09 ms: 14.25).  LAS_TAIL_WINDOW=160 switches the windows on.
TAIL_WINDOW = int(os.environ.get("LAS_TAIL_WINDOW", "0"))              # ... contracted in windows of this many sweep steps (0 = the whole sequence at once);
TAIL_FOLLOW = os.environ.get("LAS_TAIL_FOLLOW", "1") != "0"            # ... that FOLLOW the running BPTT sweep on the side stream (round 5)
TAIL_WINDOW_WGS = int(os.environ.get("LAS_TAIL_WINDOW_WGS", "192"))    # workgroups of a window that runs beside the sweep (all but the last)
TAIL_TWO_STREAMS = not os.environ.get("LAS_NO_TAIL_TWO_STREAMS")   # bottom layer's weight gradients: one direction per auxiliary stream
HOLD_SIDE = not os.environ.get("LAS_NO_HOLD_SIDE")   # side-stream weight gradients wait for the next sweep to be resident
BEFORE_TAIL_HOOK = [None]   # callable(bottom layer's parameters) run right before the end-of-step tail is enqueued (data parallel)
DIRECT_GRADS = True   # weight gradients accumulate into the flat bucket on a side stream (needs a flattened store)


def _direct_ok(p):
    """True when p is a flattened leaf parameter whose .grad is a view of the flat gradient bucket."""
    return (DIRECT_GRADS and p is not None and p.is_leaf and p.requires_grad and p.grad is not None
            and V.default_store().flat_grad is not None and p.grad.is_contiguous())


def _xproj_chunk_steps(B, T, H, cell):
    """Sweep steps per time chunk of a layer's x-projection hand-over (0: the whole projection before the sweep)."""
    cs = XPROJ_CHUNK_STEPS
    if ROW_T[0] is not None:
        return 0                                     # rows of different lengths (inference): whole x-projection, las_rnn_seq_fwd_rows
    if cs and T >= 4 * cs and _hip.rnn_seq_fwd_chunks_ok(_cellid(cell), _hip.PREC_BF16, B, H):
        return cs
    return 0


def check_handovers_consumed():
    """Called when a backward pass is complete (LAS.train): every chunked hand-over registered by a producer must have been
    taken by its consumer -- a left-over entry means some node read a tensor whose later chunks were never computed."""
    left = [n for n, r in (("_DCHUNK", _DCHUNK), ("_DOUT_CHUNKS", _DOUT_CHUNKS), ("_XCHUNK", _XCHUNK)) if r]
    if left:
        _DCHUNK.clear()
        _DOUT_CHUNKS.clear()
        _XCHUNK.clear()
        raise RuntimeError("las.layers: chunked gradient hand-over left unconsumed (%s): the autograd graph between two recurrent "
                           "layers is not the pBLSTMLayer stack -- set LAS_DOUT_CHUNK=0 for such graphs" % ", ".join(left))


def dense(x, W, b=None, tanh=False, out_f32=True):
    """x [..., K] -> [..., N]  (tf.layers.dense on a rank-3 input = same weights at every t, App. A.13).
    bf16 activations (speed-mode listener) go through las_gemm_kk; out_f32=False keeps the result in bf16 too."""
    _hip.require_gpu(x, W)
    shp = x.shape
    x2d = x.reshape(-1, shp[-1])
    if not x2d.is_contiguous():
        x2d = x2d.contiguous()
    _PARAMS["W"], _PARAMS["b"] = W, b
    if x2d.dtype == torch.bfloat16 and W.shape[1] % 4 == 0:
        y = _Dense16.apply(_as_bf16_operand(x2d), W, b, bool(tanh), bool(out_f32))
        if tanh and not out_f32 and y.requires_grad:
            _TANH_OUT[y.data_ptr()] = y
    else:
        y = _Dense.apply(x2d.float() if x2d.dtype != torch.float32 else x2d, W, b, bool(tanh), _prec())
    _PARAMS.clear()
    return y.view(*shp[:-1], W.shape[1])


class _BLSTM(torch.autograd.Function):
    """One bidirectional recurrent layer: K1 input projection + K2 persistent sweep (+ K2b BPTT).

    x_bw: optional second input block for the backward direction (training with input dropout: the reference wraps
    fw_cell and bw_cell in separate DropoutWrappers, las/layers.py:37-42, so the two directions see independently
    masked inputs).  None = both directions read x (one product over the concatenated weights)."""

    @staticmethod
    def forward(ctx, x, kfw, bfw, kbw, bbw, cell, prec, H, pad_even, x_bw=None):
        B, T, I = x.shape
        G = 4 if cell == "lstm" else 1
        GH = G * H
        dev = x.device
        x = x.contiguous()
        two = x_bw is not None
        if two:
            x_bw = x_bw.contiguous()
        Ik = I
        if I % 4 and (I + 3) // 4 * 4 <= I + H:
            # MFCC-39: pad the operand with zero columns up to a multiple of 4 so the contraction takes the branch-free
            # MFMA path (16-byte loads) -- in both precisions since round 5 (the parity mode's exact-fp32 kernels have one too; a zero
            # times a weight is an exact zero at the end of the fma chain: bit-identical).  The extra weight rows it meets are the first rows of W_hh: multiplied by zeros
            # here, and given an exactly-zero gradient contribution in backward.
            Ik = (I + 3) // 4 * 4
            x = torch.nn.functional.pad(x, (0, Ik - I))
            if two:
                x_bw = torch.nn.functional.pad(x_bw, (0, Ik - I))
        gates = torch.empty(B, T, 2, GH, device=dev, dtype=torch.float32)
        if not two:
            # x_t . W_ih + bias for all t at once (W_ih = first I rows of the TF kernel [(I+H), G*H]); both directions in
            # ONE product over the concatenated weights, so x is read once
            kcat = torch.cat((kfw[:Ik], kbw[:Ik]), 1)                   # [Ik, 2*GH]
            _hip.gemm(prec, x, kcat, gates, False, False, B * T, 2 * GH, Ik, Ik, 2 * GH, 2 * GH, bias=torch.cat((bfw, bbw)))
        else:
            for d, (xd, k, b) in enumerate(((x, kfw, bfw), (x_bw, kbw, bbw))):
                _hip.gemm(prec, xd, k, gates, False, False, B * T, GH, Ik, Ik, GH, 2 * GH, bias=b, c_off=d * GH)
        Tp = T + (T % 2) if pad_even else T
        out = torch.zeros(B, Tp, 2 * H, device=dev) if Tp != T else torch.empty(B, T, 2 * H, device=dev)
        cst = torch.empty(B, T, 2, H, device=dev) if cell == "lstm" else None
        _hip.rnn_seq_fwd(_cellid(cell), prec, B, T, H, gates, kfw, kbw, GH, out, 2 * H, Tp * 2 * H, cst,
                         1.0, wf_off=I * GH, wb_off=I * GH)
        ctx.save_for_backward(x, kfw, kbw, gates, out, cst, x_bw)
        ctx.cfg = (cell, prec, H, Tp, I)
        ctx.params = _PARAMS.get("blstm")
        return out

    @staticmethod
    def backward(ctx, dout):
        x, kfw, kbw, gates, out, cst, x_bw = ctx.saved_tensors
        cell, prec, H, Tp, I0 = ctx.cfg
        B, T, I = x.shape                              # I = operand width (I0 padded to a multiple of 4 in speed mode)
        G = 4 if cell == "lstm" else 1
        GH = G * H
        dev = x.device
        dout = dout.contiguous()
        two = x_bw is not None
        xs = (x, x_bw if two else x)
        # gates: activated gates -> d(pre-activation), in place
        P4 = ctx.params
        direct = P4 is not None and all(_direct_ok(p) for p in P4)
        # with direct gradients the sweep itself accumulates the two bias gradients (column sums of dZ held in registers) -- the speed
        # mode's kernels do; behind the parity mode's sweeps the library would run the column sums on THIS stream (0.25 ms per layer on
        # the dependency chain, round 5): they go to the side stream with the weight gradients instead (same kernel, same order)
        db_in_sweep = direct and prec != _hip.PREC_F32
        _hip.rnn_seq_bwd(_cellid(cell), prec, B, T, H, gates, kfw, kbw, GH, out, 2 * H, Tp * 2 * H, cst,
                         dout, 2 * H, Tp * 2 * H, 1.0, wf_off=I0 * GH, wb_off=I0 * GH,
                         db_fw=P4[1].grad if db_in_sweep else None, db_bw=P4[3].grad if db_in_sweep else None)
        dx = dx_bw = None
        if ctx.needs_input_grad[0] and not two:        # on the dependency chain: main stream, first
            # dZ [B*T, 2*GH] . [W_ih_fw | W_ih_bw]^T in one product (one pass over dx instead of two)
            dx = torch.empty(B, T, I0, device=dev)
            kcat = torch.cat((kfw[:I0], kbw[:I0]), 1)               # [I0, 2*GH]
            _hip.gemm(prec, gates, kcat, dx, False, True, B * T, I0, 2 * GH, 2 * GH, 2 * GH, I0)
        elif two and (ctx.needs_input_grad[0] or ctx.needs_input_grad[9]):
            dx, dx_bw = torch.empty(B, T, I0, device=dev), torch.empty(B, T, I0, device=dev)
            for d, (dxd, k) in enumerate(((dx, kfw), (dx_bw, kbw))):
                _hip.gemm(prec, gates, k, dxd, False, True, B * T, I0, GH, 2 * GH, GH, I0, a_off=d * GH)
        _hip.run_deferred()
        if direct:
            # weight gradients: off the chain -> side stream, accumulated straight into the flat gradient bucket
            def wgrad(d, strm):
                for t in (x, gates, out) + ((x_bw,) if two else ()):
                    t.record_stream(strm)
                kp, bp = P4[2 * d], P4[2 * d + 1]
                gk = kp.grad                       # [(I+H), GH] view of the flat bucket
                if not db_in_sweep:
                    _hip.colsum(gates, B * T, GH, 2 * GH, bp.grad, x_off=d * GH, beta=1.0)
                _hip.gemm(prec, xs[d], gates, gk, True, False, I, GH, B * T, I, 2 * GH, GH, beta=1.0, b_off=d * GH)
                if T > 1:
                    part = torch.empty(B, H, GH, device=dev)
                    a_off = d * H + (0 if d == 0 else 2 * H)
                    b_off = d * GH + (2 * GH if d == 0 else 0)
                    _hip.gemm(prec, out, gates, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                              strideA=Tp * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=a_off, b_off=b_off)
                    _hip.colsum(part, B, H * GH, H * GH, gk[I0:].reshape(-1), beta=1.0)

            # the bottom layer's weight gradients are the end-of-step tail (nothing left to hide behind): one direction per auxiliary
            # stream, as in the speed mode (its products do not fill the chip alone); the other layers' run on the side stream
            tail2 = TAIL_TWO_STREAMS and not ctx.needs_input_grad[0] and not two
            with _hip.on_side_stream():
                wgrad(0, _hip.side_stream())
                if not tail2:
                    wgrad(1, _hip.side_stream())
            if tail2:
                with _hip.on_chain_stream():
                    wgrad(1, _hip.chain_stream())
            return (dx, None, None, None, None, None, None, None, None, dx_bw)
        grads = []
        part = torch.empty(B, H, GH, device=dev) if T > 1 else None
        for d, k in enumerate((kfw, kbw)):
            dk = torch.empty_like(k)
            # dW_ih = x^T . dG_d        (contraction over all B*T frames; split-K inside las_gemm)
            _hip.gemm(prec, xs[d], gates, dk, True, False, I, GH, B * T, I, 2 * GH, GH, b_off=d * GH)
            # dW_hh = sum_b sum_t h_prev^T . dG_d : per-utterance products (fw pairs h[t-1] with dG[t],
            # bw pairs h[t+1] with dG[t]), then a fixed-order sum over utterances
            if T > 1:
                a_off = d * H + (0 if d == 0 else 2 * H)
                b_off = d * GH + (2 * GH if d == 0 else 0)
                _hip.gemm(prec, out, gates, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                          strideA=Tp * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=a_off, b_off=b_off)
                whh = torch.empty(H * GH, device=dev)
                _hip.colsum(part, B, H * GH, H * GH, whh)
                dk[I0:] = whh.view(H, GH)
            else:
                dk[I0:] = 0
            db = torch.empty(GH, device=dev)
            _hip.colsum(gates, B * T, GH, 2 * GH, db, x_off=d * GH)
            grads += [dk, db]
        return (dx, grads[0], grads[1], grads[2], grads[3], None, None, None, None, dx_bw)


# ------------------------------------------------------------------------------------------------
# speed mode with bf16 activation storage (SURVEY 8(d) algorithmic bytes): the Listener's x-projections / saved gates,
# cell states, h, dense outputs and their gradients live in HBM as bf16; the chain products run through las_gemm_kk
# (both operands bf16, contraction index contiguous) against bf16 SHADOWS of the fp32 master weights, kept in both
# orientations and rebuilt after every optimiser step (`_shadow`).  Parameters, their gradients and Adam stay fp32.
# ------------------------------------------------------------------------------------------------
def _shadow(tag, srcs, rows, transpose, dst_rows, dst_cols, bf16=True):
    """Operand copy of one or two fp32 parameters for the speed-mode products: D = zero-pad(op([S0[:rows] | S1[:rows]])) with
    op = transpose or identity, [dst_rows, dst_cols], bf16 (or fp32).  Valid until the parameters change (Adam / load / flatten
    clear `store.shadows`).  The first build runs through torch and registers the recipe; afterwards ALL registered shadows are
    rebuilt together by one las_build_shadows launch at the first request after an optimiser step."""
    import ctypes
    st = V.default_store()
    key = (tag, int(rows), bool(transpose), int(dst_rows), int(dst_cols), bool(bf16)) + tuple(int(q.data_ptr()) for q in srcs)
    sh = st.shadows.get(key)
    if sh is not None:
        return sh
    if key in st.shadow_recipes:
        recs = list(st.shadow_recipes.values())
        if st.shadow_table is None or st.shadow_table[1] != len(recs):
            arr = (_hip.ShadowDesc * len(recs))(*[r[1] for r in recs])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            mt = max(((r[1].dst_rows + 31) // 32) * ((r[1].dst_cols + 31) // 32) for r in recs)
            st.shadow_table = (host.to(recs[0][0].device), len(recs), mt)
        tab, n, mt = st.shadow_table
        _hip.check(_hip.lib().las_build_shadows(tab.data_ptr(), n, mt, _hip.stream()), "las_build_shadows")
        for k, r in st.shadow_recipes.items():
            st.shadows[k] = r[0]
        _prepare_sweeps(st)              # ... and, with the same trigger, everything the step's sweeps need from the weights
        return st.shadows[key]
    with torch.no_grad():
        mats = [(q.detach().reshape(1, -1) if q.dim() == 1 else q.detach())[:rows] for q in srcs]
        Lm = torch.cat(mats, 1) if len(mats) > 1 else mats[0]
        D = Lm.t() if transpose else Lm
        sh = torch.zeros(dst_rows, dst_cols, device=D.device, dtype=torch.bfloat16 if bf16 else torch.float32)
        sh[:D.shape[0], :D.shape[1]] = D
    d = _hip.ShadowDesc()
    d.src0 = mats[0].data_ptr(); d.ld0 = mats[0].stride(0) if mats[0].shape[0] > 1 else mats[0].shape[1]
    d.cols0 = mats[0].shape[1]; d.rows = mats[0].shape[0]
    if len(mats) > 1:
        assert mats[1].shape[0] == mats[0].shape[0]
        d.src1 = mats[1].data_ptr(); d.ld1 = mats[1].stride(0) if mats[1].shape[0] > 1 else mats[1].shape[1]; d.cols1 = mats[1].shape[1]
    else:
        d.src1 = None; d.ld1 = 0; d.cols1 = 0
    d.transpose = int(bool(transpose)); d.dst = sh.data_ptr(); d.dst_rows, d.dst_cols, d.dst_ld = dst_rows, dst_cols, dst_cols
    d.dst_bf16 = int(bool(bf16))
    st.shadow_recipes[key] = (sh, d, srcs)          # (srcs kept alive: the descriptor holds their addresses)
    st.shadow_table = None
    st.shadows[key] = sh
    return sh


PREPARED_SWEEPS = os.environ.get("LAS_NO_PREPARED_SWEEPS") != "1"
_STEP_START = [None]  # event on the launch stream at the start of the current train step (begin_step)
_STEP_ROWS = [0]      # batch rows of the recurrent layer that is being built (the prepared workspaces are laid out for it)


def _prepare_sweeps(st):
    """Round 5: the weight packs and exchange-state clears of ALL the recurrent sweeps of a training step -- forward and BPTT of every layer
    seen in the previous step -- in ONE launch (las_rnn_seq_prepare), at the first request after the weights changed.  Until round 4 every
    sweep launched its own pack kernel in front of itself: 8 launches of 8-11 us on the dependency chain per step."""
    if not (PREPARED_SWEEPS and st.seq_recipes and _STEP_ROWS[0] > 0):
        return
    B = _STEP_ROWS[0]
    jobs = []
    for key, r in st.seq_recipes.items():
        need = int(_hip.lib().las_rnn_seq_workspace_bytes(r["cell"], _hip.PREC_BF16, r["H"], B))
        if r["ws"] is None or r["ws"].numel() < need:
            r["ws"] = torch.empty(need, dtype=torch.uint8, device=r["wf"].device)      # (lives as long as the recipe: never recycled by the allocator)
        jobs.append((r["cell"], r["H"], B, key[-1], r["wf"], r["wb"], r["ldw"], r["off"], r["off"], r["ws"]))
    # on the side stream, ordered behind the START of the step (= behind the optimiser step that changed the weights), beside the step's
    # head on the launch stream: ~65 MB of packs and clears that nothing on the chain waits for until the first sweep
    # (the event only orders the prepare behind the weight change of the epoch it was recorded in.  A step whose own
    #  begin_step event was never consumed -- a store's first step builds its shadows one by one -- leaves it behind; a request from
    #  OUTSIDE a train step (eval or decode right after that step) must not use it: the prepare would pack W_hh in front of
    #  las_clip_adam.  A stale or missing event is replaced by one recorded NOW on the launch stream, i.e. behind the optimiser.)
    ev, epoch = _STEP_START[0] if _STEP_START[0] is not None else (None, -1)
    _STEP_START[0] = None
    if _hip.streams_overlap(jobs[0][4].device):
        if ev is None or epoch != st.weights_epoch:
            ev = torch.cuda.Event()
            ev.record()
            VARIANTS["prepare_behind_now"] += 1
        with _hip.on_side_stream(after=ev):
            for j in jobs:
                j[-1].record_stream(_hip.side_stream())       # (allocated on the launch stream, written here)
            _hip.rnn_seq_prepare(jobs)
            st.seq_prep_done = torch.cuda.Event()
            st.seq_prep_done.record()
    else:
        _hip.rnn_seq_prepare(jobs)
        st.seq_prep_done = None
    for key in st.seq_recipes:
        st.seq_ready[key] = B


def _prepared_ws(kfw, kbw, cell, H, GH, off, B, bwd):
    """The workspace las_rnn_seq_prepare has made ready for this sweep (None: the sweep packs for itself); registers the sweep so
    that the NEXT step prepares it.  Training steps only (gradient mode, rows of one length)."""
    if not PREPARED_SWEEPS or ROW_T[0] is not None:
        return None
    st = V.default_store()
    key = (int(kfw.data_ptr()), int(kbw.data_ptr()), cell, int(H), int(bool(bwd)))
    r = st.seq_recipes.get(key)
    if r is None:
        st.seq_recipes[key] = {"cell": _cellid(cell), "H": int(H), "wf": kfw, "wb": kbw, "ldw": int(GH), "off": int(off), "ws": None}
        return None
    have = st.seq_ready.pop(key, 0)          # consumed: the exchange state is dirty after one sweep
    if have >= B and r["ws"] is not None:
        VARIANTS["prepared_sweeps"] += 1
        if getattr(st, "seq_prep_done", None) is not None:          # the step's first prepared sweep orders the launch stream behind the prepare
            torch.cuda.current_stream().wait_event(st.seq_prep_done)
            st.seq_prep_done = None
        return r["ws"]
    return None


_CHUNK_FLAGS = {}


def _chunk_flag(dev):
    """A zeroed device word for the completion count of one layer's x-projection chunks (a small ring: a word is reused only
    after several later sweeps have been enqueued behind the one that reads it)."""
    ring = _flag_ring(dev)
    i = ring[1] % _RING
    ring[1] += 1
    w = ring[0][i:i + 1]
    if ring[2] > 0:
        ring[2] -= 1                                     # zeroed with the whole ring by begin_step
    else:
        VARIANTS["flag_fills"] += 1                      # (a fill on the chain: more hand-overs in one step than the ring holds, or no begin_step)
        w.zero_()
    return w


_RING = 32
_PROG = 512          # progress words of a BPTT sweep that publishes how far its dZ has reached memory (behind the ring: one fill zeroes both)


def _progress_words(dev, n):
    """n zeroed device words for las_rnn_seq_bwd_db_progress (zeroed with the flag ring by begin_step; by a fill otherwise)"""
    ring = _flag_ring(dev)
    w = ring[0][_RING:_RING + n]
    if len(ring) > 3 and ring[3]:
        ring[3] = False
    else:
        VARIANTS["flag_fills"] += 1
        w.zero_()
    return w


def _flag_ring(dev):
    # (r4: this used to be `_CHUNK_FLAGS.setdefault(key, [torch.zeros(...), 0, 0])` -- Python evaluates the default on EVERY call, i.e. one
    #  allocation + one 5-7 us fill kernel per hand-over word, eleven per step, seven of them on the dependency chain)
    key = str(dev)
    ring = _CHUNK_FLAGS.get(key)
    if ring is None:
        ring = _CHUNK_FLAGS[key] = [torch.zeros(_RING + _PROG, dtype=torch.int32, device=dev), 0, 0, False]
    return ring


# Which schedule variants the current step used (LAS.train copies it into `las.last_variants`): sweeps launched with their x-projection
# still arriving in chunks (-> rnn_seq_fwd_hw_kernel's chunk waits), BPTT sweeps launched on an upstream gradient still arriving in
# chunks (-> the rnn_seq_bwd_ks_kernel<..., CH = true> instance), side-stream weight-gradient groups held until the next sweep is
# resident; "serial": hand-overs that ran with their producers in front of the consumer on one stream (LAS_ALLOW_SERIAL_STREAMS=1 under a
# tool that serialises kernels -- the same kernel instances, nothing overlapped).  Tests and bench.py assert / print it: the variant
# that is timed must be the variant that is tested.
VARIANTS = {"xproj_chunks": 0, "dense_chunks": 0, "dout_chunks": 0, "hold_side": 0, "sweeps_fwd": 0, "sweeps_bwd": 0, "serial": 0, "flag_fills": 0,
            "prepared_sweeps": 0, "tail_windows": 0, "tail_follow": 0, "prepare_behind_now": 0}     # prepared_sweeps: sweeps that found their pack + clean exchange state ready (las_rnn_seq_prepare; from a model's second step on: all)


import contextlib


@contextlib.contextmanager
def fallback_schedule(on=True):
    """The schedule LAS._recover re-runs a lost step on: nothing in it needs workgroups of DIFFERENT launches (or more workgroups than a
    busy device can place at once) to be co-resident -- the Speller as per-step launches (LAS_SPELLER_NO_FUSED_STEP: the prefetching rows /
    the wide path, same arithmetic family as the loop kernels), every x-projection and upstream gradient complete in front of its sweep
    (no chunk hand-overs across streams), no held side stream, sweeps packing for themselves.  The recurrent sweeps themselves keep
    their clusters (2-8 workgroups each on 48-60 of the 256 CUs: they fit beside a neighbour; the hand-overs are what a stalled
    partner stream breaks)."""
    global XPROJ_CHUNK_STEPS, DOUT_CHUNK_ROWS, HOLD_SIDE, PREPARED_SWEEPS, TAIL_WINDOW
    if not on:
        yield
        return
    saved = (XPROJ_CHUNK_STEPS, DOUT_CHUNK_ROWS, HOLD_SIDE, PREPARED_SWEEPS, TAIL_WINDOW, _hip.speller_flags)
    XPROJ_CHUNK_STEPS, DOUT_CHUNK_ROWS, HOLD_SIDE, PREPARED_SWEEPS, TAIL_WINDOW = 0, 0, False, False, 0
    _hip.speller_flags = _hip.speller_flags | _hip.SPELLER_NO_FUSED_STEP
    try:
        yield
    finally:
        XPROJ_CHUNK_STEPS, DOUT_CHUNK_ROWS, HOLD_SIDE, PREPARED_SWEEPS, TAIL_WINDOW, _hip.speller_flags = saved


def begin_step(dev):
    """Called by LAS.train at the start of a step (all streams of the previous step joined): ONE fill zeroes the whole ring of
    hand-over words instead of one 5 us fill in front of every sweep and every chunked dense product (11 per step, on the chain)."""
    for k in VARIANTS:
        VARIANTS[k] = 0
    if _prec() == _hip.PREC_BF16:
        _hip.streams_overlap(dev)       # probed (once) HERE: a device whose streams cannot overlap raises before any hand-over state exists
    ring = _flag_ring(dev)
    ring[0].zero_()
    ring[1], ring[2], ring[3] = 0, _RING, True
    ev = torch.cuda.Event()
    ev.record()
    _STEP_START[0] = (ev, V.default_store().weights_epoch)


def _k64(k):
    return (k + 63) // 64 * 64


def _as_bf16_operand(x):
    """[..., K] -> bf16 with K padded by zero columns to a multiple of 64 (las_gemm_kk tiles); differentiable."""
    K = x.shape[-1]
    if K % 64:
        x = torch.nn.functional.pad(x, (0, _k64(K) - K))
    x = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
    return x.contiguous()                                   # (.to() keeps the strides of a transposed view)


class _Dense16(torch.autograd.Function):
    """tf.layers.dense (+ tanh) on bf16 activations: y = act(x.W + b), y bf16 or fp32 (`out_f32`)."""

    @staticmethod
    def forward(ctx, x2d, W, b, act, out_f32):
        M, K = x2d.shape                               # K = padded width of the operand (>= W.shape[0])
        Kw, N = W.shape
        WT = _shadow("denseT", (W,), Kw, True, N, _k64(Kw))                              # W^T: [N, K64]
        y = torch.empty(M, N, device=x2d.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        xc = _PARAMS.pop("xchunk", None)              # (cs, B, T): the consumer is a recurrent layer that takes its input in time chunks
        if xc is not None and not out_f32 and xc[1] * xc[2] == M:
            cs, nb, Tq = xc
            th = (Tq + 1) // 2
            nch = (th + cs - 1) // cs
            a = _hip.ACT_TANH if act else _hip.ACT_NONE

            def produce(k):
                lo0, lo1 = k * cs, min((k + 1) * cs, th)
                hi0, hi1 = max(Tq - lo1, lo1), Tq - lo0
                _hip.gemm_kk_frames(x2d, WT, y, nb, Tq, lo0, lo1 - lo0, hi0, hi1 - hi0, N, K, K, K, N, bias=b, act=a)

            produce(0)
            _XCHUNK[y.data_ptr()] = (cs, Tq, nch, produce, (x2d, WT, y) + ((b,) if b is not None else ()))
        else:
            _hip.gemm_kk(x2d, WT, y, M, N, K, K, K, N, bias=b, act=_hip.ACT_TANH if act else _hip.ACT_NONE)
        ctx.save_for_backward(x2d, W, y)
        ctx.act, ctx.has_b = act, b is not None
        ctx.params = (_PARAMS.get("W"), _PARAMS.get("b"))
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, W, y = ctx.saved_tensors
        M, K = x2d.shape
        Kw, N = W.shape
        dy = dy.contiguous()
        if ctx.act and y.data_ptr() in _EXPECT_DPRE:
            _EXPECT_DPRE.discard(y.data_ptr())
            if not (dy.dtype == torch.bfloat16 and dy.data_ptr() in _DPRE):
                raise RuntimeError("las.layers: the recurrent layer that consumed this dense+tanh output fused the Tanh gradient into "
                                   "its dX product, but the gradient arriving here is not that tensor (a second consumer, a tensor "
                                   "hook or retain_grad on the activation?) -- set LAS_NO_FUSE_TANH_GRAD=1 for such graphs")
            _DPRE.discard(dy.data_ptr())               # the consumer's dX product already applied 1 - y^2
            dpre = dy
        elif ctx.act:
            dpre = torch.empty(M, N, device=dy.device, dtype=torch.bfloat16)
            _hip.tanh_bwd(y, N, dy, N, dpre, N, M, N)
        else:
            dpre = dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)
        dx = None
        ch = _DCHUNK.pop(dy.data_ptr(), None) if dpre is dy else None
        chain_done = None
        if ch is not None and not (ctx.needs_input_grad[0] and _k64(N) == N):
            for k in range(1, ch[2]):                  # no hand-over here: finish the producer's product first (this stream)
                ch[3](k)
            ch = None
        if ctx.needs_input_grad[0]:                    # on the dependency chain: main stream, first
            Wb = _shadow("dense", (W,), Kw, False, K, _k64(N))                           # W, rows padded to K: [K, N64]
            dx = torch.empty(M, K, device=dy.device, dtype=torch.bfloat16)
            Nk = Wb.shape[1]
            if Nk != N:                                # contraction width padded to 64: pad dpre too (rare: N % 64 != 0)
                dpre_k = torch.nn.functional.pad(dpre, (0, Nk - N))
            else:
                dpre_k = dpre
            if ch is not None:
                # Wavefront hand-over to the BPTT sweep below (the mirror of the chunked x-projection): dPre arrives in time
                # chunks from both ends of the sequence, this product follows chunk by chunk -- chunk 0 here, the others on the
                # chain stream while the sweep already runs -- and publishes its progress in `flag`.
                c, Tq, nch, produce, holder = ch
                nb, th = M // Tq, (Tq + 1) // 2
                flag = _chunk_flag(dy.device)

                def mine(k):
                    lo0, lo1 = k * c, min((k + 1) * c, th)
                    hi0, hi1 = max(Tq - lo1, lo1), Tq - lo0
                    _hip.gemm_kk_frames(dpre, Wb, dx, nb, Tq, lo0, lo1 - lo0, hi0, hi1 - hi0, K, N, N, N, K)
                    if k > 0:                          # (chunk 0 precedes the consumer sweep in stream order: never waited for)
                        _hip.set_word(flag, k + 1)

                mine(0)
                first = torch.cuda.Event()
                first.record()
                chain_done = []

                def rest():
                    # enqueued by the consumer AFTER it has launched its sweep (the host must not spend the 3 launches per chunk
                    # in front of that launch); at the latest by the next run_deferred()
                    if chain_done:
                        return
                    if _hip.streams_overlap(dy.device):
                        with _hip.on_chain_stream(after=first):
                            for t in (dpre, dx, flag):
                                t.record_stream(_hip.chain_stream())
                            for k in range(1, nch):
                                produce(k)
                                mine(k)
                            chain_done.append(torch.cuda.Event())
                            chain_done[0].record()
                    else:                              # serialised streams: same products, this stream, in front of the consumer
                        for k in range(1, nch):
                            produce(k)
                            mine(k)
                        chain_done.append(torch.cuda.Event())
                        chain_done[0].record()
                    holder.append(chain_done[0])

                _DOUT_CHUNKS[dx.data_ptr()] = (flag, c, Tq, rest)
            else:
                _hip.gemm_kk(dpre_k, Wb, dx, M, K, Nk, Nk, Nk, K)
        Wp, bp = ctx.params
        if chain_done is None:
            _hip.run_deferred()                        # (else: the consumer sweep's node runs them, after its launch)
        Kg = (Kw + 3) // 4 * 4                         # rows of the weight gradient the TN product writes (zero operand columns beyond Kw)
        if _direct_ok(Wp) and (bp is None or _direct_ok(bp)) and Kg == Kw:
            def wgrad(after=None):
                with _hip.on_side_stream(after=after):
                    for t in (x2d, dpre):
                        t.record_stream(_hip.side_stream())
                    _hip.gemm(_hip.PREC_BF16, x2d, dpre, Wp.grad, True, False, Kw, N, M, K, N, N, beta=1.0)
                    if bp is not None:
                        _hip.colsum(dpre, M, N, N, bp.grad, beta=1.0)

            if chain_done is None:
                wgrad()
            else:
                def later():
                    rest()
                    wgrad(chain_done[0])           # all of dPre
                _hip.defer_side(later)
            return dx, None, None, None, None
        if chain_done is not None:
            rest()
            torch.cuda.current_stream().wait_event(chain_done[0])
        dW = torch.zeros(Kg, N, device=dy.device)
        _hip.gemm(_hip.PREC_BF16, x2d, dpre, dW, True, False, Kg, N, M, K, N, N)
        db = None
        if ctx.has_b:
            db = torch.empty(N, device=dy.device, dtype=torch.float32)
            _hip.colsum(dpre, M, N, N, db)
        return dx, dW[:Kw], db, None, None


class _BLSTM16(torch.autograd.Function):
    """One bidirectional recurrent layer with bf16 activation storage: K1 x-projection (las_gemm_kk) + K2 sweep + K2b BPTT.
    x: bf16 [B,T,Ik] (Ik = input width padded to a multiple of 64 with zero columns); I0 = true input width."""

    @staticmethod
    def forward(ctx, x, kfw, bfw, kbw, bbw, cell, H, pad_even, I0, x_bw=None):
        B, T, Ik = x.shape
        G = 4 if cell == "lstm" else 1
        GH = G * H
        dev = x.device
        prec = _hip.PREC_BF16
        x = x.contiguous()
        two = x_bw is not None
        if two:
            x_bw = x_bw.contiguous()
        bf = torch.bfloat16
        gates = torch.empty(B, T, 2, GH, device=dev, dtype=bf)
        rest_chunks = None
        training = any(ctx.needs_input_grad) and ROW_T[0] is None      # (grad mode itself is off inside an autograd node's forward)
        if training:
            _STEP_ROWS[0] = B                        # (the first shadow request after an optimiser step prepares every sweep of the step for B rows)
        if not two:
            # both directions in ONE product over the concatenated weights: B operand = shadow of [W_ih_fw | W_ih_bw]^T
            WT = _shadow("ihT", (kfw, kbw), I0, True, 2 * GH, _k64(I0))                     # [W_ih_fw | W_ih_bw]^T: [2GH, Ik]
            bias = _shadow("ihb", (bfw, bbw), 1, False, 1, 2 * GH, bf16=False).view(-1)
            chunk_flag, cs = None, _xproj_chunk_steps(B, T, H, cell)
            xc = _XCHUNK.pop(x.data_ptr(), None)         # the dense + tanh below has only produced its first time chunk
            if xc is not None and (xc[0] != cs or xc[1] != T):
                for k in range(1, xc[2]):                # not the chunks this layer takes: finish the producer first
                    xc[3](k)
                xc = None
            if cs:
                # The sweep consumes the x-projection in time order (forward direction from t = 0, backward from t = T - 1), so only
                # the first chunk of frames -- both ends of the sequence -- has to exist when it starts: chunk 0 on this stream,
                # the others on the side stream WHILE the sweep runs (it holds a fifth of the CUs); the sweep's helper waves wait
                # for a chunk's completion flag before they read its frames.
                th = (T + 1) // 2
                nch = (th + cs - 1) // cs
                chunk_flag = _chunk_flag(dev)

                def chunk(k):
                    if xc is not None and k > 0:
                        xc[3](k)                         # the input frames of this chunk (dense + tanh of the layer below)
                    lo0, lo1 = k * cs, min((k + 1) * cs, th)
                    hi0, hi1 = max(T - lo1, lo1), T - lo0
                    _hip.gemm_kk_frames(x, WT, gates, B, T, lo0, lo1 - lo0, hi0, hi1 - hi0, 2 * GH, Ik, Ik, Ik, 2 * GH, bias=bias)
                    if k > 0:                            # (chunk 0 precedes the sweep in stream order: the sweep never waits for it)
                        _hip.set_word(chunk_flag, k + 1)

                chunk(0)
                VARIANTS["dense_chunks"] += int(xc is not None)
                if _hip.streams_overlap(dev):
                    # Round 5: the SWEEP is launched right behind chunk 0 and the other chunks are enqueued behind its launch, on the side
                    # stream, ordered after chunk 0 by an event (r4 timeline: the host needed 0.36 ms to enqueue the 18 launches of layer 0's
                    # nine chunks in front of the sweep's launch -- the first sweep of a step started 547 us into it)
                    chunk0_done = torch.cuda.Event()
                    chunk0_done.record()

                    def rest_chunks(pad=None):
                        with _hip.on_side_stream(after=chunk0_done):
                            side = _hip.side_stream()
                            for t in (x, gates, chunk_flag) + (xc[4] if xc is not None else ()) + (() if pad is None else (pad,)):
                                t.record_stream(side)
                            if pad is not None:
                                pad.zero_()              # the zero frame behind an odd T (nobody reads it before the join below)
                            for k in range(1, nch):
                                chunk(k)
                else:
                    # serialised streams (LAS_ALLOW_SERIAL_STREAMS=1, counter passes): the SAME kernel instances -- chunk products and
                    # the chunk-aware sweep -- with every producer in front of its consumer on this stream
                    VARIANTS["serial"] += 1
                    for k in range(1, nch):
                        chunk(k)
            else:
                _hip.gemm_kk(x, WT, gates, B * T, 2 * GH, Ik, Ik, Ik, 2 * GH, bias=bias)
        else:
            for d, (xd, k, b) in enumerate(((x, kfw, bfw), (x_bw, kbw, bbw))):
                WTd = _shadow("ihT%d" % d, (k,), I0, True, GH, _k64(I0))                    # [GH, Ik]
                _hip.gemm_kk(xd, WTd, gates, B * T, GH, Ik, Ik, Ik, 2 * GH, bias=b.detach(), c_off=d * GH)
        Tp = T + (T % 2) if pad_even else T
        out = torch.empty(B, Tp, 2 * H, device=dev, dtype=bf)
        cst = torch.empty(B, T, 2, H, device=dev, dtype=bf) if cell == "lstm" else None
        row_T = ROW_T[0]
        if row_T is not None and (two or torch.is_grad_enabled() or not _hip.rnn_seq_fwd_rows_ok(_cellid(cell), prec, B, H)):
            raise RuntimeError("rows of different lengths (layers.ROW_T) are an inference-only path of the 8-row speed-mode sweep")
        VARIANTS["sweeps_fwd"] += 1
        VARIANTS["xproj_chunks"] += int(not two and chunk_flag is not None)
        _hip.rnn_seq_fwd(_cellid(cell), prec, B, T, H, gates, kfw, kbw, GH, out, 2 * H, Tp * 2 * H, cst,
                         1.0, wf_off=I0 * GH, wb_off=I0 * GH, chunk_flag=None if two else chunk_flag, chunk_steps=0 if two else cs,
                         row_T=row_T, prepared_ws=_prepared_ws(kfw, kbw, cell, H, GH, I0 * GH, B, False) if training else None)
        if rest_chunks is not None:
            # chunks 1 .. of the x-projection: side stream, enqueued behind the sweep's launch (with the pad frame's fill, off the chain)
            rest_chunks(out[:, T:] if Tp != T else None)
        elif Tp != T:
            out[:, T:].zero_()               # only the pad frame (the sweep writes every real frame, and never this one)
        if not two and chunk_flag is not None:
            _hip.join_side_stream()          # (the chunks are long finished; this orders later users of `gates` after them)
        ctx.save_for_backward(x, kfw, kbw, gates, out, cst, x_bw)
        ctx.cfg = (cell, H, Tp, I0)
        ctx.params = _PARAMS.get("blstm")
        ctx.in_pyramid = bool(_PARAMS.get("hold_side"))      # built by pBLSTMLayer on top of (recurrent layer -> dense + tanh)
        ctx.hold_side = ctx.in_pyramid and HOLD_SIDE
        # the input IS the (unpadded) tanh output of the dense layer below: its gradient can leave this node as dPre
        ctx.x_is_tanh = FUSE_TANH_GRAD and not two and I0 == Ik and _TANH_OUT.pop(x.data_ptr(), None) is not None
        if ctx.x_is_tanh:
            _EXPECT_DPRE.add(x.data_ptr())
        return out

    @staticmethod
    def backward(ctx, dout):
        x, kfw, kbw, gates, out, cst, x_bw = ctx.saved_tensors
        cell, H, Tp, I0 = ctx.cfg
        B, T, Ik = x.shape
        G = 4 if cell == "lstm" else 1
        GH = G * H
        dev = x.device
        prec = _hip.PREC_BF16
        bf = torch.bfloat16
        dout = dout.contiguous()
        two = x_bw is not None
        xs = (x, x_bw if two else x)
        P4 = ctx.params
        direct = P4 is not None and all(_direct_ok(p) for p in P4)
        produced = None
        dc = _DOUT_CHUNKS.pop(dout.data_ptr(), None)       # dout still arrives in chunks (the dense node above, chain stream)
        if dc is not None and not (dc[2] in (T, (T + 1) // 2) and _hip.rnn_seq_bwd_chunks_ok(_cellid(cell), prec, B, H)):
            dc[3]()
            _hip.join_chain_stream()
            dc = None
        VARIANTS["sweeps_bwd"] += 1
        VARIANTS["dout_chunks"] += int(dc is not None)
        serial = dc is not None and not _hip.streams_overlap(dev)
        if serial:
            VARIANTS["serial"] += 1
            dc[3]()                          # serialised streams: every chunk in front of the (same, chunk-aware) sweep
        # Round 5, the end-of-step tail: the bottom layer's weight gradients are contracted in WINDOWS of TAIL_WINDOW sweep steps (the
        # forward direction's frames from the end of the sequence, the backward direction's from its start: the order in which the sweep
        # produces dZ), and -- when the sweep can publish its progress -- each window runs on the side stream as soon as the sweep has
        # passed it, instead of all of them behind the sweep (a 240 us launch at the bench geometry).  The windows are the ARITHMETIC
        # (fixed by T and TAIL_WINDOW); following the sweep is only the schedule: without it the same launches run behind the sweep.
        one_pass_ = WGRAD_ONE_PASS and T > 1 and H % 128 == 0 and GH % 128 == 0 and B * T < (1 << 24) and out.dtype == bf and x.dtype == bf
        is_tail = (P4 is not None and all(_direct_ok(p) for p in P4) and not ctx.needs_input_grad[0] and not two and one_pass_ and TAIL_ONE_LAUNCH)
        windows = is_tail and TAIL_WINDOW > 0 and T >= 3 * TAIL_WINDOW
        follow, prog, nprog = False, None, 0
        if windows and TAIL_FOLLOW and dc is not None and not serial:
            nprog = _hip.rnn_seq_bwd_progress_words(_cellid(cell), prec, B, H)
            if 0 < nprog <= _PROG:
                follow, prog = True, _progress_words(dev, nprog)
                before_sweep = torch.cuda.Event()
                before_sweep.record()
        # gates: activated gates -> d(pre-activation) (bf16), in place; the sweep accumulates the bias gradients in fp32
        _hip.rnn_seq_bwd(_cellid(cell), prec, B, T, H, gates, kfw, kbw, GH, out, 2 * H, Tp * 2 * H, cst,
                         dout, 2 * H, Tp * 2 * H, 1.0, wf_off=I0 * GH, wb_off=I0 * GH,
                         db_fw=P4[1].grad if direct else None, db_bw=P4[3].grad if direct else None,
                         chunk_flag=None if dc is None else dc[0], chunk_rows=0 if dc is None else dc[1],
                         n_rows=0 if dc is None else dc[2], prepared_ws=_prepared_ws(kfw, kbw, cell, H, GH, I0 * GH, B, True),
                         progress=prog, progress_steps=TAIL_WINDOW)
        if dc is not None and not serial:
            dc[3]()                          # the other chunks: chain stream, enqueued behind the sweep's launch
            _hip.join_chain_stream()         # (they are finished when the sweep is; this orders later readers of dout)
        dx = dx_bw = None
        if ctx.needs_input_grad[0] and not two:        # on the dependency chain: main stream, first
            # dX [B*T, Ik] = dZ [B*T, 2GH] . [W_ih_fw | W_ih_bw]^T : B operand = shadow of the concatenated weights [Ik, 2GH]
            Wb = _shadow("ih", (kfw, kbw), I0, False, Ik, 2 * GH)                           # rows padded to Ik: [Ik, 2GH]
            dx = torch.empty(B, T, Ik, device=dev, dtype=bf)
            c = DOUT_CHUNK_ROWS
            if ctx.x_is_tanh and ctx.in_pyramid and c and T >= 4 * c and direct and _hip.rnn_seq_bwd_chunks_ok(_cellid(cell), prec, B, H):
                # first time chunk only; the dense node below (the consumer of this dPre) interleaves the others with its own.
                # (Only inside pBLSTMLayer's stack, where that dense node's input gradient goes to the recurrent layer below and
                #  nowhere else: a consumer that does not know about the chunks would read an unfinished tensor.)
                th = (T + 1) // 2

                def produce(k):
                    if k == 1 and _hip.streams_overlap(dev):
                        for t in (gates, x, dx):
                            t.record_stream(_hip.chain_stream())
                    lo0, lo1 = k * c, min((k + 1) * c, th)
                    hi0, hi1 = max(T - lo1, lo1), T - lo0
                    _hip.gemm_kk_frames(gates, Wb, dx, B, T, lo0, lo1 - lo0, hi0, hi1 - hi0, Ik, 2 * GH, 2 * GH, 2 * GH, Ik,
                                        tanh_y=x, ldy=Ik)

                produce(0)
                chain_done = []
                _DCHUNK[dx.data_ptr()] = (c, T, (th + c - 1) // c, produce, chain_done)
                _DPRE.add(dx.data_ptr())
                produced = torch.cuda.Event()
                produced.record()
            elif ctx.x_is_tanh:
                _hip.gemm_kk(gates, Wb, dx, B * T, Ik, 2 * GH, 2 * GH, 2 * GH, Ik, tanh_y=x, ldy=Ik)
                _DPRE.add(dx.data_ptr())
            else:
                _hip.gemm_kk(gates, Wb, dx, B * T, Ik, 2 * GH, 2 * GH, 2 * GH, Ik)
        elif two and (ctx.needs_input_grad[0] or ctx.needs_input_grad[9]):
            dx, dx_bw = torch.empty(B, T, Ik, device=dev, dtype=bf), torch.empty(B, T, Ik, device=dev, dtype=bf)
            for d, (dxd, k) in enumerate(((dx, kfw), (dx_bw, kbw))):
                Wd = _shadow("ih%d" % d, (k,), I0, False, Ik, GH)
                _hip.gemm_kk(gates, Wd, dxd, B * T, Ik, GH, 2 * GH, GH, Ik, a_off=d * GH)
        _hip.run_deferred()
        Ig = (I0 + 3) // 4 * 4          # rows of dW_ih the TN product writes; rows I0..Ig meet zero operand columns (exact zeros)

        one_pass = WGRAD_ONE_PASS and T > 1 and H % 128 == 0 and GH % 128 == 0 and B * T < (1 << 24) and out.dtype == bf and x.dtype == bf

        def wgrads(gk_of, d):
            # dW_ih = x^T . dG_d (contraction over all B*T frames; split-K inside las_gemm); dW_hh = sum_b sum_t h_prev^T . dG_d
            # (`part` is allocated HERE, i.e. on the stream that uses it: a block of the main stream's pool handed to the
            #  side stream would be recycled by the allocator while the side stream still writes it)
            gk = gk_of(d)
            if one_pass:
                # one pass over dG_d for both (round 4): the left operand is [x | h_prev] with h_prev read from `out` one frame back
                _hip.wgrad_ih_hh(xs[d], Ik, I0, out, 2 * H, Tp * 2 * H, gates, 2 * GH, B, T, H, GH, d, gk)
                return
            part = torch.empty(B, H, GH, device=dev) if T > 1 else None
            _hip.gemm(prec, xs[d], gates, gk, True, False, Ig, GH, B * T, Ik, 2 * GH, GH, beta=1.0, b_off=d * GH)
            if T > 1:
                a_off = d * H + (0 if d == 0 else 2 * H)
                b_off = d * GH + (2 * GH if d == 0 else 0)
                _hip.gemm(prec, out, gates, part, True, False, H, GH, T - 1, 2 * H, 2 * GH, GH, batch=B,
                          strideA=Tp * 2 * H, strideB=T * 2 * GH, strideC=H * GH, a_off=a_off, b_off=b_off)
                _hip.colsum(part, B, H * GH, H * GH, gk[I0:].reshape(-1), beta=1.0)

        def wgrads_both(gk_of, pair=False):
            if one_pass and pair:                      # both directions in one launch
                _hip.wgrad_ih_hh(x, Ik, I0, out, 2 * H, Tp * 2 * H, gates, 2 * GH, B, T, H, GH, 2, gk_of(0), gk_of(1), x_bw if two else None)
            else:
                for d in range(2):
                    wgrads(gk_of, d)

        if direct:
            # weight gradients: off the chain -> side stream, accumulated straight into the flat gradient bucket
            hold = ctx.hold_side and ctx.needs_input_grad[0] and _hip.streams_overlap(dev)

            def side_work(after=None):
                with _hip.on_side_stream(after=after):
                    side = _hip.side_stream()
                    for t in (x, gates, out) + ((x_bw,) if two else ()):
                        t.record_stream(side)
                    if after is not None:
                        # hand-over in chunks: these GEMMs would compete with the chain-stream chunks the next sweep is
                        # waiting for -- they start when the last chunk is done
                        if chain_done:
                            side.wait_event(chain_done[0])
                    elif hold:
                        VARIANTS["hold_side"] += 1
                        # keep the side stream (these GEMMs and whatever is queued behind them) off the machine until the NEXT
                        # BPTT sweep is resident: they would delay its start (it needs whole CUs) and slow the chain GEMMs in
                        # front of it; bounded wait, scheduling only
                        _hip.hold_until_next_sweep(dev)
                    wgrads_both(lambda d: P4[2 * d].grad)

            if produced is not None:
                _hip.defer_side(lambda: side_work(produced))      # run by the next sweep's node, after its launch
            elif (TAIL_TWO_STREAMS or windows) and not ctx.needs_input_grad[0] and not two:
                if BEFORE_TAIL_HOOK[0] is not None:
                    # data parallel: every gradient but this layer's is final once the work queued so far has run -- the
                    # all-reduce of that part of the bucket starts now, under the tail (las.las.LAS.train)
                    BEFORE_TAIL_HOOK[0](P4)
                # bottom layer = the end-of-step tail, nothing left to hide behind.  Round 4: both directions in ONE launch on THIS
                # stream (no event hand-over to another queue in front of it and behind it: 28 + 67 us of the 347 us tail).  Without the
                # one-pass kernel: the two directions' products on two streams (a single one of them does not fill the chip)
                if windows:
                    S = TAIL_WINDOW
                    nwin = (T + S - 1) // S
                    VARIANTS["tail_windows"] += nwin
                    VARIANTS["tail_follow"] += int(follow)

                    def window(c):
                        lo, n = c * S, min(S, T - c * S)
                        _hip.wgrad_ih_hh_window(x, Ik, I0, out, 2 * H, Tp * 2 * H, gates, 2 * GH, B, T, H, GH, 2, T - lo - n, lo, n,
                                                TAIL_WINDOW_WGS if c + 1 < nwin else 0, P4[0].grad, P4[2].grad, None)

                    if follow:
                        with _hip.on_side_stream(after=before_sweep):
                            for t in (x, gates, out, prog):
                                t.record_stream(_hip.side_stream())
                            for c in range(nwin):
                                _hip.wait_words_min(prog, nprog, min((c + 1) * S, T))
                                window(c)
                    else:
                        for c in range(nwin):
                            window(c)
                    return (dx, None, None, None, None, None, None, None, None, dx_bw)
                if one_pass and TAIL_ONE_LAUNCH:
                    wgrads_both(lambda d: P4[2 * d].grad, pair=True)
                    return (dx, None, None, None, None, None, None, None, None, dx_bw)
                with _hip.on_side_stream():
                    for t in (x, gates, out):
                        t.record_stream(_hip.side_stream())
                    wgrads(lambda d: P4[2 * d].grad, 0)
                with _hip.on_chain_stream():
                    for t in (x, gates, out):
                        t.record_stream(_hip.chain_stream())
                    wgrads(lambda d: P4[2 * d].grad, 1)
            else:
                side_work()
            return (dx, None, None, None, None, None, None, None, None, dx_bw)
        grads = []
        for d, k in enumerate((kfw, kbw)):
            dk = torch.zeros_like(k)
            wgrads(lambda d, dk=dk: dk, d)
            db = torch.empty(GH, device=dev)
            _hip.colsum(gates, B * T, GH, 2 * GH, db, x_off=d * GH)
            grads += [dk, db]
        return (dx, grads[0], grads[1], grads[2], grads[3], None, None, None, None, dx_bw)


def _blstm_params(scope, I, H, cell):
    st = V.default_store()
    G = 4 if cell == "lstm" else 1
    cs = cell_scope(cell)
    out = []
    for d in ("fw", "bw"):
        base = "%s/bidirectional_rnn/%s/%s/" % (scope, d, cs)
        out.append(st.get(base + "kernel", (I + H, G * H)))
        out.append(st.get(base + "bias", (G * H,), init="zeros"))
    return out


def blstm(inputs, cell_units, dropout_rate, is_training, scope="blstm"):
    """reference las/layers.py:28-54.  Returns ((out_fw, out_bw), (state_fw, state_bw)) like
    tf.nn.bidirectional_dynamic_rnn; no sequence_length: every padded frame is run."""
    outputs, states, _ = _blstm_full(inputs, cell_units, dropout_rate, is_training, scope)
    return outputs, states


DROPOUT_KERNEL = os.environ.get("LAS_NO_DROPOUT_KERNEL") != "1"     # input dropout of both directions in one launch (las_dropout_pair_fwd)


class _DropoutPair(torch.autograd.Function):
    """The two directions' independently masked copies of a recurrent layer's input (las/layers.py:37-47), as the x-projection's operand
    blocks: [.., K] -> two [.., ld] tensors (bf16 or fp32; columns K .. ld - 1 zero).  Masks are a counter-based function of
    (seed, direction, element) and are regenerated in backward (las_dropout_pair_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, keep, seed, out_bf16, ld):
        shp = x.shape
        K = shp[-1]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if x2.dtype not in (torch.float32, torch.bfloat16):
            x2 = x2.float()
        rows = x2.shape[0]
        odt = torch.bfloat16 if out_bf16 else torch.float32
        yf = torch.empty(rows, ld, device=x.device, dtype=odt)
        yb = torch.empty(rows, ld, device=x.device, dtype=odt)
        dt = lambda t: _hip.DT_BF16 if t.dtype == torch.bfloat16 else _hip.DT_F32
        _hip.check(_hip.lib().las_dropout_pair_fwd(_hip.p(x2), dt(x2), rows, K, K, _hip.p(yf), _hip.p(yb), dt(yf), ld, float(keep), int(seed),
                                                   _hip.stream()), "las_dropout_pair_fwd")
        ctx.cfg = (float(keep), int(seed), K, ld, x.dtype, tuple(shp))
        return yf.view(*shp[:-1], ld), yb.view(*shp[:-1], ld)

    @staticmethod
    def backward(ctx, gf, gb):
        keep, seed, K, ld, xdt, shp = ctx.cfg
        if gb.dtype != gf.dtype:
            gb = gb.to(gf.dtype)
        if gf.dtype not in (torch.float32, torch.bfloat16):
            gf, gb = gf.float(), gb.float()
        gf2, gb2 = gf.reshape(-1, gf.shape[-1]).contiguous(), gb.reshape(-1, gb.shape[-1]).contiguous()
        rows = gf2.shape[0]
        dx = torch.empty(rows, K, device=gf.device, dtype=xdt if xdt in (torch.float32, torch.bfloat16) else torch.float32)
        dt = lambda t: _hip.DT_BF16 if t.dtype == torch.bfloat16 else _hip.DT_F32
        _hip.check(_hip.lib().las_dropout_pair_bwd(_hip.p(gf2), _hip.p(gb2), dt(gf2), gf2.shape[-1], rows, K, _hip.p(dx), dt(dx), K, keep, seed,
                                                   _hip.stream()), "las_dropout_pair_bwd")
        return dx.view(shp), None, None, None, None


def _dropout_seed():
    """a fresh 62-bit seed from torch's CPU generator (reproducible under torch.manual_seed; no device round trip)"""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))


def _blstm_full(inputs, cell_units, dropout_rate, is_training, scope="blstm", pad_even=False):
    """blstm() plus the already-concatenated [B,T(+pad),2H] buffer the kernels wrote (the
    tf.concat(rnn_out, -1) of las/layers.py:69,81 is free: both directions share one tensor)."""
    _hip.require_gpu(inputs)
    x_bw = None
    I_true = inputs.shape[-1]
    if is_training is True and dropout_rate:
        # DropoutWrapper(input_keep_prob=1-rate) around fw_cell AND around bw_cell (las/layers.py:37-47): each direction
        # draws its own fresh Bernoulli mask on the cell INPUT at every time step, scaled by 1/keep.
        # The input projection is hoisted over all t, so the per-step masks of a direction are one mask over [B,T,I]
        # (torch RNG: plumbing, the masks are not on the MFMA path).
        if DROPOUT_KERNEL:
            # round 6: both directions' masked operand blocks in ONE launch, already bf16 / zero-padded to the product's K where the speed
            # mode's sweeps serve this layer (was: 2 x F.dropout + 2 x pad + 2 x cast); masks from a counter, regenerated in backward
            I_ = inputs.shape[-1]
            speed = _prec() == _hip.PREC_BF16 and _hip.rnn_seq_io_dtype(_cellid(_CFG["cell"]), _hip.PREC_BF16, int(cell_units)) == torch.bfloat16
            ld = _k64(I_) if speed else (I_ + 3) // 4 * 4
            x_fw, x_bw = _DropoutPair.apply(inputs, 1.0 - float(dropout_rate), _dropout_seed(), speed, ld)
            if not speed and ld != I_:
                x_fw, x_bw = x_fw[..., :I_], x_bw[..., :I_]
            inputs = x_fw
        else:
            x_bw = torch.nn.functional.dropout(inputs, p=float(dropout_rate), training=True)
            inputs = torch.nn.functional.dropout(inputs, p=float(dropout_rate), training=True)
    cell = _CFG["cell"]
    H = int(cell_units)
    I = I_true                                  # (the dropout launch may already have padded the operand blocks to the product's K)
    kfw, bfw, kbw, bbw = _blstm_params(scope, I, H, cell)
    after_blstm = _PARAMS.pop("after_blstm", False)
    _PARAMS["blstm"] = (kfw, bfw, kbw, bbw)
    _PARAMS["hold_side"] = after_blstm
    if _prec() == _hip.PREC_BF16 and _hip.rnn_seq_io_dtype(_cellid(cell), _hip.PREC_BF16, H) == torch.bfloat16:
        # speed mode: bf16 activation storage (the MFMA sweeps serve this H)
        out = _BLSTM16.apply(_as_bf16_operand(inputs), kfw, bfw, kbw, bbw, cell, H, pad_even, I,
                             None if x_bw is None else _as_bf16_operand(x_bw))
    else:
        if inputs.dtype != torch.float32:
            inputs = inputs.float()
            x_bw = None if x_bw is None else x_bw.float()
        out = _BLSTM.apply(inputs, kfw, bfw, kbw, bbw, cell, _prec(), H, pad_even, x_bw)
    _PARAMS.clear()
    T = inputs.shape[1]
    fw, bw = out[..., :H], out[..., H:]
    states = (fw[:, T - 1], bw[:, 0])
    return (fw, bw), states, out


def pBLSTMLayer(inputs, audiolen, num_layers, cell_units, dropout_rate, is_training, scope="Listener"):
    """Pyramidal BLSTM, reference las/layers.py:56-95 (the 7-argument call-site bug of las/las.py:15-21
    is fixed at the caller).  Returns (rnn_out [B,ceil(T/2^L),2H], states, audiolen float64)."""
    H = int(cell_units)
    st = V.default_store()
    sc = scope + "/blstm"
    if _prec() == _hip.PREC_BF16 and XPROJ_CHUNK_STEPS and ROW_T[0] is None:
        _hip.streams_overlap(inputs.device)     # (callers that do not go through LAS.train: probe before the first chunk is issued)
    _TANH_OUT.clear()
    _DPRE.clear()
    _EXPECT_DPRE.clear()
    _DCHUNK.clear()
    _DOUT_CHUNKS.clear()
    _XCHUNK.clear()

    def hint(B, Tn):
        # the dense + tanh whose output goes straight into the next recurrent layer (no dropout mask in between) follows that
        # layer's time chunks: only the first chunk of frames is on the chain in front of the sweep
        cs = _xproj_chunk_steps(B, Tn, H, _CFG["cell"]) if DENSE_CHUNKS and not (is_training is True and dropout_rate) else 0
        # (B <= 64: the sweep's clusters must leave most of the machine to the chunk products -- at B = 96, the stacked steps, the sweep holds
        #  120 of the 256 CUs and the dense chunks arrive late: 22.9 vs 21.4 ms per step with the x-projection's chunks alone)
        if cs and B <= DENSE_CHUNK_MAX_ROWS and _prec() == _hip.PREC_BF16:
            _PARAMS["xchunk"] = (cs, B, Tn)

    _, _, out = _blstm_full(inputs, H, dropout_rate, is_training, scope=sc)
    if num_layers > 0:
        hint(out.shape[0], out.shape[1])
    rnn_out = dense(out, st.get(sc + "/dense/kernel", (2 * H, 2 * H)),
                    st.get(sc + "/dense/bias", (2 * H,), init="zeros"), tanh=True, out_f32=num_layers == 0)       # :71-74
    audiolen = torch.as_tensor(audiolen).to(torch.float64)
    states = None
    for l in range(num_layers):
        sc = scope + "/pyramid_blstm_%d" % l
        _PARAMS["after_blstm"] = True     # in backward another BPTT sweep (of the layer below) follows this layer's
        _, states, out = _blstm_full(rnn_out, H, dropout_rate, is_training, scope=sc, pad_even=True)
        B, Tp, _ = out.shape
        # Eq (5): pad T to even, concat frame pairs -- a pure view of the zero-padded buffer (:83-88)
        pairs = out.view(B, Tp // 2, 4 * H)
        if l + 1 < num_layers:
            hint(B, Tp // 2)
        rnn_out = dense(pairs, st.get(sc + "/dense/kernel", (4 * H, 2 * H)),
                        st.get(sc + "/dense/bias", (2 * H,), init="zeros"), tanh=True, out_f32=l == num_layers - 1)   # :89-93
        # (the listener's LAST dense output feeds the Speller, whose interface is fp32; everything before it stays bf16)
        audiolen = (audiolen + audiolen % 2) / 2                                          # :94
        if ROW_T[0] is not None:
            ROW_T[0] = (ROW_T[0] + 1) // 2               # a row's frame pairs (an odd last frame pairs with the zero frame behind it)
    return rnn_out, states, audiolen


def _same_pad(n, k=3, s=2):
    """TF 'SAME': out = ceil(n/s); pad_total = max((out-1)*s + k - n, 0); before = total//2, rest after (App. A.6)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


BN_KERNEL = os.environ.get("LAS_NO_BN_KERNEL") != "1"        # training-mode batch norm (+ ReLU) through las_bn_relu_fwd / _bwd (round 6)


class _BNReLU(torch.autograd.Function):
    """[relu](batch_norm(x2d)) in training mode: las_bn_relu_fwd / las_bn_relu_bwd (statistics + apply; reduce + apply)."""

    @staticmethod
    def forward(ctx, x2d, gamma, beta, mov_mean, mov_var, relu):
        rows, C = x2d.shape
        dev = x2d.device
        stats = torch.empty(2, C, device=dev)
        y = torch.empty_like(x2d)
        lib = _hip.lib()
        ws = _hip.workspace(dev, lib.las_bn_workspace_bytes(rows, C), "bn")
        _hip.check(lib.las_bn_relu_fwd(_hip.p(x2d), rows, C, _hip.p(gamma), _hip.p(beta), 1e-3, _hip.p(stats[0]), _hip.p(stats[1]), _hip.p(mov_mean),
                                       _hip.p(mov_var), 0.01, int(relu), _hip.p(y), _hip.p(ws), ws.numel(), _hip.stream()), "las_bn_relu_fwd")
        ctx.save_for_backward(x2d, y, gamma, stats)
        ctx.relu = bool(relu)
        ctx.params = (_PARAMS.get("bn_gamma"), _PARAMS.get("bn_beta"))
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, y, gamma, stats = ctx.saved_tensors
        rows, C = x2d.shape
        dev = x2d.device
        dy = dy.contiguous()
        if dy.dtype != torch.float32:
            dy = dy.float()
        dx = torch.empty_like(x2d)
        gp, bp = ctx.params
        direct = _direct_ok(gp) and _direct_ok(bp)
        dg = gp.grad if direct else torch.zeros(C, device=dev)
        db = bp.grad if direct else torch.zeros(C, device=dev)
        lib = _hip.lib()
        ws = _hip.workspace(dev, lib.las_bn_workspace_bytes(rows, C), "bn")
        _hip.check(lib.las_bn_relu_bwd(_hip.p(x2d), _hip.p(y), _hip.p(dy), rows, C, _hip.p(gamma), _hip.p(stats[0]), _hip.p(stats[1]), int(ctx.relu),
                                       _hip.p(dx), _hip.p(dg), _hip.p(db), _hip.p(ws), ws.numel(), _hip.stream()), "las_bn_relu_bwd")
        if direct:
            return dx, None, None, None, None, None
        return dx, dg, db, None, None, None


def bn(inputs, is_training, scope="batch_normalization", relu=False):
    """tf.layers.batch_normalization over the last axis (reference las/layers.py:114-116; momentum 0.99, eps 1e-3,
    gamma=1, beta=0 -- SURVEY App. A.12).  Moving statistics live in the store's buffers.  relu: the ReLU the reference puts behind it
    (las/layers.py:108,161), fused into the same passes in training mode."""
    st = V.default_store()
    C = inputs.shape[-1]
    gamma = st.get(scope + "/gamma", (C,), init=lambda rng, shp: __import__("numpy").ones(shp))
    beta = st.get(scope + "/beta", (C,), init="zeros")
    mean = st.get_buffer(scope + "/moving_mean", (C,), 0.0)
    var = st.get_buffer(scope + "/moving_variance", (C,), 1.0)
    if mean.device != inputs.device:
        mean = st.buffers[scope + "/moving_mean"] = mean.to(inputs.device)
        var = st.buffers[scope + "/moving_variance"] = var.to(inputs.device)
    x2 = inputs.reshape(-1, C)
    if BN_KERNEL and bool(is_training) and x2.is_cuda and x2.dtype == torch.float32 and C % 4 == 0 and x2.shape[0] > 1:
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        _PARAMS["bn_gamma"], _PARAMS["bn_beta"] = gamma, beta
        y = _BNReLU.apply(x2, gamma, beta, mean, var, bool(relu))
        _PARAMS.pop("bn_gamma", None); _PARAMS.pop("bn_beta", None)
        return y.view(inputs.shape)
    y = torch.nn.functional.batch_norm(x2, mean, var, gamma, beta, training=bool(is_training), momentum=0.01, eps=1e-3)
    if relu:
        y = torch.relu(y)
    return y.view(inputs.shape)


def conv2d(inputs, output_dim, k_h=3, k_w=3, d_h=2, d_w=2, stddev=1, name="conv2d", apply_bn=False, is_training=True):
    """3x3 stride-2 SAME convolution + bias (+ BN) + ReLU on an NHWC block (reference las/layers.py:97-112).
    Secondary encoder path: the convolution itself runs through torch/MIOpen, with TF's
    asymmetric SAME padding applied explicitly."""
    import numpy as np
    st = V.default_store()
    Cin = inputs.shape[-1]
    w = st.get(name + "/w", (k_h, k_w, Cin, output_dim), init=lambda rng, shp: rng.randn(*shp) * stddev * 0.01)
    b = st.get(name + "/b", (output_dim,), init=lambda rng, shp: np.full(shp, 0.01))
    x = inputs.permute(0, 3, 1, 2)                                   # NHWC -> NCHW
    pt, pb = _same_pad(x.shape[2], k_h, d_h)
    pl, pr = _same_pad(x.shape[3], k_w, d_w)
    x = torch.nn.functional.pad(x, (pl, pr, pt, pb))
    y = torch.nn.functional.conv2d(x, w.permute(3, 2, 0, 1), b, stride=(d_h, d_w))
    y = y.permute(0, 2, 3, 1)                                        # back to NHWC
    if apply_bn:
        return bn(y, is_training, scope=name + "/batch_normalization", relu=True)
    return torch.relu(y)


def CNNLayer(inputs, audiolen, num_enc_layers, feat_dim, cell_units, num_channel, dropout_rate, apply_bn=False,
             is_training=False, scope="Listener"):
    """CNN listener, reference las/layers.py:118-163: 2 x [conv 3x3 s2 (+BN) + ReLU] -> reshape ->
    num_enc_layers x [BLSTM -> concat -> dense(enc_units) (-> BN if apply_bn) -> relu(bn(.))]  (the trailing
    BN+ReLU is unconditional in the reference: SURVEY quirk Q4).  Returns (enc_out, enc_state, audiolen)."""
    enc_units = int(cell_units)
    st = V.default_store()
    audiolen = torch.as_tensor(audiolen).to(torch.float64)
    conv_out = inputs
    fd = float(feat_dim)
    for i in range(2):
        fd = (fd + fd % 2) / 2                                       # :127-129 (float arithmetic as in the reference)
        audiolen = (audiolen + audiolen % 2) / 2
        conv_out = conv2d(conv_out, num_channel, name=scope + "/conv2d_%d" % i, apply_bn=apply_bn, is_training=is_training)
    B = conv_out.shape[0]
    enc_out = conv_out.reshape(B, -1, int(fd * num_channel))         # :139-140
    enc_state = None
    for i in range(num_enc_layers):
        sc = scope + "/blstm_%d" % i
        _, enc_state, out = _blstm_full(enc_out.contiguous(), cell_units, dropout_rate, is_training, scope=sc)
        enc_out = dense(out, st.get(sc + "/dense/kernel", (2 * cell_units, enc_units)),
                        st.get(sc + "/dense/bias", (enc_units,), init="zeros"))
        if apply_bn:
            enc_out = bn(enc_out, is_training, scope=sc + "/batch_normalization")
            enc_out = bn(enc_out, is_training, scope=sc + "/batch_normalization_1", relu=True)
        else:
            enc_out = bn(enc_out, is_training, scope=sc + "/batch_normalization", relu=True)
    return enc_out, enc_state, audiolen


# ------------------------------------------------------------------------------------------------
# attention (forward-only single-step objects; training goes through las.las.Speller's fused loop)
# ------------------------------------------------------------------------------------------------
class BaseAttention:
    """reference las/layers.py:165-213."""

    def __init__(self, att_size, smoothing):
        self.att_size = att_size
        self.smoothing = smoothing

    def mask(self, original_len, padded_len):
        """mask[b,t] = float(t+1 <= int32(len[b]))   (las/layers.py:172-197)"""
        ln = torch.as_tensor(original_len)
        y = torch.arange(1, int(padded_len) + 1, dtype=torch.int32, device=ln.device)[None, :]
        return (y <= ln.to(torch.int32)[:, None]).to(torch.float32)

    def attend(self, inputs, energy, seqlen):
        """las/layers.py:199-213 (torch elementwise ops: a convenience entry, not the hot path --
        the hot path fuses this into the K5 row kernel)."""
        m = self.mask(seqlen, inputs.shape[1]).to(energy.device)
        energy = torch.where(m == 0, torch.full_like(energy, -1e8), energy)
        alphas = torch.softmax(energy, -1)
        return (inputs * alphas[..., None]).sum(1), alphas


class _AttentionStep(BaseAttention):
    mode = "add"

    def __init__(self, h_dim, s_dim, att_size, kernel_size=10, num_channels=201, smoothing=False,
                 scope="Speller/decode/attention"):
        super().__init__(att_size, smoothing)
        self.h_dim, self.s_dim = h_dim, s_dim
        self.kernel_size, self.num_channels = kernel_size, num_channels
        self.scope = scope

    def params(self):
        """Variables with the reference's TF names (las/layers.py:248-251,295-306;
        `u` lives at Speller/while/decode/attention/Variable, las/beam_search.py:257,264)."""
        st = V.default_store()
        sc = self.scope
        p = {"Wh": st.get(sc + "/dense/kernel", (self.h_dim, self.att_size)),
             "Ws": st.get(sc + "/dense_1/kernel", (self.s_dim, self.att_size)),
             "u": st.get("Speller/while/decode/attention/Variable", (self.att_size,), init="uniform1")}
        if self.mode == "loc":
            Kc, C = self.kernel_size, self.num_channels
            p["loc_w"] = st.get(sc + "/conv1d/kernel", (Kc, 1, C), fan=(Kc, Kc * C))
            p["loc_b"] = st.get(sc + "/conv1d/bias", (C,), init="zeros")
            p["Wf"] = st.get(sc + "/dense_2/kernel", (C, self.att_size))
        return p

    def __call__(self, hidden, state, align, seqlen):
        """(context [B,h_dim], alphas [B,T]) for one decode step -- forward only."""
        from las.las import attention_forward_step
        return attention_forward_step(self, hidden, state, align, seqlen)


class AdditiveAttention(_AttentionStep):
    """Bahdanau attention, reference las/layers.py:215-257."""
    mode = "add"

    def __init__(self, h_dim, s_dim, att_size, smoothing=False):
        super().__init__(h_dim, s_dim, att_size, smoothing=smoothing)


class LocationAwareAttention(_AttentionStep):
    """Location-aware attention, reference las/layers.py:259-311."""
    mode = "loc"

    def __init__(self, h_dim, s_dim, att_size, kernel_size=10, num_channels=201, smoothing=False):
        super().__init__(h_dim, s_dim, att_size, kernel_size, num_channels, smoothing)
