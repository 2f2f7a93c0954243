"""las.las -- Listener / Attention / Speller / LAS with the reference's construction API
(reference las/las.py) on the MI355X engine.

Semantic shift: the reference builds a TF graph once (`las.train(xs, ys)` at
train.py:75) and `sess.run`s its handles; here `LAS.train(xs, ys)` EXECUTES one optimisation step and
returns the same 7-tuple with concrete values, `LAS.inference(xs)` executes greedy decoding.  All
arithmetic runs in liblas_hip.so (`las._hip`); there is no CPU path.
"""
import ctypes
import os
import math

import numpy as np
import torch

from las import _hip
from las import layers as L
from las import variables as V
from las.layers import AdditiveAttention, LocationAwareAttention, pBLSTMLayer, CNNLayer  # noqa: F401
from las.parallel import sampling_seed
from las.utils import convert_idx_to_token_tensor

SOS_ID = 1  # tf.ones(...) look-up at reference las/las.py:81
RECOVER_STEPS = os.environ.get("LAS_NO_STEP_RECOVERY") != "1"     # LAS.train re-runs steps lost to a residency time-out
DP_LAG = 3                                                          # data parallel: a step's all-reduced guard slot is looked at this many steps later


class Listener:
    """reference las/las.py:6-36."""

    def __init__(self, args):
        self.args = args

    def output_length(self, audiolen, encoder='cnn'):
        """Encoder frame counts after the time halvings (pblstm: one per pyramid layer, las/layers.py:94; cnn: the
        two stride-2 convolutions, las/layers.py:127-129), float64 on the host as the reference computes them."""
        n = torch.as_tensor(audiolen).to(torch.float64)
        for _ in range(self.args.num_enc_layers if encoder == 'pblstm' else 2):
            n = (n + n % 2) / 2
        return n

    def __call__(self, inputs, audiolen, encoder='cnn', is_training=True):
        if encoder == 'pblstm':
            x = inputs.reshape(inputs.shape[0], -1, self.args.feat_dim * 3)            # las/las.py:14
            # NB the reference passes a 7th positional `apply_bn` here (las/las.py:15-21) which
            # pBLSTMLayer's signature (las/layers.py:56) does not accept; dropped.
            enc_out, enc_state, enc_len = pBLSTMLayer(x, audiolen, self.args.num_enc_layers, self.args.enc_units,
                                                      self.args.dropout_rate, is_training, scope="Listener")
        elif encoder == 'cnn':
            enc_out, enc_state, enc_len = CNNLayer(inputs, audiolen, self.args.num_enc_layers, self.args.feat_dim,
                                                   self.args.enc_units, self.args.num_enc_channels,
                                                   self.args.dropout_rate, self.args.apply_bn, is_training)
        else:
            raise NotImplementedError
        return enc_out, enc_state, enc_len


class Attention:
    """reference las/las.py:39-54."""

    def __init__(self, h_dim, s_dim, att_size, kernel_size, num_channels, mode='add'):
        self.mode = mode
        if self.mode == 'add':
            self.att_layer = AdditiveAttention(h_dim, s_dim, att_size)
        elif self.mode == 'loc':
            self.att_layer = LocationAwareAttention(h_dim, s_dim, att_size, kernel_size, num_channels)
        else:
            raise NotImplementedError

    def __call__(self, hidden, state, align, seqlen):
        return self.att_layer(hidden, state, align, seqlen)


# ------------------------------------------------------------------------------------------------
# the fused decode loop
# ------------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _fill_fwd_args(fa, dims, P, enc, keys, enc_len_i32, tokens_in, tokens_out, bufs, step_logits, seed,
                   keep_state0=False, align0=None, emb_mask=None, emb_noise=None):
    for k, v in dims.items():
        setattr(fa, k, v)
    fa.step_logits = int(step_logits)
    fa.flags = int(_hip.speller_flags)
    fa.keep_state0 = int(keep_state0)
    fa.forget_bias = 1.0
    fa.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    fa.enc, fa.keys, fa.enc_len = enc.data_ptr(), keys.data_ptr(), enc_len_i32.data_ptr()
    fa.Ws, fa.u, fa.emb = P["Ws"].data_ptr(), P["u"].data_ptr(), P["emb"].data_ptr()
    fa.Wv, fa.bv = P["Wv"].data_ptr(), P["bv"].data_ptr()
    if dims["mode"] == _hip.ATT_LOC:
        fa.loc_w, fa.loc_b, fa.Wf = P["loc_w"].data_ptr(), P["loc_b"].data_ptr(), P["Wf"].data_ptr()
    else:
        fa.loc_w = fa.loc_b = fa.Wf = None
    keep = (_ptr_array(P["cellW"]), _ptr_array(P["cellb"]))   # caller must hold these until the C call returns
    fa.cellW, fa.cellb = keep
    fa.tokens_in = tokens_in.data_ptr()
    fa.tokens_out = tokens_out.data_ptr() if tokens_out is not None else None
    fa.logits, fa.alphas = bufs["logits"].data_ptr(), bufs["alphas"].data_ptr()
    fa.align0 = align0.data_ptr() if align0 is not None else None
    fa.emb_mask = emb_mask.data_ptr() if emb_mask is not None else None
    fa.emb_noise = emb_noise.data_ptr() if emb_noise is not None else None
    fa.hs = bufs["hs"].data_ptr()
    fa.cs = bufs["cs"].data_ptr() if bufs["cs"] is not None else None
    fa.gates, fa.xin0 = bufs["gates"].data_ptr(), bufs["xin0"].data_ptr()
    fa.act_save = bufs["act"].data_ptr() if bufs.get("act") is not None else None
    fa.ws, fa.ws_bytes = None, 0
    fa.status = _hip.status_word(enc.device).data_ptr()      # a loop-kernel poll time-out is reported here (never a trap / hang)
    return keep


def _alloc_bufs(dims, dev):
    B, Tp, Hd, D, NL, E, V_, U = (dims[k] for k in ("B", "Tp", "Hd", "D", "NL", "E", "V", "U"))
    G = 4 if dims["cell"] == _hip.CELL_LSTM else 1
    return {
        "logits": torch.empty(U, B, V_, device=dev), "alphas": torch.empty(U, B, Tp, device=dev),
        "hs": torch.empty(NL, U + 1, B, D, device=dev),
        "cs": torch.empty(NL, U + 1, B, D, device=dev) if G == 4 else None,
        "gates": torch.empty(NL, U, B, G * D, device=dev), "xin0": torch.empty(U, B, E + Hd + D, device=dev),
    }


SAVE_ACTIVATIONS = os.environ.get("LAS_SPELLER_SAVE_ACT", "1") != "0"


class _SpellerLoop(torch.autograd.Function):
    """K4 (hoisted key projection) + las_speller_fwd / las_speller_bwd."""

    @staticmethod
    def forward(ctx, enc, Wh, Ws, u, emb, Wv, bv, loc_w, loc_b, Wf, cfg, enc_len_i32, tokens_in, *cell_params):
        dims, prec, step_logits, seed, emb_mask, emb_noise = cfg
        dev = enc.device
        enc = enc.contiguous()
        B, Tp, Hd = enc.shape
        A = Wh.shape[1]
        NL = dims["NL"]
        keys = torch.empty(B, Tp, A, device=dev)
        _hip.gemm(prec, enc, Wh, keys, False, False, B * Tp, A, Hd, Hd, A, A)          # dense(hidden), hoisted
        P = {"Ws": Ws, "u": u, "emb": emb, "Wv": Wv, "bv": bv, "loc_w": loc_w, "loc_b": loc_b, "Wf": Wf,
             "cellW": list(cell_params[:NL]), "cellb": list(cell_params[NL:])}
        bufs = _alloc_bufs(dims, dev)
        if SAVE_ACTIVATIONS and prec == _hip.PREC_BF16 and dims["mode"] == _hip.ATT_LOC and any(ctx.needs_input_grad):
            # location-aware attention: the rows' tanh(keys + q + f . Wf), kept for the gradient loop (fp16, 2 A T' bytes per row and step)
            # -- recomputing it there costs 20 FMAs + 2 tanh per lane and frame (23 -> 20.4 us per gradient step at K = 201, C = 10).  The
            # additive loop recomputes: its tanh hides under the Ws loads, the saved values' 41 KB per step do not (10.4 -> 10.6 us)
            bufs["act"] = torch.empty(_hip.lib().las_speller_act_save_bytes(dims["U"], B, Tp, A, dims["C"]), dtype=torch.uint8, device=dev)
        tokens_out = torch.zeros(dims["U"], B, dtype=torch.int32, device=dev) if step_logits else None
        fa = _hip.SpellerFwdArgs()
        keep = _fill_fwd_args(fa, dims, P, enc, keys, enc_len_i32, tokens_in, tokens_out, bufs, step_logits, seed,
                              emb_mask=emb_mask, emb_noise=emb_noise)
        nbytes = _hip.lib().las_speller_workspace_bytes(B, Tp, Hd, A, dims["D"], NL, dims["E"], dims["V"], dims["U"], dims["cell"])
        ws = _hip.workspace(dev, nbytes, "speller")
        ctx.ws_epoch = _hip.workspace_epoch(dev, "speller")
        fa.ws, fa.ws_bytes = ws.data_ptr(), ws.numel()
        with _hip._timed("speller_fwd[U=%d]" % dims["U"]):
            _hip.check(_hip.lib().las_speller_fwd(ctypes.byref(fa), _hip.stream()), "las_speller_fwd")
        del keep
        ctx.saved = (enc, keys, Wh, P, bufs, enc_len_i32, tokens_in, tokens_out, dims, prec, step_logits, seed, emb_mask, emb_noise)
        ctx.mark_non_differentiable(bufs["alphas"])
        ctx.tokens_out = tokens_out
        return bufs["logits"], bufs["alphas"]

    @staticmethod
    def backward(ctx, dlogits, _dalphas):
        enc, keys, Wh, P, bufs, enc_len_i32, tokens_in, tokens_out, dims, prec, step_logits, seed, emb_mask, emb_noise = ctx.saved
        dev = enc.device
        B, Tp, Hd = enc.shape
        A = Wh.shape[1]
        NL, U, V_ = dims["NL"], dims["U"], dims["V"]
        dlogits = dlogits.contiguous()
        loc = dims["mode"] == _hip.ATT_LOC
        names = ["Ws", "u", "emb", "Wv", "bv"] + (["loc_w", "loc_b", "Wf"] if loc else [])
        plist = [P[k] for k in names] + list(P["cellW"]) + list(P["cellb"]) + [Wh]
        # parameter gradients accumulate (+=) straight into the flat gradient bucket when the store is flattened;
        # they are off the dependency chain, so they run on the side stream while the Listener's BPTT proceeds
        direct = all(L._direct_ok(p) for p in plist)
        sizes = [B * Tp * Hd, B * Tp * A] + ([] if direct else [p.numel() for p in plist])
        zbuf = torch.zeros(sum(sizes), device=dev)                       # ONE fill for every caller-zeroed buffer
        views, o = [], 0
        for n in sizes:
            views.append(zbuf[o:o + n])
            o += n
        d_enc, d_keys = views[0].view(B, Tp, Hd), views[1].view(B, Tp, A)
        gl = [p.grad for p in plist] if direct else [v.view(p.shape) for v, p in zip(views[2:], plist)]
        g = dict(zip(names, gl))
        dcW = gl[len(names):len(names) + NL]
        dcb = gl[len(names) + NL:len(names) + 2 * NL]
        dWh = gl[-1]
        nbytes = _hip.lib().las_speller_workspace_bytes(B, Tp, Hd, A, dims["D"], NL, dims["E"], V_, U, dims["cell"])
        # nobody asked for the Speller's workspace since this node's forward: its bf16 copies of enc / keys / Ws are still there
        reuse = _hip.workspace_epoch(dev, "speller") == ctx.ws_epoch
        ws = _hip.workspace(dev, nbytes, "speller")
        ba = _hip.SpellerBwdArgs()
        keepf = _fill_fwd_args(ba.f, dims, P, enc, keys, enc_len_i32, tokens_in, tokens_out, bufs, step_logits, seed,
                               emb_mask=emb_mask, emb_noise=emb_noise)
        if reuse:
            ba.f.flags |= _hip.SPELLER_REUSE_PREP
        ba.f.ws, ba.f.ws_bytes = ws.data_ptr(), ws.numel()
        ba.dlogits = dlogits.data_ptr()
        ba.d_enc, ba.d_keys = d_enc.data_ptr(), d_keys.data_ptr()
        ba.dWs, ba.du, ba.demb, ba.dWv, ba.dbv = (g[k].data_ptr() for k in ("Ws", "u", "emb", "Wv", "bv"))
        if loc:
            ba.dloc_w, ba.dloc_b, ba.dWf = g["loc_w"].data_ptr(), g["loc_b"].data_ptr(), g["Wf"].data_ptr()
        keep = (_ptr_array(dcW), _ptr_array(dcb))
        ba.dcellW, ba.dcellb = keep
        lib = _hip.lib()
        with _hip._timed("speller_bwd[U=%d]" % dims["U"]):
            _hip.check(lib.las_speller_bwd_part(ctypes.byref(ba), 1, _hip.stream()), "las_speller_bwd_part(1)")
        # key projection backward (K4), input side: d_enc += d_keys . Wh^T
        _hip.gemm(prec, d_keys, Wh, d_enc, False, True, B * Tp, Hd, A, A, A, Hd, beta=1.0)
        if direct:
            main_done = torch.cuda.Event()
            main_done.record()
            held = [enc, keys, dlogits, d_keys, ws, tokens_in, zbuf] + [v for v in bufs.values() if v is not None] + \
                   ([emb_mask] if emb_mask is not None else []) + ([emb_noise] if emb_noise is not None else [])

            def side_part(ba=ba, keep=(keep, keepf), held=held):
                # runs after the next backward node has enqueued its chain kernels (_hip.run_deferred)
                side = _hip.side_stream()
                with _hip.on_side_stream(after=main_done):
                    for t in held:
                        t.record_stream(side)
                    if L.HOLD_SIDE and _hip.streams_overlap(enc.device):
                        L.VARIANTS["hold_side"] += 1
                        # these products would otherwise start beside the chain GEMMs in front of the listener's first BPTT sweep
                        # (r3 timeline: the dense layer's dX product took 101 us next to them): wait until that sweep is resident
                        _hip.hold_until_last_sweep(enc.device)
                    _hip.check(lib.las_speller_bwd_part(ctypes.byref(ba), 2, _hip.stream()), "las_speller_bwd_part(2)")
                    _hip.gemm(prec, enc, d_keys, dWh, True, False, Hd, A, B * Tp, Hd, A, A, beta=1.0)  # dWh += enc^T . d_keys

            _hip.defer_side(side_part)
            return (d_enc, None, None, None, None, None, None, None, None, None, None, None, None, *([None] * (2 * NL)))
        _hip.check(lib.las_speller_bwd_part(ctypes.byref(ba), 2, _hip.stream()), "las_speller_bwd_part(2)")
        _hip.gemm(prec, enc, d_keys, dWh, True, False, Hd, A, B * Tp, Hd, A, A, beta=1.0)
        del keep, keepf
        return (d_enc, dWh, g["Ws"], g["u"], g["emb"], g["Wv"], g["bv"], g.get("loc_w"), g.get("loc_b"), g.get("Wf"),
                None, None, None, *dcW, *dcb)


class Speller:
    """reference las/las.py:57-207."""

    def __init__(self, args):
        self.args = args
        # reference: hidden_dim = enc_units (las/las.py:61) although pBLSTMLayer emits 2*enc_units
        # (las/layers.py:66) -- SURVEY quirk Q2.  The pblstm listener needs 2*enc_units.
        self.hidden_dim = self.args.enc_units * (2 if str(getattr(args, "enc_type", "cnn")).lower() == "pblstm" else 1)
        self.state_dim = self.args.dec_units * self.args.num_dec_layers
        self.cell = L.get_cell()
        self._build_decoder_cell()
        self._build_embeddings()
        self.att_layer = Attention(h_dim=self.hidden_dim, s_dim=self.state_dim, att_size=self.args.attention_size,
                                   kernel_size=self.args.loc_kernel_size, num_channels=self.args.loc_num_channels,
                                   mode=self.args.mode)
        self.last_tokens_out = None
        self.logits_every_step = False     # (tests: step_logits = 1 in training too -- the form before round 6's step_logits = 2)
        self.rank = 0              # data-parallel rank: per-rank sampling noise (set by LAS.train from las.dp)

    # -- variables ---------------------------------------------------------------------------------
    def _build_decoder_cell(self):
        """las/las.py:191-199: BasicRNNCell(dec_units) or MultiRNNCell of them (or the LSTM variant)."""
        a = self.args
        G = 4 if self.cell == "lstm" else 1
        cs = L.cell_scope(self.cell)
        self.dec_cell = []
        self._cell_shapes = []
        for l in range(a.num_dec_layers):
            I = (a.embedding_size + self.hidden_dim) if l == 0 else a.dec_units
            base = ("Speller/decode/%s/" % cs) if a.num_dec_layers == 1 else \
                ("Speller/decode/multi_rnn_cell/cell_%d/%s/" % (l, cs))
            self._cell_shapes.append((base, (I + a.dec_units, G * a.dec_units)))

    def _build_embeddings(self):
        """las/las.py:201-207 (scope 'embedding', U(-1,1))."""
        self._emb_shape = (self.args.vocab_size, self.args.embedding_size)

    @property
    def embedding_matrix(self):
        return V.default_store().get("embedding/embedding_matrix", self._emb_shape, init="uniform1")

    def _params(self):
        st = V.default_store()
        a = self.args
        P = dict(self.att_layer.att_layer.params())
        P["emb"] = self.embedding_matrix
        P["cellW"] = [st.get(b + "kernel", s) for b, s in self._cell_shapes]
        P["cellb"] = [st.get(b + "bias", (s[1],), init="zeros") for b, s in self._cell_shapes]
        P["Wv"] = st.get("Speller/decode/dense/kernel", (a.dec_units, a.vocab_size))
        P["bv"] = st.get("Speller/decode/dense/bias", (a.vocab_size,), init="zeros")
        return P

    def _dims(self, B, Tp, U):
        a = self.args
        return {"B": B, "Tp": Tp, "Hd": self.hidden_dim, "A": a.attention_size, "D": a.dec_units,
                "NL": a.num_dec_layers, "E": a.embedding_size, "V": a.vocab_size, "U": int(U),
                "cell": L._cellid(self.cell), "mode": _hip.ATT_LOC if a.mode == "loc" else _hip.ATT_ADD,
                "prec": L._prec(), "Kc": a.loc_kernel_size if a.mode == "loc" else 0,
                "C": a.loc_num_channels if a.mode == "loc" else 0}

    # -- schedules ---------------------------------------------------------------------------------
    def _scheduled_sampling(self, step=None):
        """las/las.py:177-183 (float32 arithmetic as in the TF graph)."""
        a = self.args
        step = np.float32(V.default_store().global_step if step is None else step)
        progress = min((step - np.float32(a.warmup_step)) / np.float32(float(a.max_step - a.warmup_step)), np.float32(1.0))
        return float(min(np.float32(1.0), np.float32(1.0) - np.float32(progress) * np.float32(1.0 - a.min_rate)))

    # -- the loop ----------------------------------------------------------------------------------
    def prepare(self, B, enc_len, dec_steps, dev, teacher=None, is_training=True, coins=None, sampled=None):
        """Host-side half of the decode loop: token schedule (teacher forcing / scheduled-sampling coins,
        las/las.py:87-101), encoder lengths on the device, dropout mask.  LAS.train calls this BEFORE it enqueues the
        Listener so that the small host->device copies and the Python work overlap with the encoder kernels."""
        a = self.args
        U = int(dec_steps)
        if torch.is_tensor(enc_len) and enc_len.is_cuda and enc_len.dtype == torch.int32:
            enc_len_i32 = enc_len.contiguous()
        else:
            # float -> int32 as tf.sequence_mask's cast does (las/layers.py:193); uploaded from pinned memory with a
            # non-blocking copy, so the host is not held until the stream drains (the caching host allocator keeps the
            # staging block alive until the copy has executed)
            host = torch.as_tensor(enc_len).to(torch.float64).to(torch.int32).reshape(-1)
            enc_len_i32 = host.pin_memory().to(dev, non_blocking=True)
        st = V.default_store()
        tokens_in = torch.full((U, B), -1, dtype=torch.int32, device=dev)
        tokens_in[0] = SOS_ID
        step_logits = True
        if is_training:
            tf_rate = self._scheduled_sampling() if a.scheduled_sampling else 1.0          # las/las.py:87-90
            y = torch.as_tensor(teacher).to(dev).to(torch.int32)
            if coins is None:
                # one coin per step, shared by every data-parallel rank (seeded by the global step)
                rng = np.random.RandomState((1234567 + 7919 * st.global_step) % (2 ** 31))
                coins = tf_rate > rng.uniform(0.0, 1.0, size=U).astype(np.float32)
            coins = np.asarray(coins, bool)
            if U > 1:
                tokens_in[1:] = y[:, :U - 1].t()
                if not coins[:U - 1].all():
                    # (pinned + non-blocking: a pageable upload would wait for the stream, i.e. for the whole previous step)
                    idx = torch.as_tensor(np.nonzero(~coins[:U - 1])[0] + 1).pin_memory().to(dev, non_blocking=True)
                    if sampled is not None:
                        sm = torch.as_tensor(sampled).to(dev).to(torch.int32)
                        tokens_in[idx] = sm[:, :U - 1].t()[idx - 1]
                    else:
                        tokens_in[idx] = -2
            # in-loop logits / draws are needed exactly when some step samples ON THE DEVICE (token -2): known on the host,
            # no read-back (a `.item()` here would make every scheduled-sampling step wait for the whole previous step)
            # (2: only at the steps that sample -- a teacher-forced step keeps the projection, two arg-max and the draws off its chain; the loss's
            #  logits come from the batched product behind the loop, as without sampling: las_hip.h)
            step_logits = (1 if self.logits_every_step else 2) if (U > 1 and not coins[:U - 1].all() and sampled is None) else False
        emb_mask = None
        if is_training and a.dropout_rate:
            # tf.layers.dropout on the embedded input token of every step (las/las.py:107-108); step 0's SOS
            # embedding is looked up before the loop (las/las.py:81) and is not dropped
            keep = 1.0 - float(a.dropout_rate)
            emb_mask = (torch.rand(U, B, a.embedding_size, device=dev) < keep).to(torch.float32) / keep
            emb_mask[0] = 1.0
        emb_noise = None
        if a.add_vn:
            # variational noise (las/las.py:164-166): `_look_up` adds a fresh N(0, 0.075) draw to the whole embedding matrix
            # on EVERY call -- training and inference alike (the reference does not test is_training there):
            # one [V,E] noise matrix per decode step (step 0 = the SOS look-up of las/las.py:81)
            emb_noise = self.vn_noise if getattr(self, "vn_noise", None) is not None else \
                torch.randn(U, a.vocab_size, a.embedding_size, device=dev) * 0.075
        return {"B": B, "U": U, "enc_len_i32": enc_len_i32, "tokens_in": tokens_in, "step_logits": step_logits,
                "emb_mask": emb_mask, "emb_noise": emb_noise}

    def __call__(self, enc_out, enc_len, dec_steps, teacher=None, is_training=True, coins=None, sampled=None, prepared=None):
        """reference las/las.py:72-143.  Returns (logits [B,U,V], ctc_logits, alphas [B,U,T']).

        coins / sampled are test hooks: coins[t] True = teacher forcing at step t (one scalar coin
        per step for the whole batch, las/las.py:101); sampled [B,U] supplies the Categorical draws."""
        a = self.args
        _hip.require_gpu(enc_out)
        if a.ctc:
            raise NotImplementedError("CTC head (las/las.py:75-77,335-349): README 'not yet fully tested', out of scope (SURVEY T5)")
        dev = enc_out.device
        B, Tp, _ = enc_out.shape
        U = int(dec_steps)
        st = V.default_store()
        if prepared is None:
            prepared = self.prepare(B, enc_len, U, dev, teacher, is_training, coins, sampled)
        assert prepared["B"] == B and prepared["U"] == U
        enc_len_i32, tokens_in = prepared["enc_len_i32"], prepared["tokens_in"]
        step_logits, emb_mask = prepared["step_logits"], prepared["emb_mask"]
        P = self._params()
        cfg = (self._dims(B, Tp, U), L._prec(), step_logits, sampling_seed(st.global_step, self.rank), emb_mask,
               prepared.get("emb_noise"))
        cp = list(P["cellW"]) + list(P["cellb"])
        logits_tm, alphas_tm = _SpellerLoop.apply(enc_out, P["Wh"], P["Ws"], P["u"], P["emb"], P["Wv"], P["bv"],
                                                  P.get("loc_w"), P.get("loc_b"), P.get("Wf"), cfg, enc_len_i32,
                                                  tokens_in, *cp)
        self.last_tokens_in = tokens_in
        return logits_tm.permute(1, 0, 2), None, alphas_tm.permute(1, 0, 2)

    def decode(self, enc_out, enc_len, dec_state, prev_token, prev_align, is_training, keys=None, token_ids=None):
        """One decode step, reference las/las.py:145-160 -- forward only (beam search).

        dec_state: tuple over layers of h [N,D] (rnn) or (c,h) (lstm).  prev_token: embedded token
        [N,E] (as `_look_up` returns) -- or pass token_ids to skip the host-side embedding match.
        Returns (cur_token logits [N,V], new dec_state, alphas [N,T'])."""
        return _decode_step(self, enc_out, enc_len, dec_state, prev_token, prev_align, keys, token_ids)

    def _look_up(self, token):
        """las/las.py:162-168 (host-side convenience; the decode loop looks tokens up inside the row kernels)."""
        m = self.embedding_matrix.detach()
        if self.args.add_vn:
            m = m + torch.randn_like(m) * 0.075
        return m[torch.as_tensor(token).long()]

    def _get_hidden_state(self, dec_state):
        """las/las.py:185-189: concat of the layers' states (h only for the lstm variant)."""
        hs = [s if not isinstance(s, (tuple, list)) else s[1] for s in dec_state]
        return torch.cat(hs, -1)

    def zero_state(self, n, device):
        D, NL = self.args.dec_units, self.args.num_dec_layers
        z = lambda: torch.zeros(n, D, device=device)
        return tuple((z(), z()) if self.cell == "lstm" else z() for _ in range(NL))


def _decode_step(sp, enc_out, enc_len, dec_state, prev_token, prev_align, keys=None, token_ids=None):
    a = sp.args
    dev = enc_out.device
    N, Tp, Hd = enc_out.shape
    P = sp._params()
    prec = L._prec()
    enc = enc_out.contiguous()
    if keys is None:
        keys = torch.empty(N, Tp, a.attention_size, device=dev)
        _hip.gemm(prec, enc, P["Wh"].detach(), keys, False, False, N * Tp, a.attention_size, Hd, Hd, a.attention_size,
                  a.attention_size)
    dims = sp._dims(N, Tp, 1)
    bufs = _alloc_bufs(dims, dev)
    for l, s in enumerate(dec_state):
        if sp.cell == "lstm":
            bufs["cs"][l, 0].copy_(s[0])
            bufs["hs"][l, 0].copy_(s[1])
        else:
            bufs["hs"][l, 0].copy_(s)
    if token_ids is None:
        # recover the id from the embedded vector (API parity with the reference's decode signature)
        emb = P["emb"].detach()
        token_ids = torch.cdist(prev_token.to(dev), emb).argmin(-1)
    tokens_in = torch.as_tensor(token_ids).to(dev).to(torch.int32).reshape(1, N).contiguous()
    tokens_out = torch.zeros(1, N, dtype=torch.int32, device=dev)
    enc_len_i32 = torch.as_tensor(enc_len).to(torch.float64).to(torch.int32).to(dev).contiguous()
    align0 = None if prev_align is None else torch.as_tensor(prev_align).to(dev).float().contiguous()
    fa = _hip.SpellerFwdArgs()
    Pd = {k: (v.detach() if torch.is_tensor(v) else [t.detach() for t in v]) for k, v in P.items()}
    keep = _fill_fwd_args(fa, dims, Pd, enc, keys, enc_len_i32, tokens_in, tokens_out, bufs, True, 0, keep_state0=True,
                          align0=align0)
    _hip.check(_hip.lib().las_speller_fwd(ctypes.byref(fa), _hip.stream()), "las_speller_fwd")
    del keep
    NL = a.num_dec_layers
    if sp.cell == "lstm":
        new_state = tuple((bufs["cs"][l, 1], bufs["hs"][l, 1]) for l in range(NL))
    else:
        new_state = tuple(bufs["hs"][l, 1] for l in range(NL))
    return bufs["logits"][0], new_state, bufs["alphas"][0]


def attention_forward_step(att, hidden, state, align, seqlen):
    """AdditiveAttention / LocationAwareAttention.__call__ (reference las/layers.py:234-257 / :281-311) for ONE
    step, forward only, through the same fused row kernel as the Speller loop (U=1, throw-away cell)."""
    dev = hidden.device
    _hip.require_gpu(hidden, state)
    N, Tp, Hd = hidden.shape
    P = att.params()
    A = att.att_size
    S = att.s_dim
    prec = L._prec()
    enc = hidden.contiguous()
    keys = torch.empty(N, Tp, A, device=dev)
    _hip.gemm(prec, enc, P["Wh"].detach(), keys, False, False, N * Tp, A, Hd, Hd, A, A)
    E, Vv = 4, 4
    loc = att.mode == "loc"
    dims = {"B": N, "Tp": Tp, "Hd": Hd, "A": A, "D": S, "NL": 1, "E": E, "V": Vv, "U": 1, "cell": _hip.CELL_RNN,
            "mode": _hip.ATT_LOC if loc else _hip.ATT_ADD, "prec": prec, "Kc": att.kernel_size if loc else 0,
            "C": att.num_channels if loc else 0}
    bufs = _alloc_bufs(dims, dev)
    bufs["hs"][0, 0].copy_(state.reshape(N, S))
    z = lambda *s: torch.zeros(*s, device=dev)
    Pd = {"Ws": P["Ws"].detach(), "u": P["u"].detach(), "emb": z(Vv, E), "Wv": z(S, Vv), "bv": z(Vv),
          "cellW": [z(E + Hd + S, S)], "cellb": [z(S)]}
    if loc:
        Pd.update(loc_w=P["loc_w"].detach(), loc_b=P["loc_b"].detach(), Wf=P["Wf"].detach())
    tokens_in = torch.zeros(1, N, dtype=torch.int32, device=dev)
    enc_len_i32 = torch.as_tensor(seqlen).to(torch.float64).to(torch.int32).to(dev).contiguous()
    align0 = None if (align is None or not loc) else torch.as_tensor(align).to(dev).float().contiguous()
    fa = _hip.SpellerFwdArgs()
    keep = _fill_fwd_args(fa, dims, Pd, enc, keys, enc_len_i32, tokens_in, None, bufs, False, 0, keep_state0=True,
                          align0=align0)
    _hip.check(_hip.lib().las_speller_fwd(ctypes.byref(fa), _hip.stream()), "las_speller_fwd")
    del keep
    return bufs["xin0"][0, :, E:E + Hd], bufs["alphas"][0]


# ------------------------------------------------------------------------------------------------
# loss / optimiser
# ------------------------------------------------------------------------------------------------
class _CELoss(torch.autograd.Function):
    """K8: label-smoothed masked CE over the Speller's time-major logits, value + gradient in one pass."""

    @staticmethod
    def forward(ctx, logits_bt, y_i32, V_, smooth, scale):
        # logits_bt is the [B,U,V] VIEW of the time-major buffer; consume it in place through strides
        B, U, _ = logits_bt.shape
        sb, st_, sv = logits_bt.stride()
        assert sv == 1
        loss, dl, sums = _ce_loss(logits_bt, y_i32, V_, smooth, scale)
        ctx.save_for_backward(dl)
        ctx.sums = sums
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None


def _ce_loss(logits_bt, y_i32, V_, smooth, scale):
    """-> (loss = sum(ce * mask) * scale as a 0-d view, dlogits with `scale` already in it, sums [3]).  One kernel pair: the sums are
    written (no fill in front), the scaled loss is sums[2] (no multiply behind)."""
    B, U, _ = logits_bt.shape
    sb, st_, sv = logits_bt.stride()
    assert sv == 1
    dev = logits_bt.device
    sums = torch.empty(3, device=dev)
    dl = torch.empty_strided(logits_bt.shape, logits_bt.stride(), device=dev)
    nb = _hip.lib().las_ce_loss_workspace_bytes(B, U)
    ws = _hip.workspace(dev, nb, "ce")
    _hip.check(_hip.lib().las_ce_loss(_hip.p(logits_bt), sb, st_, _hip.p(y_i32), y_i32.shape[1], B, U, V_, 0.01,
                                      int(bool(smooth)) | 4, _hip.p(sums), _hip.p(scale), _hip.p(dl), _hip.p(ws), ws.numel(),
                                      _hip.stream()), "las_ce_loss")
    return sums[2], dl, sums


class LAS:
    """reference las/las.py:209-369."""

    def __init__(self, args, Listener, Speller, id_to_token):
        self.args = args
        self.listener = Listener(args)
        self.speller = Speller(args)
        self.id_to_token = id_to_token
        self.dp = None            # optional las.parallel.DataParallel (set by train.py)
        self.last = {}
        self.last_variants = {}   # layers.VARIANTS of the last train step: which cross-stream hand-overs it used
        import collections
        self._recent = collections.deque(maxlen=6)   # (global step, batch) of the newest steps: what _recover can re-run
        self._pending_status = 0
        self._step_ends = collections.deque()        # events at the ends of the newest steps (bounds the host's run-ahead)
        self._in_recovery = False
        self._guard_probes = collections.deque()     # data parallel: (global step, pinned copy of the all-reduced guard slot, event)
        self._fallback_until = -1  # global step up to which steps run on las.layers.fallback_schedule (set by _recover)
        self._backoff = 16
        self._warned_recover = False
        self.recovered_steps = 0  # steps re-run after a time-out status (tests / logs)
        self.last_out = None      # what the newest train step returned -- of its RE-RUN when the step was lost and recovered

    # -- helpers -----------------------------------------------------------------------------------
    @staticmethod
    def _to_dev(x, dev, dtype=None):
        t = torch.as_tensor(x)
        if dtype is not None:
            t = t.to(dtype)
        return t.to(dev)

    def _device(self):
        st = V.default_store()
        if st.device is None:
            st.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if st.device is None or st.device.type != "cuda":
            raise RuntimeError("LAS needs a ROCm device: the hot path runs in liblas_hip.so and has no CPU fallback")
        return st.device

    def build_variables(self):
        """Create every variable up front (all shapes follow from args) and flatten the store, so that
        parameters, gradients and Adam slots live in flat buckets before the first step."""
        st = V.default_store()
        if st.flat is not None:
            return
        self._device()
        a = self.args
        if a.enc_type.lower() == "pblstm":
            H = a.enc_units
            cell = L.get_cell()
            scopes = ["Listener/blstm"] + ["Listener/pyramid_blstm_%d" % l for l in range(a.num_enc_layers)]
            for i, sc in enumerate(scopes):
                L._blstm_params(sc, a.feat_dim * 3 if i == 0 else 2 * H, H, cell)
                st.get(sc + "/dense/kernel", (2 * H if i == 0 else 4 * H, 2 * H))
                st.get(sc + "/dense/bias", (2 * H,), init="zeros")
        elif a.enc_type.lower() == "cnn":
            # run the (cheap) variable creation path once on a dummy block: every shape follows from args
            with torch.no_grad():
                dummy = torch.zeros(1, 8, a.feat_dim, 3, device=st.device)
                self.listener(dummy, [8], "cnn", is_training=False)
        self.speller._params()
        st.flatten()

    @staticmethod
    def _loss_scale(n_total):
        return (1.0 / (n_total + 1e-9)).reshape(1).to(torch.float32)

    def _get_loss(self, logits, y, n_total=None, scale=None):
        """las/las.py:320-333.  n_total (device scalar) overrides the local non-PAD count so that
        data-parallel ranks normalise by the GLOBAL token count; scale = _loss_scale(n_total) when the
        caller has already computed it (off the chain)."""
        B, U, V_ = logits.shape
        y = y[:, :U].contiguous()
        if scale is None:
            if n_total is None:
                n_total = (y != 0).sum().to(torch.float32)
            scale = self._loss_scale(n_total)
        return _CELoss.apply(logits, y, V_, self.args.label_smoothing, scale)

    def _scheduled_learning_rate(self, start=50000, decay_step=100000, decay_rate=0.5, min_rate=0.01, global_step=0):
        """las/las.py:351-369 (non-staircase exponential decay)."""
        gs = max(global_step - start, 0)
        return max(self.args.lr * decay_rate ** (gs / decay_step), min_rate * self.args.lr)

    # -- one optimisation step ---------------------------------------------------------------------
    def train(self, xs, ys, coins=None, sampled=None):
        """reference las/las.py:226-304, executed eagerly.

        Returns (loss, train_op, global_step, logits, alphas, summaries, sample_rate) -- `train_op`
        is None (the update has already been applied), `summaries` a dict of the quantities the
        reference attaches to tf.summary (las/las.py:292-299).

        Round 6: a recurrent sweep or a one-launch Speller loop whose workgroups were not all resident (a
        neighbour on the device: a monitoring agent's kernel, a user's second stream) reports through the status word, and the
        device skips that step's update and -- the word is sticky -- every later one.  Instead of raising, a single-process run
        re-runs the lost steps: the first of them on the schedule that needs no co-residency (per-step Speller launches, no
        cross-stream hand-overs), the others as usual -- see _recover."""
        out = self._train_step(xs, ys, coins, sampled)
        code = self._pending_status
        self._pending_status = 0
        if code:
            out = self._recover(code) or out
        self.last_out = out               # (check_status may replace it: what train() returned for a step that is lost later is not valid)
        return out

    def _recover(self, code, first=None):
        """The device reported a time-out (`code`) some steps ago.  Everything it has been asked to do since then was computed
        but NOT applied (las_clip_adam's guard; the status word stays set until the host clears it), and `store.applied` counts the
        updates that were: the first lost step is the `applied`-th optimiser launch.  Re-run from there on the batches kept in
        `_recent`, on the fall-back schedule (las.layers.fallback_schedule: per-step Speller launches, no cross-stream hand-overs --
        nothing that needs every CU or a partner stream), and STAY on it for `_backoff` steps (a neighbour that was there once is
        probably still there; the back-off doubles with every recovery, up to 4096 steps) before the fast schedule is tried again.
        Returns the result of the newest re-run step.  Raises (as rounds 1-5 did) when the lost step is older than what was kept, or
        when a step fails on the fall-back schedule too."""
        import warnings
        dev = self._device()
        st = V.default_store()
        torch.cuda.synchronize(dev)
        msg = _hip.status_message(str(dev), code)
        n_applied = int(st.applied.item()) - st.adam_base if st.applied is not None else -1
        if first is None:
            first = st.adam_launches[n_applied] if 0 <= n_applied < len(st.adam_launches) else None
        else:                                        # data parallel: the ranks agreed on the step (the first whose all-reduced guard was set)
            n_applied = st.adam_launches.index(first) if first in st.adam_launches else -1
            self._guard_probes.clear()
        todo = [e for e in self._recent if first is not None and e[0] >= first]
        _hip.clear_status(dev)
        on_fallback = first is not None and first < self._fallback_until
        if not todo or todo[0][0] != first or on_fallback:
            raise RuntimeError(msg + " -- the cluster workgroups were not all resident (shared / partitioned GPU?); the lost step could "
                               "not be re-run (%s)" % ("it was already on the fall-back schedule" if on_fallback else "its batch is no longer held"))
        if not self._warned_recover:
            self._warned_recover = True
            warnings.warn(msg + " -- re-running %d step(s) from global step %d on the per-step kernels (no co-residency needed) and staying "
                          "on them for %d steps.  Another kernel was resident on the device; this warning is printed once."
                          % (len(todo), first, self._backoff))
        self.recovered_steps += len(todo)
        st.global_step = first
        st.adam_launches = st.adam_launches[:n_applied]
        self._recent.clear()
        self._fallback_until = first + self._backoff
        self._backoff = min(2 * self._backoff, 4096)
        out = None
        self._in_recovery = True
        try:
            for gs, xs, ys, coins, sampled in todo:
                out = self._train_step(xs, ys, coins, sampled)
        finally:
            self._in_recovery = False
        torch.cuda.synchronize(dev)                  # the re-run steps must have gone through before anything is built on them
        self._pending_status = 0
        code2 = int(_hip.status_word(dev)[0].item()) if self.dp is None else int(float(st.guard[0].item()) != 0.0) * 3
        if code2:
            _hip.clear_status(dev)
            raise RuntimeError(_hip.status_message(str(dev), code2) + " -- again, on the fall-back schedule: the device cannot hold the "
                               "recurrent sweeps' clusters beside what else is running on it")
        return out

    def _train_step(self, xs, ys, coins=None, sampled=None):
        if RECOVER_STEPS and self.dp is not None and not self._in_recovery:
            # data parallel: every rank looks at the ALL-REDUCED guard slot of the step DP_LAG steps back (a pinned copy enqueued behind
            # that step's exchange; its event is long past in steady state) -- the same value on every rank, examined at the same call,
            # so the ranks decide together and re-run in lock step
            dev = self._device()
            while len(self._guard_probes) > DP_LAG:
                gs, pin, ev = self._guard_probes.popleft()
                ev.synchronize()
                if float(pin[0]) != 0.0:
                    self._recover(int(_hip.status_word(dev)[0].item()) or 3, first=gs)
                    break
        with L.fallback_schedule(V.default_store().global_step < self._fallback_until):
            return self._train_step_impl(xs, ys, coins, sampled)

    def _train_step_impl(self, xs, ys, coins=None, sampled=None):
        dev = self._device()
        st = V.default_store()
        self.build_variables()
        if RECOVER_STEPS and self.dp is None:
            # the host may run ahead of the device, but not further than _recover can reach back: wait for the end of the step four
            # steps ago (normally long past -- the host is one or two steps ahead -- so this costs nothing; behind a time-out, whose
            # bounded polls take a second to drain, it keeps the host from enqueueing hundreds of steps that will all be skipped)
            if len(self._step_ends) >= 4:
                self._step_ends.popleft().synchronize()
        audio, audiolen = xs
        y, tokenlen = ys
        audio = self._to_dev(audio, dev, torch.float32)
        y = self._to_dev(y, dev, torch.int32)
        dec_steps = int(torch.as_tensor(tokenlen).max())                                  # las/las.py:248
        enc_type = self.args.enc_type.lower()
        # everything that does not depend on the encoder goes first, off the chain between the decode loop and its
        # gradient: gradient bucket reset, global token count (one small all-reduce under data parallelism)
        st.flatten()
        L.begin_step(dev)
        self.speller.rank = self.dp.rank if self.dp is not None else 0
        # ... and on the auxiliary ("chain") stream: a dozen tiny kernels (0.1 ms back to back) that only the Speller and the
        # backward pass need run next to the first Listener sweep instead of in front of it
        with _hip.on_chain_stream():
            if RECOVER_STEPS:
                # what _recover needs to re-run this step: the batch as it is NOW.  A caller may hand in device tensors that it overwrites
                # a few steps later (las.input_pipeline.DeviceFeeder keeps a ring of three device slots): those are copied -- here, on the
                # auxiliary stream beside the first Listener sweep, off the dependency chain; tensors this call made itself are held as is
                a_hold = audio.clone() if (torch.is_tensor(xs[0]) and xs[0].is_cuda) else audio
                y_hold = y.clone() if (torch.is_tensor(ys[0]) and ys[0].is_cuda) else y
                hostcopy = lambda v: None if v is None else (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else np.array(v, copy=True))
                self._recent.append((st.global_step, (a_hold, hostcopy(audiolen)), (y_hold, hostcopy(tokenlen)), hostcopy(coins), hostcopy(sampled)))
            st.zero_grad()
            n_local = (y[:, :dec_steps] != 0).sum().to(torch.float32)
            n_total = self.dp.all_reduce_scalar(n_local) if self.dp is not None else n_local
            loss_scale = self._loss_scale(n_total)
            y_u = y[:, :dec_steps].contiguous()               # (the loss's label block: copied here, off the chain)
            # the Speller's host-side preparation: token schedule, encoder lengths, masks
            prep = self.speller.prepare(audio.shape[0], self.listener.output_length(audiolen, enc_type), dec_steps, dev, y,
                                        True, coins, sampled)
        with _hip.roctx_range("listener fwd"):
            h, enc_state, enc_len = self.listener(audio, audiolen, enc_type)              # is_training default True
        _hip.join_chain_stream()
        with _hip.roctx_range("speller fwd"):
            logits, ctc_logits, alphas = self.speller(h, enc_len, dec_steps, y, coins=coins, sampled=sampled, prepared=prep)

        with _hip.roctx_range("loss"):
            # the loss is the root of the graph: its gradient w.r.t. the logits (scale included) comes out of the same kernel pair and is
            # handed to the Speller's node directly -- no ones_like fill, no [B, U, V] multiply by 1.0, no scalar multiply on the chain
            loss, dlogits, _ = _ce_loss(logits.detach(), y_u, logits.shape[2], self.args.label_smoothing, loss_scale)   # sum_local / n_total
        early = []
        if self.dp is not None:
            def before_tail(P4, st=st, dev=dev):
                # Called by the bottom recurrent layer's backward right before it enqueues the end-of-step tail (its own weight
                # gradients).  Those parameters were created first, so they -- and the guard slot in front of them -- are the
                # head of the bucket; everything behind is final once the work enqueued so far (main, side, chain streams) has
                # run.  That part is all-reduced NOW on the communication stream, under the last sweep's tail.
                mine = {n for n in st.order if any(st.vars[n] is q for q in P4)}
                n_head = 4 + (max(st.offsets[n] + st.vars[n].numel() for n in mine) + 3) // 4 * 4
                if any(st.offsets[n] + 4 < n_head for n in st.order if n not in mine) or n_head >= st.grad_bucket.numel():
                    return                                   # (unexpected layout: fall back to one exchange at the end)
                comm = _hip.comm_stream()
                for s_ in (torch.cuda.current_stream(), _hip.side_stream(), _hip.chain_stream()):
                    comm.wait_stream(s_)
                with torch.cuda.stream(comm):
                    early.append((n_head, self.dp.all_reduce_(st.grad_bucket[n_head:], async_op=True)))
            L.BEFORE_TAIL_HOOK[0] = before_tail
        with _hip.roctx_range("backward"):
            try:
                logits.backward(dlogits)
            finally:
                L.BEFORE_TAIL_HOOK[0] = None
            _hip.join_side_stream()                   # weight gradients accumulated on the side stream
            L.check_handovers_consumed()
        self.last_variants = dict(L.VARIANTS)          # which cross-stream hand-overs this step really used (tests / bench.py assert on it)
        if self.dp is not None:
            # one flat bucket (C1); its first slot carries this rank's sweep status, so that a time-out on ANY rank makes EVERY
            # rank skip the update (las_clip_adam's guard) and the replicas stay identical
            st.guard.copy_(_hip.status_word(dev)[0:1].ne(0))
            if early:
                n_head, work = early[0]
                self.dp.all_reduce_(st.grad_bucket[:n_head])          # guard + the bottom layer's gradients (the tail's output)
                work.wait()                                           # the launch stream waits for the early part
            else:
                self.dp.all_reduce_(st.grad_bucket)
            loss_val = self.dp.all_reduce_scalar(loss.detach())
        else:
            loss_val = loss.detach()

        lr = self._scheduled_learning_rate(start=50000, decay_step=100000, decay_rate=0.5, min_rate=0.01,
                                           global_step=st.global_step)
        with _hip.roctx_range("clip + adam"):
            self._apply_adam(st, lr)
        # a sweep / Speller-loop time-out of an earlier step surfaces here: raised under data parallelism (the ranks would have to agree
        # on what to re-run), handed to train() otherwise
        recover = RECOVER_STEPS
        self._pending_status = _hip.poll_status(dev, raise_on_error=not recover)
        if recover and self.dp is not None:
            self._pending_status = 0                    # (data parallel: the guard probes decide, for every rank at once)
            pin = torch.zeros(1, dtype=torch.float32).pin_memory()
            pin.copy_(st.guard, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._guard_probes.append((st.global_step, pin, ev))
        elif recover:
            ev = torch.cuda.Event()
            ev.record()
            self._step_ends.append(ev)
        st.global_step += 1
        sample_rate = self.speller._scheduled_sampling()
        summaries = {"loss": loss_val, "global_step": st.global_step, "lr": lr}
        self.last = {"logits": logits, "y": y}
        return loss_val, None, st.global_step, logits.detach(), alphas.detach(), summaries, sample_rate

    def _apply_adam(self, st, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """clip_by_global_norm + Adam (las/las.py:272-283; TF epsilon-hat form)."""
        t = st.global_step + 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        n = st.flat.numel()
        dev = st.flat.device
        sumsq = torch.empty(1, device=dev)
        clip = float(self.args.grad_clip)
        lib = _hip.lib()
        if clip > 0:
            ws = _hip.workspace(dev, lib.las_sumsq_workspace_bytes(n), "sumsq")
            _hip.check(lib.las_sumsq(_hip.p(st.flat_grad), n, _hip.p(sumsq), _hip.p(ws), ws.numel(), _hip.stream()), "las_sumsq")
        if st.applied is None:
            st.applied = torch.zeros(1, dtype=torch.int32, device=dev)
            st.adam_launches, st.adam_base = [], 0
        if len(st.adam_launches) >= 4096:                            # (bounded bookkeeping: the device counter keeps counting)
            st.adam_launches, st.adam_base = st.adam_launches[2048:], st.adam_base + 2048
        st.adam_launches.append(st.global_step)
        _hip.check(lib.las_clip_adam(_hip.p(st.flat), _hip.p(st.flat_grad), _hip.p(st.adam_m), _hip.p(st.adam_v), n,
                                     _hip.p(sumsq), clip if clip > 0 else 0.0, lr_t, beta1, beta2, eps,
                                     _hip.p(_hip.status_word(dev)), _hip.p(st.guard), _hip.p(st.applied), _hip.stream()),
                   "las_clip_adam")
        st.weights_changed()                          # bf16 weight shadows / prepared sweep workspaces are rebuilt from the updated masters
        self.last_grad_sumsq = sumsq

    def train_stacked(self, batches, coins=None, sampled=None):
        """ONE optimisation step on k batches of one bucket stacked along the batch axis: batches = [(xs, ys), ...] with equal frame
        counts T.  The rows of a batch never interact in the forward pass (las/las.py:93-117 run every utterance through the same
        weights) and the loss is the token sum over ALL rows divided by their token count (las/las.py:329-331), so this IS the update
        of k data-parallel ranks that hold one of the batches each (las/parallel.py) -- on one GPU, where the recurrent sweeps are
        latency-bound on a fifth of the compute units and k times the rows cost them (almost) nothing.  Returns what train() does."""
        if len(batches) == 1:
            return self.train(batches[0][0], batches[0][1], coins=coins, sampled=sampled)
        dev = self._device()
        T = {tuple(b[0][0].shape[1:]) for b in batches}
        if len(T) != 1:
            raise ValueError("train_stacked: the batches must come from one bucket (equal frame count and feature shape), got %s" % sorted(T))
        W = max(int(b[1][0].shape[1]) for b in batches)

        def cat(parts, dtype, pad_to=None):
            ts = [self._to_dev(x, dev, dtype) for x in parts]
            if pad_to is not None:
                ts = [torch.nn.functional.pad(t, (0, pad_to - t.shape[1])) for t in ts]
            return torch.cat(ts, 0)

        audio = cat([b[0][0] for b in batches], torch.float32)
        audiolen = np.concatenate([np.asarray(torch.as_tensor(b[0][1]).cpu()) for b in batches])
        y = cat([b[1][0] for b in batches], torch.int32, W)
        tokenlen = np.concatenate([np.asarray(torch.as_tensor(b[1][1]).cpu()) for b in batches])
        return self.train((audio, audiolen), (y, tokenlen), coins=coins, sampled=sampled)

    def check_status(self):
        """Synchronising check that no recurrent sweep / Speller loop reported a time-out: the lost steps are re-run (_recover; single
        process) or RuntimeError is raised."""
        dev = self._device()
        if RECOVER_STEPS and self._recent and self.dp is not None:
            torch.cuda.synchronize(dev)
            while self._guard_probes:                    # (the same probes, in the same order, on every rank)
                gs, pin, ev = self._guard_probes.popleft()
                if float(pin[0]) != 0.0:
                    self.last_out = self._recover(int(_hip.status_word(dev)[0].item()) or 3, first=gs) or self.last_out
                    break
            return
        if RECOVER_STEPS and self._recent:
            torch.cuda.synchronize(dev)
            code = int(_hip.status_word(dev)[0].item())
            if code:
                self.last_out = self._recover(code) or self.last_out
            return
        _hip.check_status(dev)

    def sample_texts(self):
        """The HYP / REF strings the reference builds for its text summaries (las/las.py:286-289)."""
        logits, y = self.last["logits"], self.last["y"]
        hyp = convert_idx_to_token_tensor(torch.argmax(logits[0], -1).cpu().numpy(), self.id_to_token, self.args.unit)
        ref = convert_idx_to_token_tensor(y[0].cpu().numpy(), self.id_to_token, self.args.unit)
        return hyp, ref

    # -- greedy inference --------------------------------------------------------------------------
    def inference(self, xs):
        """reference las/las.py:306-318.  The reference hard-codes encoder='cnn' here;
        this build honours --enc_type so that the trained listener is the one evaluated."""
        dev = self._device()
        audio, audiolen = xs
        audio = self._to_dev(audio, dev, torch.float32)
        mx = float(torch.as_tensor(audiolen).max())
        dec_steps = int(np.int32(np.float32(self.args.convert_rate) * np.float32(mx)))     # las/las.py:310-312
        dec_steps = max(dec_steps, 1)
        with torch.no_grad():
            h, enc_state, enc_len = self.listener(audio, audiolen, encoder=self.args.enc_type.lower(), is_training=False)
            logits, ctc_logits, alphas = self.speller(h, enc_len, dec_steps, is_training=False)
        y_hat = torch.argmax(logits, -1)
        _hip.check_status(dev)
        return logits, y_hat
