"""lang.char_rnn_model -- the char RNNLM (reference lang/char_rnn_model.py) on the MI355X engine.

Inference (`step`, `step_tensors`): embedding (or one-hot) -> L x BasicLSTMCell(forget_bias=0) -> `logits = out . softmax_w +
softmax_b`, one unrolling at a time for N hypotheses -- what shallow fusion in beam search needs (reference
las/beam_search.py:109-116,226-236; decode.py:27-39).
Training (`is_training=True`: `train_step`, `run_epoch`; reference lang/char_rnn_model.py:54-193 graph, :195-244 loop):
truncated BPTT over `num_unrollings` steps with the state carried from batch to batch, mean sparse cross entropy,
clip_by_global_norm + Adam (TF epsilon-hat form).  Contractions run through las_gemm, gate math through
las_lstm_pointwise / las_lstm_pointwise_bwd, the loss through las_ce_loss, the update through las_sumsq + las_clip_adam
(liblas_hip.so); the LM keeps its own VariableStore (flat parameter / gradient / Adam buckets)."""
import logging
import math
import time

import ctypes
import numpy as np
import torch

from las import _hip
from las import layers as L
from las import variables as V


def create_vocab():
    """The LM's 28-symbol vocabulary ['.', ' ', 'A'..'Z'] (reference train_lm.py:378-386); LAS char ids are
    LM ids + 2 (las/beam_search.py:116,228)."""
    chars = ['.', ' '] + [chr(ord('A') + i) for i in range(26)]
    return {c: i for i, c in enumerate(chars)}, dict(enumerate(chars)), len(chars)


class CharRNN(object):
    def __init__(self, is_training, batch_size, num_unrollings, vocab_size, hidden_size, max_grad_norm=5.0,
                 embedding_size=0, num_layers=2, learning_rate=0.0, model='lstm', dropout=0.0, input_dropout=0.0,
                 use_batch=True, scope="lm", store=None):
        if model != 'lstm':
            raise NotImplementedError("the shipped LM configuration is model='lstm' (reference decode.py:27-39, train_lm.py default)")
        self.is_training = bool(is_training)
        self.batch_size, self.num_unrollings = (batch_size, num_unrollings) if use_batch else (1, 1)     # :17-21
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.embedding_size, self.num_layers = embedding_size, num_layers
        self.input_size = embedding_size if embedding_size > 0 else vocab_size     # char_rnn_model.py:30-35
        self.max_grad_norm, self.learning_rate = float(max_grad_norm), float(learning_rate)
        self.dropout = float(dropout)
        self.input_dropout = float(input_dropout) if embedding_size > 0 else 0.0    # no dropout on the one-hot representation (:33)
        self.scope = scope
        self.store = store or V.default_store()
        self.global_step = 0
        self._sum_mean_loss, self._count = 0.0, 0.0                                # the reference's loss monitor (:151-166)

    # -- variables (TF names of the reference graph under the given scope) -----------------------------
    def params(self):
        st, sc, H = self.store, self.scope, self.hidden_size
        p = {"cells": []}
        if self.embedding_size > 0:
            p["embedding"] = st.get(sc + "/embedding", (self.vocab_size, self.embedding_size))
        for l in range(self.num_layers):
            I = self.input_size if l == 0 else H
            base = "%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % (sc, l)
            p["cells"].append((st.get(base + "kernel", (I + H, 4 * H)), st.get(base + "bias", (4 * H,), init="zeros")))
        p["softmax_w"] = st.get(sc + "/softmax/softmax_w", (H, self.vocab_size))
        p["softmax_b"] = st.get(sc + "/softmax/softmax_b", (self.vocab_size,), init="zeros")
        return p

    # -- training ------------------------------------------------------------------------------------------
    def train_step(self, inputs, targets, state=None, train=True):
        """One batch of `num_unrollings` steps (reference graph :110-190): inputs / targets int [B,U].  state: list over layers
        of (c,h) [B,H] carried from the previous batch (None = zeros).  train=False evaluates only (no update).
        Returns (mean_loss float tensor, new_state)."""
        P = self.params()
        st = self.store
        dev = P["softmax_w"].device
        if train:
            st.flatten()
            st.zero_grad()
        prec = L._prec()
        lib = _hip.lib()
        ids = torch.as_tensor(np.asarray(inputs), device=dev).long()
        tgt = torch.as_tensor(np.asarray(targets), device=dev).to(torch.int32).contiguous()
        B, U = ids.shape
        H, NL, Vn = self.hidden_size, self.num_layers, self.vocab_size
        if state is None:
            state = [(torch.zeros(B, H, device=dev), torch.zeros(B, H, device=dev)) for _ in range(NL)]
        with torch.no_grad():
            if self.embedding_size > 0:
                x_all = P["embedding"].detach()[ids]                                  # [B,U,E]
            else:
                x_all = torch.nn.functional.one_hot(ids, Vn).to(torch.float32)
            in_mask = None
            if train and self.input_dropout > 0:
                keep = 1.0 - self.input_dropout
                in_mask = (torch.rand_like(x_all) < keep).float() / keep
                x_all = x_all * in_mask
            Is = [self.input_size] + [H] * (NL - 1)
            xin = [torch.empty(U, B, Is[l] + H, device=dev) for l in range(NL)]      # saved [x ; h_prev] rows
            zs = [torch.empty(U, B, 4 * H, device=dev) for l in range(NL)]           # saved pre-activations
            cps = [torch.empty(U, B, H, device=dev) for l in range(NL)]              # saved c_{t-1}
            masks = [None] * NL
            if train and self.dropout > 0:                                            # DropoutWrapper(output_keep_prob) (:81-85)
                keep = 1.0 - self.dropout
                masks = [(torch.rand(U, B, H, device=dev) < keep).float() / keep for _ in range(NL)]
            top = torch.empty(B, U, H, device=dev)                                    # batch-major, as tf.concat(axis=1) flattens (:127-129)
            c = [s_[0].contiguous() for s_ in state]
            h = [s_[1].contiguous() for s_ in state]
            for t in range(U):
                x = x_all[:, t]
                for l, (k, b) in enumerate(P["cells"]):
                    xi = xin[l][t]
                    xi[:, :Is[l]] = x
                    xi[:, Is[l]:] = h[l]
                    cps[l][t] = c[l]
                    _hip.gemm(prec, xi, k.detach(), zs[l][t], False, False, B, 4 * H, Is[l] + H, Is[l] + H, 4 * H, 4 * H, bias=b.detach())
                    c_new, h_new = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
                    _hip.check(lib.las_lstm_pointwise(_hip.p(zs[l][t]), _hip.p(cps[l][t]), B, H, 0.0, _hip.p(c_new), _hip.p(h_new),
                                                      _hip.stream()), "las_lstm_pointwise")
                    c[l], h[l] = c_new, h_new
                    x = h_new if masks[l] is None else h_new * masks[l][t]
                top[:, t] = x
            flat = top.view(B * U, H)
            logits = torch.empty(B, U, Vn, device=dev)
            _hip.gemm(prec, flat, P["softmax_w"].detach(), logits, False, False, B * U, Vn, H, H, Vn, Vn, bias=P["softmax_b"].detach())
            # mean sparse cross entropy over all B*U positions (:146-149) + its gradient
            sums = torch.zeros(2, device=dev)
            scale = torch.full((1,), 1.0 / (B * U), device=dev)
            dlog = torch.empty_like(logits) if train else None
            ws = _hip.workspace(dev, lib.las_ce_loss_workspace_bytes(B, U), "ce")
            _hip.check(lib.las_ce_loss(_hip.p(logits), U * Vn, Vn, _hip.p(tgt), U, B, U, Vn, 0.0, 2, _hip.p(sums), _hip.p(scale),
                                       _hip.p(dlog), _hip.p(ws), ws.numel(), _hip.stream()), "las_ce_loss")
            mean_loss = sums[0] * scale[0]
            new_state = [(c[l], h[l]) for l in range(NL)]
            if not train:
                return mean_loss, new_state
            # ---- backward
            g = {n: v.grad for n, v in st.vars.items()}
            sc = self.scope
            dl2 = dlog.view(B * U, Vn)
            _hip.gemm(prec, flat, dl2, g[sc + "/softmax/softmax_w"], True, False, H, Vn, B * U, H, Vn, Vn, beta=1.0)
            _hip.colsum(dl2, B * U, Vn, Vn, g[sc + "/softmax/softmax_b"], beta=1.0)
            dtop = torch.empty(B, U, H, device=dev)
            _hip.gemm(prec, dl2, P["softmax_w"].detach(), dtop.view(B * U, H), False, True, B * U, H, Vn, Vn, Vn, H)
            dzs = [torch.empty(U, B, 4 * H, device=dev) for l in range(NL)]
            dh_next = [torch.zeros(B, H, device=dev) for _ in range(NL)]
            dc_next = [None] * NL
            dx_all = torch.empty(U, B, self.input_size, device=dev) if self.embedding_size > 0 else None
            for t in range(U - 1, -1, -1):
                dabove = dtop[:, t].contiguous()
                for l in range(NL - 1, -1, -1):
                    k = P["cells"][l][0].detach()
                    dh = (dabove if masks[l] is None else dabove * masks[l][t]) + dh_next[l]
                    dcp = torch.empty(B, H, device=dev)
                    _hip.check(lib.las_lstm_pointwise_bwd(_hip.p(zs[l][t]), _hip.p(cps[l][t]), _hip.p(dh), _hip.p(dc_next[l]), B, H, 0.0,
                                                          _hip.p(dzs[l][t]), _hip.p(dcp), _hip.stream()), "las_lstm_pointwise_bwd")
                    dc_next[l] = dcp
                    dxi = torch.empty(B, Is[l] + H, device=dev)
                    _hip.gemm(prec, dzs[l][t], k, dxi, False, True, B, Is[l] + H, 4 * H, 4 * H, 4 * H, Is[l] + H)
                    dh_next[l] = dxi[:, Is[l]:].contiguous()
                    dabove = dxi[:, :Is[l]].contiguous()
                if dx_all is not None:
                    dx_all[t] = dabove if in_mask is None else dabove * in_mask[:, t]
            for l in range(NL):
                base = "%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % (sc, l)
                K = Is[l] + H
                _hip.gemm(prec, xin[l].view(U * B, K), dzs[l].view(U * B, 4 * H), g[base + "kernel"], True, False, K, 4 * H, U * B, K, 4 * H,
                          4 * H, beta=1.0)
                _hip.colsum(dzs[l].view(U * B, 4 * H), U * B, 4 * H, 4 * H, g[base + "bias"], beta=1.0)
            if dx_all is not None:
                g[sc + "/embedding"].index_add_(0, ids.t().reshape(-1), dx_all.view(U * B, -1))
            # ---- clip_by_global_norm + Adam (:177-190)
            t_ = self.global_step + 1
            b1, b2, eps = 0.9, 0.999, 1e-8
            lr_t = self.learning_rate * math.sqrt(1.0 - b2 ** t_) / (1.0 - b1 ** t_)
            n = st.flat.numel()
            sumsq = torch.empty(1, device=dev)
            wss = _hip.workspace(dev, lib.las_sumsq_workspace_bytes(n), "sumsq")
            _hip.check(lib.las_sumsq(_hip.p(st.flat_grad), n, _hip.p(sumsq), _hip.p(wss), wss.numel(), _hip.stream()), "las_sumsq")
            _hip.check(lib.las_clip_adam(_hip.p(st.flat), _hip.p(st.flat_grad), _hip.p(st.adam_m), _hip.p(st.adam_v), n, _hip.p(sumsq),
                                         self.max_grad_norm, lr_t, b1, b2, eps, None, None, None, _hip.stream()), "las_clip_adam")
            st.weights_changed()
            self.global_step += 1
        return mean_loss, new_state

    def run_epoch(self, session, data_size, batch_generator, is_training, verbose=0, freq=10, summary_writer=None, debug=False,
                  divide_by_n=1):
        """One pass over `data_size` characters (the reference's epoch driver, lang/char_rnn_model.py:195-244: same signature and
        return value).  Returns (perplexity of the mean batch loss, None, global_step).  The losses are summed ON THE DEVICE into one
        running scalar that is read back once per progress report / once at the end: the loop never waits for a step, and a pass of
        millions of one-character batches (train_lm's test pass: batch_size = num_unrollings = 1) keeps one tensor alive, not one each."""
        per_batch = self.batch_size * self.num_unrollings
        n_batches = -(-data_size // per_batch) // divide_by_n
        if verbose > 0:
            logging.info('epoch_size: %d', -(-data_size // per_batch))
            logging.info('data_size: %d', data_size)
            logging.info('num_unrollings: %d', self.num_unrollings)
            logging.info('batch_size: %d', self.batch_size)
        t0 = time.time()
        total, count, state = None, 0, None
        self._sum_mean_loss, self._count = 0.0, 0.0

        def perplexity():
            self._sum_mean_loss = float(total) if total is not None else 0.0       # the one read-back
            self._count = float(count)
            return float(np.exp(self._sum_mean_loss / self._count)) if count else float("nan")

        def rate():
            return count * per_batch / max(time.time() - t0, 1e-9)

        for k in range(n_batches):
            ids = batch_generator.next_ids()                            # [U + 1, B]: row u + 1 is the target of row u
            loss, state = self.train_step(ids[:-1].T, ids[1:].T, state, train=is_training)
            # the reference's loss monitor is a running sum too (tf.Variable sum_mean_loss, lang/char_rnn_model.py:151-166); float64
            # here: a float32 running sum stops taking the per-batch terms in after ~1e7 of them
            total = loss.detach().to(torch.float64) if total is None else total + loss.detach()
            count += 1
            if verbose > 0 and (k + 1) % freq == 0:
                logging.info("%.1f%%, step:%d, perplexity: %.3f, speed: %.0f words", 100.0 * (k + 1) / max(n_batches * divide_by_n, 1), k,
                             perplexity(), rate())
        ppl = perplexity()
        logging.info("Perplexity: %.3f, speed: %.0f words per sec", ppl, rate())
        return ppl, None, self.global_step

    def zero_state(self, n=1):
        dev = self.store.device
        z = lambda: torch.zeros(self.hidden_size, device=dev)
        return tuple((z(), z()) for _ in range(self.num_layers))

    def step_tensors(self, ids, c_prev, h_prev):
        """One unrolling for N rows with the state as tensors (device-resident beam search): ids int64 [N] (LM ids),
        c_prev / h_prev lists over layers of [N,H].  Returns (logits [N,V_lm], c_new list, h_new list)."""
        P = self.params()
        dev = P["softmax_w"].device
        N, H = ids.shape[0], self.hidden_size
        prec = L._prec()
        with torch.no_grad():
            if self.embedding_size > 0:
                x = P["embedding"].detach()[ids]
            else:
                x = torch.nn.functional.one_hot(ids, self.vocab_size).to(torch.float32)
            cs, hs = [], []
            for l, (k, b) in enumerate(P["cells"]):
                xin = torch.cat([x, h_prev[l]], 1).contiguous()
                I = xin.shape[1]
                z = torch.empty(N, 4 * H, device=dev)
                _hip.gemm(prec, xin, k.detach(), z, False, False, N, 4 * H, I, I, 4 * H, 4 * H, bias=b.detach())
                c_new = torch.empty(N, H, device=dev)
                h_new = torch.empty(N, H, device=dev)
                _hip.check(_hip.lib().las_lstm_pointwise(_hip.p(z), _hip.p(c_prev[l]), N, H, 0.0, _hip.p(c_new), _hip.p(h_new),
                                                         _hip.stream()), "las_lstm_pointwise")
                cs.append(c_new)
                hs.append(h_new)
                x = h_new
            logits = torch.empty(N, self.vocab_size, device=dev)
            _hip.gemm(prec, x, P["softmax_w"].detach(), logits, False, False, N, self.vocab_size, H, H, self.vocab_size,
                      self.vocab_size, bias=P["softmax_b"].detach())
        return logits, cs, hs

    def fusion_plan(self, lm_weight):
        """Constants of a beam search with shallow fusion (made once per search): the first layer's input rows for the
        one-hot case (W_x[id], rounded like a bf16 GEMM operand in speed mode) and the output projection pre-scaled by
        lm_weight, so that the step can ADD lm_weight * lm_logits into the acoustic logits with the GEMM's own beta.
        Speed mode: every product of the step goes through the skinny-M kernel (las_gemm_skinny, M = beam rows), so the
        recurrent / input halves of the cell kernels and the scaled output projection are packed into bf16 MFMA fragments
        here, once (r3 decode trace: five M = 256 GEMMs at ~30 us each were 127 of a step's 270 us)."""
        P = self.params()
        H = self.hidden_size
        k0 = P["cells"][0][0].detach()
        plan = {"w": float(lm_weight)}
        if self.embedding_size == 0:
            wx = k0[:self.vocab_size]
            plan["wx"] = (wx.to(torch.bfloat16).to(torch.float32) if L._prec() == _hip.PREC_BF16 else wx).contiguous()
        plan["sw"] = (P["softmax_w"].detach() * plan["w"]).contiguous()
        plan["sb"] = (P["softmax_b"].detach() * plan["w"]).contiguous()
        if L._prec() == _hip.PREC_BF16 and H % 8 == 0 and (self.embedding_size == 0 or self.input_size % 8 == 0):
            packs = []
            for l, (k, b) in enumerate(P["cells"]):
                k = k.detach()
                I = k.shape[0] - H
                hh = _hip.skinny_pack(k, H, 4 * H, row0=I)                         # recurrent rows
                ih = None if (l == 0 and "wx" in plan) else _hip.skinny_pack(k, I, 4 * H)
                packs.append((hh, ih))
            plan["packs"] = packs
            plan["swp"] = _hip.skinny_pack(plan["sw"], H, self.vocab_size)
        return plan

    TWIN_MIN_ROWS = 384          # (las_lstm_cell_rows serves bf16 state copies from 128 rows on; at 256 rows they measured neutral)

    def twins_ok(self, plan, N):
        """True if every layer's step runs through las_lstm_cell_rows at N rows with 128-row workgroups: the cells can then read the
        recurrent state (and the layer below's output) from bf16 copies a previous launch wrote (h_out_bf16) -- half the bytes of the
        fp32 rows they would convert to bf16 anyway, bit-identical results."""
        H = self.hidden_size
        return ("packs" in plan and "wx" in plan and self.TWIN_MIN_ROWS <= N <= 1024 and H % 32 == 0 and self.embedding_size == 0
                and all(pk[1] is not None for pk in plan["packs"][1:]))

    def step_fused(self, plan, ids, c_prev, h_prev, logits, col0, id_shift=0, project=True, layer0=None, twins=None):
        """step_tensors for the device-resident beam search with half the launches: ids int32 [N] (LM id = max(ids - id_shift, 0):
        with id_shift = 2 the beam search's LAS ids are read as they are), the result is ACCUMULATED into
        logits[:, col0:col0 + V_lm] (+= lm_weight * lm_logits).  No one-hot, no concatenations: the input and recurrent halves
        of every cell product are separate GEMMs against row blocks of the TF kernel, and the one-hot input's half is a row
        look-up inside the gate kernel (las_lstm_pointwise_rows).  project=False stops after the cells (the caller runs them beside
        the acoustic model's step on another stream and calls project_fused once both are done).  layer0 = (c_new, h_new) of the first
        layer when somebody else has already computed it (first_cell_args: the Speller call launches it beside its own cell).
        Returns (c_new list, h_new list)."""
        P = self.params()
        dev = logits.device
        N, H, Vn = ids.shape[0], self.hidden_size, self.vocab_size
        prec = L._prec()
        lib = _hip.lib()
        cs, hs = [], []
        x = None
        skinny = "packs" in plan and N <= 1024
        cell_rows = skinny and H % 32 == 0 and (self.embedding_size == 0 or self.input_size % 32 == 0)
        with torch.no_grad():
            for l, (k, b) in enumerate(P["cells"]):
                if l == 0 and twins is not None:
                    twins["new"] = []
                if l == 0 and layer0 is not None:
                    cs.append(layer0[0]); hs.append(layer0[1])
                    x = layer0[1]
                    if twins is not None:
                        twins["new"].append(twins["layer0"])
                    continue
                k, b = k.detach(), b.detach()
                I = k.shape[0] - H
                rows = l == 0 and "wx" in plan
                if l == 0 and not rows:
                    lm_ids = (ids - id_shift).clamp_min_(0) if id_shift else ids
                    x = P["embedding"].detach().index_select(0, lm_ids)
                c_new = torch.empty(N, H, device=dev)
                h_new = torch.empty(N, H, device=dev)
                if cell_rows and twins is not None:
                    # bf16 operand rows: the state copy of the step before (gathered with the fp32 state) and the layer below's new copy
                    hh, ih = plan["packs"][l]
                    hb_new = torch.empty(N, H, dtype=torch.bfloat16, device=dev)
                    ca = _hip.LstmCellArgs()
                    if rows:
                        ca.x, ca.x_bf16, ca.ldx, ca.I = None, 0, 0, 0
                        ca.ids, ca.id_shift, ca.xrows = ids.data_ptr(), int(id_shift), plan["wx"].data_ptr()
                        ca.Wx = None
                    else:
                        ca.x, ca.x_bf16, ca.ldx, ca.I = twins["new"][l - 1].data_ptr(), 1, I, I
                        ca.ids, ca.id_shift, ca.xrows = None, 0, None
                        ca.Wx = ih.data_ptr()
                    ca.h, ca.h_bf16, ca.ldh, ca.Wh = twins["prev"][l].data_ptr(), 1, H, hh.data_ptr()
                    ca.bias, ca.c_prev, ca.fb = b.data_ptr(), c_prev[l].data_ptr(), 0.0
                    ca.c_out, ca.h_out, ca.gates_out, ca.h_out_bf16 = c_new.data_ptr(), h_new.data_ptr(), None, hb_new.data_ptr()
                    ca.M, ca.H, ca.fast = N, H, 0
                    _hip.check(lib.las_lstm_cell_rows_args(ctypes.byref(ca), _hip.stream()), "las_lstm_cell_rows_args")
                    twins["new"].append(hb_new)
                    cs.append(c_new)
                    hs.append(h_new)
                    x = h_new
                    continue
                if cell_rows:                                    # the whole cell in one launch (las_lstm_cell_rows)
                    hh, ih = plan["packs"][l]
                    _hip.check(lib.las_lstm_cell_rows(None if rows else _hip.p(x), 0 if rows else I, 0 if rows else I,
                                                      _hip.p(ids) if rows else None, int(id_shift), _hip.p(plan["wx"]) if rows else None,
                                                      _hip.p(h_prev[l]), H, None if rows else _hip.p(ih), _hip.p(hh), _hip.p(b), _hip.p(c_prev[l]),
                                                      N, H, 0.0, _hip.p(c_new), _hip.p(h_new), _hip.stream()), "las_lstm_cell_rows")
                    cs.append(c_new)
                    hs.append(h_new)
                    x = h_new
                    continue
                z = torch.empty(N, 4 * H, device=dev)
                # recurrent half (+ bias), then the input half on top
                if skinny:
                    _hip.skinny_gemm(h_prev[l], plan["packs"][l][0], z, N, H, 4 * H, H, 4 * H, bias=b)
                else:
                    _hip.gemm(prec, h_prev[l], k, z, False, False, N, 4 * H, H, H, 4 * H, 4 * H, bias=b, b_off=I * 4 * H)
                if not rows:
                    if skinny:
                        _hip.skinny_gemm(x, plan["packs"][l][1], z, N, I, 4 * H, I, 4 * H, accumulate=True)
                    else:
                        _hip.gemm(prec, x, k, z, False, False, N, 4 * H, I, I, 4 * H, 4 * H, beta=1.0)
                if rows:
                    _hip.check(lib.las_lstm_pointwise_rows(_hip.p(z), _hip.p(plan["wx"]), _hip.p(ids), int(id_shift), _hip.p(c_prev[l]), N, H, 0.0,
                                                           _hip.p(c_new), _hip.p(h_new), _hip.stream()), "las_lstm_pointwise_rows")
                else:
                    _hip.check(lib.las_lstm_pointwise(_hip.p(z), _hip.p(c_prev[l]), N, H, 0.0, _hip.p(c_new), _hip.p(h_new),
                                                      _hip.stream()), "las_lstm_pointwise")
                cs.append(c_new)
                hs.append(h_new)
                x = h_new
            if project:
                self.project_fused(plan, x, logits, col0)
        return cs, hs

    def first_cell_args(self, plan, ids, id_shift, c_prev, h_prev, c_out, h_out, hb_prev=None, hb_out=None):
        """las_lstm_cell_args (ctypes) of the FIRST layer's step for a one-hot LM in speed mode, or None when step_fused would not run
        that layer through las_lstm_cell_rows.  (ids, c_prev, h_prev, c_out, h_out: the tensors of every step -- fixed buffers.)"""
        P = self.params()
        H, N = self.hidden_size, ids.shape[0]
        if not ("packs" in plan and "wx" in plan and N <= 1024 and H % 32 == 0):
            return None
        a = _hip.LstmCellArgs()
        a.x, a.x_bf16, a.ldx, a.I = None, 0, 0, 0
        a.ids, a.id_shift, a.xrows = ids.data_ptr(), int(id_shift), plan["wx"].data_ptr()
        a.h, a.ldh, a.Wx, a.Wh = h_prev.data_ptr(), H, None, plan["packs"][0][0].data_ptr()
        a.bias, a.c_prev, a.fb = P["cells"][0][1].detach().data_ptr(), c_prev.data_ptr(), 0.0
        a.c_out, a.h_out, a.gates_out, a.M, a.H, a.fast = c_out.data_ptr(), h_out.data_ptr(), None, N, H, 0
        if hb_prev is not None:                                  # (twins_ok: the recurrent state from its bf16 copy, a copy of h' for the next readers)
            a.h, a.h_bf16, a.h_out_bf16 = hb_prev.data_ptr(), 1, hb_out.data_ptr()
        return a

    def cell_args(self, plan, layer, x, c_prev, h_prev, c_out, h_out):
        """las_lstm_cell_args of layer >= 1 (dense fp32 input rows x = the layer below's new h) in speed mode, or None when step_fused
        would not run it through las_lstm_cell_rows.  Fixed buffers, as for first_cell_args."""
        P = self.params()
        H, N = self.hidden_size, x.shape[0]
        if not ("packs" in plan and N <= 1024 and H % 32 == 0 and layer >= 1 and plan["packs"][layer][1] is not None):
            return None
        a = _hip.LstmCellArgs()
        a.x, a.x_bf16, a.ldx, a.I = x.data_ptr(), 0, H, H
        a.ids, a.id_shift, a.xrows = None, 0, None
        a.h, a.ldh, a.Wx, a.Wh = h_prev.data_ptr(), H, plan["packs"][layer][1].data_ptr(), plan["packs"][layer][0].data_ptr()
        a.bias, a.c_prev, a.fb = P["cells"][layer][1].detach().data_ptr(), c_prev.data_ptr(), 0.0
        a.c_out, a.h_out, a.gates_out, a.M, a.H, a.fast = c_out.data_ptr(), h_out.data_ptr(), None, N, H, 0
        return a

    def project_fused(self, plan, h_top, logits, col0):
        """logits[:, col0:col0 + V_lm] += lm_weight * (h_top . softmax_w + softmax_b)   (the weights were pre-scaled by fusion_plan)"""
        N, H, Vn = h_top.shape[0], self.hidden_size, self.vocab_size
        V_all = logits.shape[1]
        if "packs" in plan and N <= 1024:
            _hip.skinny_gemm(h_top, plan["swp"], logits, N, H, Vn, H, V_all, bias=plan["sb"], accumulate=True, c_off=col0)
        else:
            _hip.gemm(L._prec(), h_top, plan["sw"], logits, False, False, N, Vn, H, H, Vn, V_all, beta=1.0, bias=plan["sb"], c_off=col0)

    def step(self, token_ids, states):
        """token_ids [N] (LM ids), states: list over hypotheses of tuple over layers of (c,h) rows.
        Returns (logits [N,V_lm], list over hypotheses of new states)."""
        dev = self.params()["softmax_w"].device
        ids = torch.as_tensor(token_ids, device=dev).long()
        c_prev = [torch.stack([s[l][0] for s in states]).contiguous() for l in range(self.num_layers)]
        h_prev = [torch.stack([s[l][1] for s in states]).contiguous() for l in range(self.num_layers)]
        logits, cs, hs = self.step_tensors(ids, c_prev, h_prev)
        out_states = [tuple((cs[l][i], hs[l][i]) for l in range(self.num_layers)) for i in range(ids.shape[0])]
        return logits, out_states


class BatchGenerator(object):
    """The reference's batch iterator (lang/char_rnn_model.py:285-321; behaviour pinned by golden G7): `batch_size` read cursors
    start evenly spaced over the text and advance together, wrapping at the end; `next()` returns `n_unrollings + 1` id vectors
    -- the last vector of the previous call followed by `n_unrollings` new ones -- each a float64 array of `batch_size` ids.

    Here the text is encoded ONCE into an id array and a call is one fancy-indexing gather of a [n_unrollings, batch_size] block
    (the reference encodes one character at a time); `next_ids()` is the same block as one int64 array for the trainer."""

    def __init__(self, text, batch_size, n_unrollings, vocab_size, vocab_index_dict, index_vocab_dict):
        self.vocab_size = vocab_size
        self.vocab_index_dict, self.index_vocab_dict = vocab_index_dict, index_vocab_dict
        self._ids = encode_text(text, vocab_index_dict)
        self._n, self._rows = len(text), int(n_unrollings)
        self._pos = (len(text) // batch_size) * np.arange(batch_size, dtype=np.int64)      # cursor of every stream
        self._tail = self._take(1)[0]

    def _take(self, rows):
        idx = (self._pos[None, :] + np.arange(rows, dtype=np.int64)[:, None]) % self._n
        self._pos = (self._pos + rows) % self._n
        return self._ids[idx]

    def next_ids(self):
        block = np.concatenate([self._tail[None, :], self._take(self._rows)], 0)
        self._tail = block[-1]
        return block

    def next(self):
        return list(self.next_ids().astype(np.float64))


def encode_text(text, vocab_index_dict):
    """ids of a string as an int64 array; characters outside the vocabulary are logged and mapped to id 0 (the reference's
    `char2id`, lang/char_rnn_model.py:367-372)."""
    table = np.zeros(max(0x110000 if any(ord(c) > 0xFFFF for c in vocab_index_dict) else 0x10000, 1), np.int64)
    known = np.zeros(table.shape[0], bool)
    for c, i in vocab_index_dict.items():
        table[ord(c)], known[ord(c)] = i, True
    cps = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32).astype(np.int64)
    cps = np.minimum(cps, table.shape[0] - 1)            # (beyond the table: certainly not in the vocabulary)
    for cp in np.unique(cps[~known[cps]]):
        logging.info('Unexpected char %s', chr(int(cp)))
    return table[cps]


def char2id(char, vocab_index_dict):
    return int(encode_text(char, vocab_index_dict)[0])


def id2char(index, index_vocab_dict):
    return index_vocab_dict[index]


def batches2string(batches, index_vocab_dict):
    """the `batch_size` strings spelled by a list of id vectors (debug print of the reference's trainer)"""
    block = np.stack([np.asarray(b) for b in batches], 1).astype(np.int64)                 # [batch_size, len]
    return [''.join(index_vocab_dict[int(i)] for i in row) for row in block]
