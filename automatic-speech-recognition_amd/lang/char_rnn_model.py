"""lang.char_rnn_model -- inference side of the char RNNLM (reference lang/char_rnn_model.py:11-142) used for
shallow fusion in beam search (reference las/beam_search.py:109-116,226-236; decode.py:27-39).

Only what decoding needs is built: embedding (or one-hot) -> L x BasicLSTMCell(forget_bias=0) ->
`logits = out . softmax_w + softmax_b`, one unrolling at a time for N hypotheses.  Contractions run through
las_gemm, gate math through las_lstm_pointwise (liblas_hip.so).  Training the LM (`run_epoch`, reference
lang/char_rnn_model.py:195-244, train_lm.py) is SURVEY 8(f) row F4."""
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
        if is_training:
            raise NotImplementedError("RNNLM training (reference train_lm.py) is SURVEY 8(f) row F4")
        if model != 'lstm':
            raise NotImplementedError("the shipped LM configuration is model='lstm' (reference decode.py:27-39)")
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.embedding_size, self.num_layers = embedding_size, num_layers
        self.input_size = embedding_size if embedding_size > 0 else vocab_size     # char_rnn_model.py:30-35
        self.scope = scope
        self.store = store or V.default_store()

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
