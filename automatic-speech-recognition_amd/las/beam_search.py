"""las.beam_search -- BeamState / BeamSearch with the reference's API (reference las/beam_search.py)
on the MI355X engine.

What changed underneath: the reference issues 1 encode + U decode `sess.run`s per
utterance, re-uploading `np.tile(h)` and re-projecting the keys at every step.  Here the encoder output and
its key projection are computed once and stay resident; each step is one fused Speller step for all live
hypotheses (`Speller.decode`, las_speller_fwd with U=1) followed by the K10 pruning kernel
(`las_beam_step`): expansion, SOS skip, length-normalised ranking and top-`beam` selection on device.
Only the `beam` winners (parent, token, score) cross to the host, where the reference's bookkeeping
(EOS retirement, `selected` list, exhaustion path) is kept verbatim.

Scores are RAW logits summed in float32 and ranked by sum/len, as the reference does."""
import ctypes

import numpy as np
import torch

from las import _hip
from las import layers as L

NORM = True


class BeamState(object):
    """hypothesis record (reference las/beam_search.py:7-30)."""

    def __init__(self, token_ids, log_prob, att, dec_state, lm_state):
        self.token_ids = token_ids
        self.log_prob = log_prob
        self.att = att
        self.dec_state = dec_state
        self.lm_state = lm_state

    def update(self, token_id, log_prob, att, dec_state, lm_state):
        """new state extended by one token; log_prob accumulates (float32 once a float32 logit is added)."""
        return BeamState(self.token_ids + [token_id], self.log_prob + log_prob, self.att + [att], dec_state, lm_state)


class BeamSearch(object):
    """reference las/beam_search.py:32-312."""

    TOPN = 64   # las/beam_search.py:123

    def __init__(self, args, las, token_to_id, language_model):
        self.args = args
        self.listener = las.listener
        self.speller = las.speller
        self.beam_size = args.beam_size
        self.token_to_id = token_to_id
        self.start_id = token_to_id['<SOS>']
        self.end_id = token_to_id['<EOS>']
        if args.unit.lower() != "char" and args.unit.lower() != "subword":
            raise ValueError('Other units are currently not support!')
        if args.apply_lm:
            self.lm = language_model
        self._las = las

    # -- model calls (the reference's sess.run wrappers, las/beam_search.py:203-246) -------------------
    def _get_encode(self, sess, audio, audiolen):
        dev = self._las._device()
        with torch.no_grad():
            a = torch.as_tensor(np.asarray(audio), dtype=torch.float32, device=dev)
            h, _, enc_len = self.listener(a, np.asarray(audiolen), encoder=self.args.enc_type.lower(), is_training=False)
        return h, enc_len

    def _get_dec_init(self, sess):
        return self.speller.zero_state(1, self._las._device())

    def _get_decode(self, sess, enc_out, enc_len, prev_token_id, prev_align, dec_states, keys=None):
        """One step for N hypotheses.  enc_out/keys are already tiled to N rows on device."""
        with torch.no_grad():
            return self.speller.decode(enc_out, enc_len, dec_states, None, prev_align, False, keys=keys,
                                       token_ids=torch.as_tensor(prev_token_id, dtype=torch.int32))

    # -- the search ----------------------------------------------------------------------------------------
    def decode(self, sess, xs):
        """xs = (audio [1,T,feat_dim,3], audiolen [1]) -> list of BeamState, ascending (best last),
        exactly the reference's contract (las/beam_search.py:61-158; decode.py:136 reads [-1])."""
        audio, audiolen = xs
        if len(audio) != 1:
            raise ValueError('batch size must be 1 while performing beam search.')
        h, enc_len = self._get_encode(sess, audio, audiolen)
        dev = h.device
        Tp, Hd = h.shape[1], h.shape[2]
        dec_step = int(np.asarray(audiolen).reshape(-1)[0] * self.args.convert_rate)
        beam = self.beam_size
        V_ = self.args.vocab_size
        A = self.args.attention_size
        P = self.speller._params()
        keys = torch.empty(1, Tp, A, device=dev)
        _hip.gemm(L._prec(), h.contiguous(), P["Wh"].detach(), keys, False, False, Tp, A, Hd, Hd, A, A)
        enc_t = h.expand(beam, Tp, Hd).contiguous()          # resident for the whole search
        keys_t = keys.expand(beam, Tp, A).contiguous()
        enc_len_t = torch.as_tensor(enc_len).to(torch.float64).reshape(1).repeat(beam)

        init_state = self._get_dec_init(sess)
        init_rows = tuple((s[0][0], s[1][0]) if isinstance(s, tuple) else s[0] for s in init_state)
        lm_init = self.lm.zero_state(1) if self.args.apply_lm else None
        beam_set = [BeamState(token_ids=[self.start_id], log_prob=0, att=[torch.zeros(Tp, device=dev)],
                              dec_state=init_rows, lm_state=lm_init)] * beam
        selected = []
        lstm = self.speller.cell == "lstm"
        NL = self.args.num_dec_layers
        # device-side pruning buffers
        d_score = torch.zeros(1, beam, device=dev)
        d_len = torch.zeros(1, beam, dtype=torch.int32, device=dev)
        d_nlive = torch.zeros(1, dtype=torch.int32, device=dev)
        o_parent = torch.zeros(1, beam, dtype=torch.int32, device=dev)
        o_token = torch.zeros(1, beam, dtype=torch.int32, device=dev)
        o_score = torch.zeros(1, beam, device=dev)
        o_n = torch.zeros(1, dtype=torch.int32, device=dev)
        lg_buf = torch.zeros(1, beam, V_, device=dev)
        t = 0
        while t < dec_step and len(selected) < beam:
            N = len(beam_set)
            prev_ids = [b.token_ids[-1] for b in beam_set]
            prev_align = torch.stack([b.att[-1] for b in beam_set])
            if lstm:
                states = tuple((torch.stack([b.dec_state[l][0] for b in beam_set]),
                                torch.stack([b.dec_state[l][1] for b in beam_set])) for l in range(NL))
            else:
                states = tuple(torch.stack([b.dec_state[l] for b in beam_set]) for l in range(NL))
            logits, new_states, alphas = self._get_decode(sess, enc_t[:N], enc_len_t[:N], prev_ids, prev_align, states,
                                                          keys=keys_t[:N])
            lm_states = None
            if self.args.apply_lm:
                # evident intent of the (syntactically broken) branch at las/beam_search.py:109-116,131-135:
                # LM ids = LAS ids - 2, SOS (-> -1) fed as id 0; logits[:, 2:] += lm_weight * lm_logits
                lm_ids = torch.as_tensor([max(i - 2, 0) for i in prev_ids], device=dev)
                lm_out, lm_states = self.lm.step(lm_ids, [b.lm_state for b in beam_set])
                logits = logits.clone()
                logits[:, 2:] += lm_out * np.float32(self.args.lm_weight)
            lg_buf[0, :N] = logits
            d_score[0, :N] = torch.as_tensor([float(b.log_prob) for b in beam_set], device=dev)
            d_len[0, :N] = torch.as_tensor([len(b.token_ids) - 1 for b in beam_set], dtype=torch.int32, device=dev)
            d_nlive[0] = N
            _hip.check(_hip.lib().las_beam_step(_hip.p(lg_buf), _hip.p(d_score), _hip.p(d_len), _hip.p(d_nlive), 1, beam, V_,
                                                self.TOPN, t, self.start_id, _hip.p(o_parent), _hip.p(o_token),
                                                _hip.p(o_score), _hip.p(o_n), _hip.stream()), "las_beam_step")
            n = int(o_n[0])                                   # the only device->host crossing of the step
            par = o_parent[0, :n].tolist()
            tok = o_token[0, :n].tolist()
            sc = o_score[0, :n].cpu().numpy()
            lg_host = lg_buf[0].cpu().numpy()
            beam_set_new = []
            for j in range(n):
                i, v = par[j], tok[j]
                st_i = tuple((new_states[l][0][i], new_states[l][1][i]) if lstm else new_states[l][i] for l in range(NL))
                b = beam_set[i].update(v, lg_host[i, v], alphas[i], st_i, None if lm_states is None else lm_states[i])
                b.log_prob = np.float32(sc[j])               # the float32 sum computed on device (same arithmetic)
                if v == self.end_id:
                    selected.append(b)
                else:
                    beam_set_new.append(b)
            beam_set = beam_set_new
            t += 1
            if not beam_set:
                break
        if t == dec_step:
            selected.extend(beam_set)
        _hip.check_status(dev)
        return self._select_best_k(selected, NORM)

    def restore_las(self, sess, save_path, restore_epoch):
        """Restore LAS weights (reference las/beam_search.py:272-281; the TF name remapping of
        :252-270 is unnecessary: train and decode share one variable store)."""
        from las import checkpoint
        return checkpoint.restore(save_path, restore_epoch)

    def _select_best_k(self, beam_set, norm=False):
        """reference las/beam_search.py:297-312 (host-side: used for the final ranking of <= 2*beam items)."""
        if not beam_set:
            return []
        if norm:
            log_prob = [b.log_prob / (len(b.token_ids) - 1) for b in beam_set]
        else:
            log_prob = [b.log_prob for b in beam_set]
        idx = np.argsort(np.asarray(log_prob), kind="stable")[-self.beam_size:]
        return [beam_set[i] for i in idx]
