"""las.beam_search -- BeamState / BeamSearch with the reference's API (reference las/beam_search.py)
on the MI355X engine.

What changed underneath: the reference issues 1 encode + U decode `sess.run`s per
utterance, re-uploading `np.tile(h)` and re-projecting the keys at every step.  Here the encoder output and
its key projection are computed once and stay resident; each step is one fused Speller step for all live
hypotheses (`Speller.decode`, las_speller_fwd with U=1) followed by the K10 pruning kernel
(`las_beam_loop_step`): expansion, SOS skip, length-normalised ranking, top-`beam` selection, EOS retirement,
termination and the gather of the survivors' recurrent state -- all on the device, for several utterances at once.
The host replays the launches and rebuilds the reference's `BeamState` objects from back-pointer records after the
last step.

Scores are RAW logits summed in float32 and ranked by sum/len, as the reference does."""
import collections.abc
import ctypes

import os

import numpy as np
import torch

from las import _hip
from las import layers as L

NORM = True


class BeamState(object):
    """hypothesis record (reference las/beam_search.py:7-30)."""

    def __init__(self, token_ids, log_prob, att, dec_state, lm_state):
        self._token_ids = token_ids             # a list, or (round 5, decode_batch's results) a (start id, numpy row) pair turned into the list on first use
        self.log_prob = log_prob
        self.att = att
        self.dec_state = dec_state
        self.lm_state = lm_state

    @property
    def token_ids(self):
        if isinstance(self._token_ids, tuple):
            self._token_ids = [self._token_ids[0]] + self._token_ids[1].tolist()
        return self._token_ids

    @token_ids.setter
    def token_ids(self, v):
        self._token_ids = v

    def update(self, token_id, log_prob, att, dec_state, lm_state):
        """new state extended by one token; log_prob accumulates (float32 once a float32 logit is added)."""
        return BeamState(self.token_ids + [token_id], self.log_prob + log_prob, self.att + [att], dec_state, lm_state)


class _AttRows(collections.abc.Sequence):
    """`BeamState.att` of a hypothesis returned by decode_batch: item i is the alignment that produced token i (item 0 = zeros,
    las/beam_search.py:88), all items views of ONE [len, T'] tensor gathered from the device-side history in a single
    indexing operation -- building a Python list of 200 slices per hypothesis was a third of the decode time."""

    def __init__(self, rows, lo=None, n=None, width=None):
        # (round 5) lazily sliced: `rows` may be the gathered tensor of ALL hypotheses of a batch, this hypothesis being rows
        # [lo, lo + n) x [0, width) -- a torch slice per hypothesis was most of the 4 ms the host spent after a 64-utterance search
        self._all, self._lo, self._n, self._w = rows, lo, n, width
        self._cut = rows if lo is None else None

    @property
    def _rows(self):
        if self._cut is None:
            self._cut = self._all[self._lo:self._lo + self._n, :self._w]
        return self._cut

    def __len__(self):
        return self._rows.shape[0] if self._lo is None else self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self._rows[i].unbind(0))
        return self._rows[i]

    def __add__(self, other):
        return list(self) + list(other)


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
        self.use_graph = os.environ.get("LAS_NO_DECODE_GRAPH") != "1"     # decode_batch replays one captured step
        self.fuse_projection = os.environ.get("LAS_NO_DECODE_FUSED_PROJ") != "1"   # decode_batch: cell in one launch, projection inside the beam kernel
        # decode_batch (round 5): rows + LM 1 | Speller cell + LM 2 | pruning + gather -- three dependent launches instead of five.  OFF by
        # default: measured in the replayed graph (profiles/r5_decode_three_launches.txt) the three launches take 73-80 us of kernel time per
        # step against 59 us for the five (14.3 + 12.7 + 11.2 + 15.9 + 4.8): every kernel of the step fills the machine by itself (256 rows = 256
        # workgroups of 16 waves), so rows + LM 1 run one after the other inside their launch (22-26 us), and 16 pruning workgroups gather
        # 207 KB each more slowly (30-34 us) than 256 x ntens gather workgroups do (4.8 us); dependent launches cost ~0 us in a replayed graph
        self.three_launches = os.environ.get("LAS_DECODE_3_LAUNCHES") == "1"
        # decode_batch (round 5): the attention rows of FOUR hypotheses of an utterance in one workgroup (LAS_SPELLER_ROWS_SHARE4: Ws, keys and
        # encoder rows read once for the four) from this many hypothesis rows on (0 = never); bit-identical to one row per workgroup
        self.share_rows_from = int(os.environ.get("LAS_DECODE_SHARE_ROWS_FROM", "512"))
        self.xcd_local_rows = os.environ.get("LAS_NO_XCD_LOCAL_ROWS") != "1"       # decode_batch (round 6): an utterance's hypothesis rows on one XCD
        self.shared_operands = os.environ.get("LAS_NO_SHARED_OPERANDS") != "1"    # ... and ONE copy of its keys / encoder rows for all of them (not np.tile's 16)
        self.lm_state_copies = os.environ.get("LAS_NO_LM_STATE_COPIES") != "1"      # decode_batch (round 5): bf16 copies of the LM's state from 384 rows on
        self.steps_per_graph = int(os.environ.get("LAS_DECODE_STEPS_PER_GRAPH", "8"))   # search steps per captured HIP graph (one replay = that many steps)
        self.ragged_encoder = os.environ.get("LAS_NO_RAGGED_ENCODER") != "1"        # decode_batch: one encoder pass over rows of different lengths
        self.parallel_encoders = os.environ.get("LAS_NO_PARALLEL_ENCODERS") != "1"   # decode_batch: encoders of different lengths on several streams
        self._enc_streams = None
        self.measure = os.environ.get("LAS_DECODE_TIMING") == "1"         # decode_batch leaves its phase / per-part timing in last_timing
        self.last_timing = None
        self._capture_stream = None
        self._tail_stream = None            # decode_batches: a batch's back-tracking and read-back run here, beside the next batch's search

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
        return self.decode_batch(sess, [xs])[0]

    def decode_batches(self, sess, batches, sync_every=32):
        """decode_batch over a SEQUENCE of batches (decode.py's loop over a test set), as a generator of its results -- with the encoders
        of batch k+1 on a second stream UNDER the search of batch k.  An encoder pass is four latency-bound sweeps on 8-72 of the 256 CUs and
        a search step is three small launches, so the two fit side by side; one after the other the encoders are a fifth of a 64-utterance
        batch (7 of 35 ms).  The results are those of decode_batch, batch by batch (the same kernels on the same inputs; only the
        stream differs)."""
        dev = self._las._device()
        main = torch.cuda.current_stream(dev)
        # the device's `comm` auxiliary stream (idle outside training) -- it owns a hardware queue (_hip.aux_streams); a fresh stream would
        # share one of the four queues with the launch stream or an auxiliary stream and its kernels would run in line with theirs
        es = _hip.aux_streams(dev)["comm"]

        first_launch = [True]

        def launch(xs_list):
            if first_launch[0]:              # (everything queued so far: a caller's weight update.  Later launches must NOT wait for the launch
                es.wait_stream(main)         #  stream -- they are issued while the current batch's search steps are queued on it)
                first_launch[0] = False
            with torch.cuda.stream(es):
                pre = self._run_encoders(sess, xs_list)
            ev = torch.cuda.Event()
            ev.record(es)
            seen = set()
            for h in pre[0]:                 # allocated on `es`, read by the search on `main`
                if h.untyped_storage().data_ptr() not in seen:
                    seen.add(h.untyped_storage().data_ptr())
                    h.record_stream(main)
            return xs_list, pre, ev

        it = iter(batches)
        first = next(it, None)
        cur = launch(first) if first is not None else None
        pending = None                       # the previous batch's back-tracking / read-back / host objects, not yet run
        try:
            yield from self._decode_batches_loop(sess, it, cur, launch, main, sync_every)
        finally:
            # a consumer that stops early, or a batch that raises, must not leave the next batch's encoders running into tensors
            # nobody holds: the launch stream waits for everything that was put on the encoder stream
            main.wait_stream(es)

    def _decode_batches_loop(self, sess, it, cur, launch, main, sync_every):
        pending = None
        while cur is not None:
            box, out = [], []

            def between(pending=pending):
                # runs inside decode_batch, behind the launches of its first `sync_every` search steps, i.e. while the device is searching:
                # the host's part of the NEXT batch (stacking the utterances, the copy to the device, the encoder launches on `es`) and
                # everything behind the PREVIOUS batch's search (on the tail stream)
                nxt_xs = next(it, None)
                box.append(launch(nxt_xs) if nxt_xs is not None else None)
                if pending is not None:
                    out.append(pending())
            main.wait_event(cur[2])
            pending = self.decode_batch(sess, cur[0], sync_every, _pre=cur[1], _after_launch=between, _defer=True)
            if out:
                yield out[0]
            if not callable(pending):        # (LAS_DECODE_TIMING: decode_batch measured its phases and returned the results)
                yield pending
                pending = None
            cur = box[0]
        if pending is not None:
            yield pending()

    def _run_encoders(self, sess, xs_list):
        """The encoders of a batch of utterances on the CURRENT stream, without waiting for the device (the encoded lengths are host
        values): -> (encs [per utterance: [1, T'_u, Hd] views], enc_lens, dec_steps, h_one).  decode_batch's first phase; decode_batches
        runs it for the NEXT batch on a second stream under the search of the current one."""
        a = self.args
        dev = self._las._device()
        n = len(xs_list)
        sp = self.speller
        prec = L._prec()
        encs, enc_lens, dec_steps = [None] * n, [None] * n, []
        groups = {}
        for u, (audio, audiolen) in enumerate(xs_list):
            if len(audio) != 1:
                raise ValueError('every entry of xs_list is one utterance: audio [1,T,feat_dim,3]')
            al = np.asarray(audiolen).reshape(-1)
            groups.setdefault((tuple(np.shape(audio)[1:]), float(al[0])), []).append(u)
            dec_steps.append(int(al[0] * a.convert_rate))                                       # las/beam_search.py:78
        def encode_group(us):
            # utterances of the SAME shape and length share one encoder launch: every row of the encoder is computed
            # independently of the other rows, so this is exactly the one-at-a-time result (unlike padding, see above)
            audio = np.concatenate([np.asarray(xs_list[u][0]) for u in us], 0)
            audiolen = np.concatenate([np.asarray(xs_list[u][1]).reshape(-1)[:1] for u in us], 0)
            h, enc_len = self._get_encode(sess, audio, audiolen)
            el = torch.as_tensor(enc_len).reshape(-1).cpu().tolist()       # (one read-back, not one per utterance)
            for i, u in enumerate(us):
                encs[u] = h[i:i + 1]
                enc_lens[u] = float(el[i])
            return h

        glist = list(groups.values())
        h_one = None                         # the encoder output of all utterances when they form one group
        cellid = L._cellid(sp.cell)
        H_enc = a.enc_units
        ragged = (len(glist) > 1 and self.ragged_encoder and prec == _hip.PREC_BF16 and a.enc_type.lower() == "pblstm" and
                  len({np.shape(x[0])[2:] for x in xs_list}) == 1 and
                  _hip.rnn_seq_io_dtype(cellid, prec, H_enc) == torch.bfloat16 and _hip.rnn_seq_fwd_rows_ok(cellid, prec, n, H_enc))
        if ragged:
            # ONE encoder pass over all utterances although their lengths differ (the reference feeds them one at a time, unpadded):
            # the forward sweeps take the rows' frame counts (las_rnn_seq_fwd_rows: a row's state and outputs are zero behind its last
            # frame, so its backward direction starts from the zero state at ITS last frame and an odd length pairs with a zero frame,
            # exactly as alone); the products in between are row-wise.  Every real frame equals the one-at-a-time result.
            lens = [int(np.shape(x[0])[1]) for x in xs_list]
            Tmax = max(lens)
            audio = np.zeros((n, Tmax) + tuple(np.shape(xs_list[0][0])[2:]), np.float32)
            for u, x in enumerate(xs_list):
                audio[u, :lens[u]] = np.asarray(x[0])[0]
            audiolen = np.concatenate([np.asarray(x[1]).reshape(-1)[:1] for x in xs_list], 0)
            L.ROW_T[0] = torch.tensor(lens, dtype=torch.int32, device=dev)
            try:
                h, enc_len = self._get_encode(sess, audio, audiolen)
            finally:
                L.ROW_T[0] = None
            el = torch.as_tensor(enc_len).reshape(-1).cpu().tolist()
            for u in range(n):
                tp = lens[u]
                for _ in range(a.num_enc_layers):
                    tp = (tp + 1) // 2
                encs[u] = h[u:u + 1, :tp]
                enc_lens[u] = float(el[u])
            glist = []
        else:
            h_one = encode_group(glist[0])
            if len(glist) > 1:
                h_one = None
        if ragged:
            pass
        elif len(glist) > 1 and self.parallel_encoders:
            # Groups of DIFFERENT lengths (a real test set: every utterance its own) run side by side on a few streams: an encoder is
            # four latency-bound sweeps that occupy 8-16 of the 256 CUs each, one after the other they are the bulk of a search
            # (16 x 4 ms against 15 ms of search).  The first group ran on this stream (it also builds the weights' bf16 shadows once);
            # the others start behind it.  The chunked x-projection hand-over shares one side stream and one ring of flag words:
            # it is off for these launches (whole products in front of the sweeps -- they overlap across the streams anyway).
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record()
            if self._enc_streams is None:
                self._enc_streams = [torch.cuda.Stream() for _ in range(8)]
            saved, L.XPROJ_CHUNK_STEPS = L.XPROJ_CHUNK_STEPS, 0
            try:
                for i, us in enumerate(glist[1:]):
                    s_ = self._enc_streams[i % len(self._enc_streams)]
                    s_.wait_event(ready)
                    with torch.cuda.stream(s_):
                        encode_group(us).record_stream(cur)
            finally:
                L.XPROJ_CHUNK_STEPS = saved
            for s_ in self._enc_streams:
                cur.wait_stream(s_)
        else:
            for us in glist[1:]:
                encode_group(us)
        return encs, enc_lens, dec_steps, h_one

    def decode_batch(self, sess, xs_list, sync_every=32, _pre=None, _after_launch=None, _defer=False):
        """Beam search for several utterances at once (what decode.py's loop over utterances, decode.py:131-149, becomes on
        one GPU): xs_list = [(audio [1,T_u,feat_dim,3], audiolen [1]), ...] -> [list of BeamState (ascending), ...].

        Every utterance is ENCODED on its own, at its own length (the reference's encoder has no sequence mask, so its
        output depends on the padded length -- SURVEY fact 4 -- and decode.py feeds unpadded utterances); the search then
        runs all utterances x beam hypotheses as ONE batch of rows per step: fused Speller step (las_speller_fwd, U = 1)
        [+ LM step + shallow fusion] + las_beam_loop_step (pruning, EOS retirement, termination, state gather -- all on
        the device).  The host replays the launches without waiting and reads the back-pointer records once at the end
        (plus one `done` poll every `sync_every` steps to stop early)."""
        import ctypes
        from las.las import _alloc_bufs, _fill_fwd_args
        a = self.args
        dev = self._las._device()
        n, beam, V_, A, NL, D = len(xs_list), self.beam_size, a.vocab_size, a.attention_size, a.num_dec_layers, a.dec_units
        sp = self.speller
        lstm = sp.cell == "lstm"
        prec = L._prec()
        P = sp._params()
        # ---- encoders (one per utterance) and the hoisted key projection
        import time
        tm = {}
        def mark(name):                                  # wall-clock marks (with a device sync) only when asked for
            if self.measure:
                torch.cuda.synchronize(dev)
                tm[name] = time.perf_counter()
        mark("start")
        if _pre is None:
            _pre = self._run_encoders(sess, xs_list)
        encs, enc_lens, dec_steps, h_one = _pre
        Tps = [h.shape[1] for h in encs]
        Tp, Hd = max(Tps), encs[0].shape[2]
        N = n * beam
        # Round 6: ONE operand block per utterance, resident for the whole search (frames past T'_u are masked).  The reference feeds
        # np.tile(h) -- a copy of the encoder output per hypothesis (las/beam_search.py:216) -- and so did rounds 1-5: 16 copies of every
        # utterance's keys and encoder rows, each read through its own addresses at every step (54 MB per step at 256 rows, r5_decode_pmc.json).
        # The search step's row kernels now take the utterance's block (LAS_SPELLER_SHARED_OPERANDS); the tiled copies are only made for
        # the long form of the step (kernel families that index per row).
        Wh = P["Wh"].detach()
        if h_one is not None and h_one.shape[0] == n:
            # every utterance in ONE encoder group (equal lengths): the hoisted key projection as one product
            enc_u = h_one.contiguous()
            keys_u = torch.empty(n, Tp, A, device=dev)
            _hip.gemm(prec, enc_u, Wh, keys_u, False, False, n * Tp, A, Hd, Hd, A, A)
        else:
            enc_u = torch.zeros(n, Tp, Hd, device=dev)
            keys_u = torch.zeros(n, Tp, A, device=dev)
            for u, h in enumerate(encs):
                k = torch.empty(1, Tps[u], A, device=dev)
                _hip.gemm(prec, h.contiguous(), Wh, k, False, False, Tps[u], A, Hd, Hd, A, A)
                enc_u[u, :Tps[u]] = h[0]
                keys_u[u, :Tps[u]] = k[0]
        tiled_ops = []

        def tiled():
            """the per-row copies [N, Tp, .] (made once, on demand)"""
            if not tiled_ops:
                enc_t = torch.empty(N, Tp, Hd, device=dev)
                keys_t = torch.empty(N, Tp, A, device=dev)
                enc_t.view(n, beam, Tp, Hd).copy_(enc_u.unsqueeze(1).expand(n, beam, Tp, Hd))
                keys_t.view(n, beam, Tp, A).copy_(keys_u.unsqueeze(1).expand(n, beam, Tp, A))
                tiled_ops.extend([enc_t, keys_t])
            return tiled_ops
        enc_len_i32 = torch.tensor(np.repeat(np.asarray(enc_lens, np.float64), beam)).to(torch.int32).to(dev)
        Umax = max(max(dec_steps), 1)
        # ---- device-resident loop state
        i32 = dict(dtype=torch.int32, device=dev)
        selcap = 3 * beam
        score = torch.zeros(n, beam, device=dev)
        length = torch.zeros(n, beam, **i32)
        nlive = torch.full((n,), beam, **i32)
        nsel = torch.zeros(n, **i32)
        done = torch.zeros(n, **i32)
        dstep = torch.tensor(dec_steps, **i32)
        step = torch.zeros(2, **i32)                       # [0] the device step counter, [1] arrival counter of the folded gather (fold_gather)
        hist_parent = torch.zeros(Umax, n, beam, **i32)
        hist_token = torch.zeros(Umax, n, beam, **i32)
        hist_slot = torch.zeros(Umax, n, beam, **i32)
        hist_score = torch.zeros(Umax, n, beam, device=dev)
        hist_n = torch.zeros(Umax, n, **i32)
        sel_t = torch.zeros(n, selcap, **i32)
        sel_j = torch.zeros(n, selcap, **i32)
        src_row = torch.zeros(n, beam, **i32)
        next_token = torch.full((N,), self.start_id, **i32)
        alphas_hist = torch.zeros(Umax + 1, N, Tp, device=dev)      # [Umax] stays zero: item 0 of every hypothesis' att
        align_prev = torch.zeros(N, Tp, device=dev)
        # ---- the fused Speller step for all rows: slot 0 of hs / cs = state entering the step, slot 1 = state leaving it
        dims = sp._dims(N, Tp, 1)
        bufs = _alloc_bufs(dims, dev)
        bufs["hs"].zero_()
        if lstm:
            bufs["cs"].zero_()
        tokens_out = torch.zeros(1, N, **i32)
        Pd = {k: (v.detach() if torch.is_tensor(v) else [t.detach() for t in v]) for k, v in P.items()}
        fa = _hip.SpellerFwdArgs()
        keep = _fill_fwd_args(fa, dims, Pd, enc_u, keys_u, enc_len_i32, next_token, tokens_out, bufs, True, 0, keep_state0=True,
                              align0=align_prev)
        lib = _hip.lib()
        nbytes = lib.las_speller_workspace_bytes(N, Tp, Hd, A, D, NL, a.embedding_size, V_, 1, dims["cell"])
        ws = _hip.workspace(dev, nbytes, "speller")
        fa.ws, fa.ws_bytes = ws.data_ptr(), ws.numel()
        logits = bufs["logits"][0]                                                      # [N, V]
        # ---- state tensors that follow their hypotheses (gathered by las_beam_loop_step)
        st_in, st_out = [], []
        for l in range(NL):
            st_in.append(bufs["hs"][l, 1]); st_out.append(bufs["hs"][l, 0])
            if lstm:
                st_in.append(bufs["cs"][l, 1]); st_out.append(bufs["cs"][l, 0])
        k_align = len(st_in)
        st_in.append(alphas_hist[0]); st_out.append(align_prev)
        lm = self.lm if a.apply_lm else None
        lm_w = np.float32(a.lm_weight) if lm is not None else None
        lm_plan = lm.fusion_plan(lm_w) if lm is not None else None
        lm_hb = None
        if lm is not None:
            Hl, NLl = lm.hidden_size, lm.num_layers
            lm_c = [torch.zeros(N, Hl, device=dev) for _ in range(NLl)]
            lm_h = [torch.zeros(N, Hl, device=dev) for _ in range(NLl)]
            k_lm = len(st_in)
            # bf16 copies of the LM's recurrent state (round 5): from 384 rows on the cells read them -- and the layer below's copy -- instead
            # of converting the fp32 rows while staging them (half the bytes through a CU that is bound by its ingest); the copies follow
            # their hypotheses through the gather as rows of H / 2 floats -- INSTEAD of the fp32 h rows, which nobody reads then (the
            # projection takes the step's fresh h).  Bit-identical: the copy is the staging's own rounding.
            if self.lm_state_copies and prec == _hip.PREC_BF16 and not self.three_launches and lm.twins_ok(lm_plan, N):
                lm_hb = [torch.zeros(N, Hl, dtype=torch.bfloat16, device=dev) for _ in range(NLl)]
                for l in range(NLl):
                    st_in.append(lm_c[l]); st_out.append(lm_c[l])                  # inputs are re-pointed every step
                k_tw = len(st_in)
                for l in range(NLl):
                    st_in.append(lm_hb[l].view(torch.float32)); st_out.append(lm_hb[l].view(torch.float32))
            else:
                for l in range(NLl):
                    st_in += [lm_c[l], lm_h[l]]; st_out += [lm_c[l], lm_h[l]]
        ba = _hip.BeamLoopArgs()
        for name, t in (("logits", logits), ("score", score), ("length", length), ("nlive", nlive), ("nsel", nsel), ("done", done),
                        ("dec_step", dstep), ("step", step), ("hist_parent", hist_parent), ("hist_token", hist_token),
                        ("hist_slot", hist_slot), ("hist_score", hist_score), ("hist_n", hist_n), ("sel_t", sel_t), ("sel_j", sel_j),
                        ("src_row", src_row), ("next_token", next_token)):
            setattr(ba, name, t.data_ptr())
        ba.nutt, ba.beam, ba.V, ba.Umax, ba.selcap, ba.topn = n, beam, V_, Umax, selcap, self.TOPN
        ba.start_id, ba.end_id, ba.ntens = self.start_id, self.end_id, len(st_in)
        for k, (ti, to) in enumerate(zip(st_in, st_out)):
            ba.state_in[k], ba.state_out[k], ba.state_width[k] = ti.data_ptr(), to.data_ptr(), ti.shape[-1]
        mark("encoded")
        # the step's alignments land in a fixed buffer and are filed under the DEVICE step counter, so that one step is the
        # same sequence of launches with the same arguments every time: it is captured into a HIP graph after the first
        # (eager) step and replayed -- the loop is bound by the host's launch rate otherwise (5-8 launches and their Python glue per step)
        alphas_cur = torch.zeros(N, Tp, device=dev)
        fa.alphas = alphas_cur.data_ptr()
        # rows u beam .. (u + 1) beam - 1 are the hypotheses of utterance u (same enc / keys): the row launches keep them on one XCD
        fa.row_group = beam if self.xcd_local_rows else 0
        ba.state_in[k_align] = alphas_cur.data_ptr()
        ba.file_in, ba.file_out, ba.file_width = alphas_cur.data_ptr(), alphas_hist.data_ptr(), Tp
        held = []

        mode = {"fused": False}

        def speller_part():
            rc = lib.las_speller_fwd(ctypes.byref(fa), _hip.stream())
            if rc < 0 and mode["fused"] and not (fa.flags & _hip.SPELLER_REUSE_PREP):
                # the library refuses the short form for this geometry (it needs its prefetching row kernels): the long form it is
                mode["fused"] = False
                mode.pop("three", None)
                fa.flags &= ~(_hip.SPELLER_NO_LOGITS | _hip.SPELLER_ROWS_SHARE4 | _hip.SPELLER_SHARED_OPERANDS)
                fa.enc, fa.keys = (t.data_ptr() for t in tiled())       # (the long form's kernel families index enc / keys per row)
                fa.companion = None
                fa.companion_rows = None
                ba.proj_w = None
                ba.fold_gather = 0
                rc = lib.las_speller_fwd(ctypes.byref(fa), _hip.stream())
            _hip.check(rc, "las_speller_fwd")
            fa.flags |= _hip.SPELLER_REUSE_PREP      # enc / keys / weights are fixed for the search: their bf16 copies are made once

        # ---- the short form of a step (speed mode, one LSTM layer): the Speller call stops after its cell (LAS_SPELLER_NO_LOGITS: attention
        # rows + ONE cell launch) and the vocabulary projection -- the Speller's output layer and, concatenated along K, the LM's softmax
        # layer scaled by lm_weight and shifted to its token columns -- happens inside las_beam_loop_step: 5 launches per step instead of 8
        E_ = a.embedding_size
        fused_proj = (prec == _hip.PREC_BF16 and lstm and NL == 1 and self.fuse_projection and D % 32 == 0 and (E_ + Hd + D) % 32 == 0 and
                      ((beam + 15) // 16) * ((V_ + 15) // 16) <= 8 and (lm is None or (lm.hidden_size % 32 == 0 and "packs" in lm_plan)))
        proj_keep = None
        mode["fused"] = fused_proj
        shared = fused_proj and self.shared_operands and self.xcd_local_rows
        if not shared:
            fa.enc, fa.keys = (t.data_ptr() for t in tiled())          # np.tile(enc, beam) of the reference (las/beam_search.py:60-63), physically
        if fused_proj:
            Wcat, bcat = Pd["Wv"].contiguous(), Pd["bv"].clone()
            if lm is not None:
                Wl = torch.zeros(lm.hidden_size, V_, device=dev)
                Wl[:, 2:2 + lm.vocab_size] = lm_plan["sw"]
                bcat[2:2 + lm.vocab_size] += lm_plan["sb"]
                Wcat = torch.cat([Wcat, Wl], 0)
            Wcat = Wcat.contiguous()
            proj_keep = (_hip.skinny_pack(Wcat, Wcat.shape[0], V_), bcat.contiguous())
            fa.flags |= _hip.SPELLER_NO_LOGITS
            if shared:
                fa.flags |= _hip.SPELLER_SHARED_OPERANDS     # enc / keys: one block per utterance (fa.row_group = beam rows share it)
            if self.share_rows_from and N >= self.share_rows_from and beam % 4 == 0 and a.mode == "add":
                fa.flags |= _hip.SPELLER_ROWS_SHARE4         # rows 4g .. 4g+3 are hypotheses of ONE utterance: same enc / keys / length
            ba.proj_w, ba.proj_b = proj_keep[0].data_ptr(), proj_keep[1].data_ptr()
            ba.proj_h0, ba.proj_k0 = bufs["hs"][NL - 1, 1].data_ptr(), D
            ba.proj_h1, ba.proj_k1 = None, (lm.hidden_size if lm is not None else 0)
            if lm is not None:
                lm0 = (torch.empty(N, lm.hidden_size, device=dev), torch.empty(N, lm.hidden_size, device=dev))
                lm0_hb = torch.empty(N, lm.hidden_size, dtype=torch.bfloat16, device=dev) if lm_hb is not None else None
                lm0_args = lm.first_cell_args(lm_plan, next_token, 2, lm_c[0], lm_h[0], lm0[0], lm0[1],
                                              hb_prev=lm_hb[0] if lm_hb is not None else None, hb_out=lm0_hb)
                lm1 = lm1_args = None
                if self.three_launches and lm0_args is not None and NLl == 2:
                    lm1 = (torch.empty(N, lm.hidden_size, device=dev), torch.empty(N, lm.hidden_size, device=dev))
                    lm1_args = lm.cell_args(lm_plan, 1, lm0[1], lm_c[1], lm_h[1], lm1[0], lm1[1])
                if lm1_args is not None:
                    # Round 5: a search step in THREE dependent launches (las/beam_search.py:94-158 is one loop).  The LM's first layer depends
                    # on the step's tokens only: it runs as extra workgroups of the attention-row launch; its second layer rides with the
                    # Speller's cell (two problems of one grid); the pruning launch gathers the survivors' state rows itself.  All state in
                    # fixed buffers: the captured step is the same three launches every time.
                    fa.companion_rows = ctypes.pointer(lm0_args)
                    fa.companion = ctypes.pointer(lm1_args)
                    mode["three"] = (lm0, lm1)
                    for l, (cn, hn) in enumerate((lm0, lm1)):
                        ba.state_in[k_lm + 2 * l], ba.state_in[k_lm + 2 * l + 1] = cn.data_ptr(), hn.data_ptr()
                    ba.proj_h1 = lm1[1].data_ptr()
                    ba.fold_gather = 1
                elif lm0_args is not None:
                    # the LM's first layer depends on the tokens only: it rides with the Speller's cell as the second problem of one grid
                    fa.companion = ctypes.pointer(lm0_args)
                    mode["lm0"] = lm0
                    mode["lm0_hb"] = lm0_hb

        def lm_cells():
            # evident intent of the (syntactically broken) branch at las/beam_search.py:109-116,131-135:
            # LM ids = LAS ids - 2, SOS (-> -1) fed as id 0; logits[:, 2:] += lm_weight * lm_logits
            layer0 = mode.get("lm0") if mode["fused"] else None
            tw = None
            if lm_hb is not None:
                tw = {"prev": lm_hb, "layer0": mode.get("lm0_hb") if layer0 is not None else None}
            cs_new, hs_new = lm.step_fused(lm_plan, next_token, lm_c, lm_h, logits, 2, id_shift=2, project=False, layer0=layer0, twins=tw)
            for l in range(NLl):
                if tw is not None:
                    ba.state_in[k_lm + l], ba.state_in[k_tw + l] = cs_new[l].data_ptr(), tw["new"][l].data_ptr()
                else:
                    ba.state_in[k_lm + 2 * l], ba.state_in[k_lm + 2 * l + 1] = cs_new[l].data_ptr(), hs_new[l].data_ptr()
            if tw is not None:
                cs_new = cs_new + tw["new"]                                       # (kept alive with the rest)
            if mode["fused"]:
                ba.proj_h1 = hs_new[-1].data_ptr()
            held[:] = [cs_new, hs_new]                                            # alive until the gather has been enqueued

        def lm_part():
            if mode["fused"] and mode.get("three"):
                return                               # both LM layers ran inside the Speller call's two launches
            lm_cells()
            if not mode["fused"]:
                lm.project_fused(lm_plan, held[1][-1], logits, 2)

        def beam_part():                                 # files alphas_cur under the device step counter, prunes, gathers, advances the counter
            _hip.check(lib.las_beam_loop_step(ctypes.byref(ba), _hip.stream()), "las_beam_loop_step")

        def one_step():
            # (the LM's cells depend only on the tokens, but running them as a parallel branch on a second stream beside the Speller
            # step LOSES on this device: 111 against 99.5 us per captured step -- the cross-queue joins cost more than the overlap gives)
            speller_part()
            if lm is not None:
                lm_part()
            beam_part()

        steps_run = 0
        graph = None
        use_graph = self.use_graph and Umax > 2
        # One hipGraphLaunch costs the host more than a 3-launch step costs the device (r5 probe: 75-85 us per replayed step whatever the
        # step's kernels added up to -- 55-62 us): the captured graph holds K consecutive steps.  Every launch of a step reads the DEVICE
        # step counter, and a step at t >= Umax does nothing that is kept (las_beam_loop_step returns before it files or records), so
        # the last replay may run past the bound.
        K = max(1, int(self.steps_per_graph))
        with torch.no_grad():
            t, next_check = 0, sync_every * (2 if _after_launch is not None else 1)     # (the host work done at the first check wants ~6 ms of queued steps)
            while t < Umax:
                if graph is not None:
                    graph.replay()
                    t += K
                elif use_graph and t == 1:
                    try:
                        # (capture_begin / capture_end by hand: the torch.cuda.graph context synchronises the device and EMPTIES the
                        #  caching allocator on entry -- a millisecond per search, and every buffer of the next search a fresh hipMalloc)
                        g = torch.cuda.CUDAGraph()
                        if self._capture_stream is None:
                            self._capture_stream = torch.cuda.Stream()
                        cs_ = self._capture_stream
                        cs_.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(cs_):
                            g.capture_begin()
                            try:
                                for _ in range(K):
                                    one_step()
                            finally:
                                g.capture_end()
                        torch.cuda.current_stream().wait_stream(cs_)
                        graph = g                      # (capturing does not execute: the captured steps run as the replay)
                        graph.replay()
                        t += K
                    except Exception as e:             # capture refused (e.g. an op that allocates host memory): stay eager
                        use_graph = False
                        torch.cuda.synchronize(dev)
                        if self.args.verbose > 0:
                            print("decode_batch: graph capture failed, running eagerly: %s" % e)
                        one_step()
                        t += 1
                else:
                    one_step()
                    t += 1
                steps_run = min(t, Umax)
                if t >= next_check:                                                     # the only host wait inside the loop
                    next_check += sync_every
                    if _after_launch is not None:                                       # (decode_batches: the next batch's host work and encoder
                        _after_launch(); _after_launch = None                          #  launches, while the device runs the steps enqueued so far)
                    if bool(done.all()):
                        break
        if _after_launch is not None:
            _after_launch()
        del keep
        mark("searched")
        if _defer and not self.measure:
            # decode_batches: everything behind the search -- back-tracking, the alignment gather, the read-back, the host objects -- as a
            # closure that runs on a side stream behind THIS point of the launch stream: the caller launches the next batch's search first and
            # calls it while the device is busy with that (the read-back waits for the side stream only).  This batch's tensors live in the
            # closure until then.
            ev = torch.cuda.Event()
            ev.record()
            if self._tail_stream is None:
                self._tail_stream = torch.cuda.Stream()

            alive = (hist_parent, hist_token, hist_slot, hist_score, hist_n, sel_t, sel_j, nsel, score, length, nlive, done, dstep, step,
                     src_row, next_token)                 # `ba` holds their addresses

            def finish(alive=alive):
                ts_ = self._tail_stream
                ts_.wait_event(ev)
                with torch.cuda.stream(ts_):
                    return self._decode_tail(dev, n, selcap, Umax, i32, ba, lib, alphas_hist, Tps, mark)
            return finish
        results = self._decode_tail(dev, n, selcap, Umax, i32, ba, lib, alphas_hist, Tps, mark)
        mark("done")
        parts = {}
        if tm:        # device time of the three parts of a decode step (HIP events, 50 eager repetitions each, after the search)
            step.zero_()
            for name, fn in (("speller", speller_part), ("lm", lm_part if lm is not None else None), ("beam", beam_part)):
                if fn is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.no_grad():
                    fn(); e0.record()
                    for _ in range(50):
                        fn()
                        if name == "beam":
                            step.zero_()
                    e1.record()
                torch.cuda.synchronize(dev)
                parts[name] = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
            ks = list(tm)
            self.last_timing = {ks[i + 1]: round(tm[ks[i + 1]] - tm[ks[i]], 4) for i in range(len(ks) - 1)}
            self.last_timing.update(steps=steps_run, rows=N, frames=Tp, graph=graph is not None, parts_us=parts)
            if self.args.verbose > 0:
                print("decode_batch timing (s):", self.last_timing)
        return results

    def _decode_tail(self, dev, n, selcap, Umax, i32, ba, lib, alphas_hist, Tps, mark):
        """What follows a search (decode_batch; on the current stream): the back pointers are walked on the device (las_beam_backtrack), one
        read-back, then the reference's host-side objects (las/beam_search.py:136-158, :297-312)."""
        import ctypes
        # ---- the back pointers are walked on the device (las_beam_backtrack); one read-back, then the reference's host-side objects
        W = n * selcap
        w_ids = torch.zeros(W, Umax, **i32)
        w_rows = torch.zeros(W, Umax, **i32)
        w_len = torch.zeros(W, **i32)
        w_score = torch.zeros(W, device=dev)
        _hip.check(lib.las_beam_backtrack(ctypes.byref(ba), _hip.p(w_ids), _hip.p(w_rows), _hip.p(w_len), _hip.p(w_score), _hip.stream()),
                   "las_beam_backtrack")
        # every hypothesis' alignments in ONE gather: item 0 of a hypothesis is the all-zero slab [Umax], item 1 + p the row it occupied at step p
        pos = torch.arange(Umax + 1, device=dev)
        item = (pos.unsqueeze(0) <= w_len.unsqueeze(1)) & (w_len > 0).unsqueeze(1)                 # [W, Umax + 1]
        t_idx = torch.cat([torch.full((1,), Umax, device=dev, dtype=torch.int64), pos[:-1]]).expand(W, -1)[item]
        r_idx = torch.cat([torch.zeros(W, 1, **i32), w_rows], 1)[item].long()
        g_att = alphas_hist[t_idx, r_idx]
        lens, ids_h, sc_h = w_len.cpu().numpy(), w_ids.cpu().numpy(), w_score.cpu().numpy()
        _hip.check_status(dev)
        mark("read back")
        # The reference's final ranking (_select_best_k: stable argsort of log_prob [/ (len - 1)], the last beam_size) on the read-back
        # arrays; host objects are built for the SELECTED hypotheses only (round 5: one BeamState + one tensor slice for each of the
        # <= 2 x beam candidates of every utterance was 4.2 ms behind a 64-utterance search, a tenth of the batch)
        nz = np.nonzero(lens)[0]
        offs = np.concatenate(([0], np.cumsum(lens[nz] + 1)))[:-1] if nz.size else np.zeros(0, np.int64)
        results = [[] for _ in range(n)]
        first = np.searchsorted(nz, np.arange(n + 1) * selcap)
        for u in range(n):
            ws, of = nz[first[u]:first[u + 1]], offs[first[u]:first[u + 1]]
            if ws.size == 0:
                continue
            sc = sc_h[ws].astype(np.float32)
            key = sc / lens[ws].astype(np.float32) if NORM else sc
            for i in np.argsort(key, kind="stable")[-self.beam_size:].tolist():
                w, ln_ = int(ws[i]), int(lens[ws[i]])
                results[u].append(BeamState((self.start_id, ids_h[w, :ln_]), np.float32(sc_h[w]),
                                            _AttRows(g_att, int(of[i]), ln_ + 1, Tps[u]), None, None))
        return results

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
