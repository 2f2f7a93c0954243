"""K10b: the device-resident beam loop (las_beam_loop_step) and the batched product decode built on it.

 * The kernel, driven by the goldens' toy spellers, must reproduce the REFERENCE's hypotheses exactly (token ids,
   float32 scores, alignment-history length): every single-utterance case G5 (beam 1..16) and the four utterances of
   G6 processed TOGETHER in one launch per step (nutt = 4; they retire at different steps, two by EOS, two by step
   exhaustion).  Pruning, EOS retirement, termination and the state gather all happen on the device; the host
   only back-tracks the records at the end.
 * BeamSearch.decode_batch (product): several utterances of different lengths in one batch == one utterance at a time
   == the oracle's beam search with the oracle's Speller; beam 16 with a 2 x 512 char RNNLM (configs[4]).
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

import helpers
from helpers import make_args, synthetic_batch, lm_params, oracle_lm, oracle_decode

pytestmark = pytest.mark.gpu


def _toy_mod():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(helpers.ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run_loop(cases):
    """cases: golden records sharing (V, beam, Tp, D).  Returns per utterance the list of (token_ids, log_prob, n_att)."""
    import ctypes
    from las import _hip
    m = _toy_mod()
    dev = "cuda"
    n = len(cases)
    V, beam, Tp, D = cases[0]["V"], cases[0]["beam"], cases[0]["Tp"], cases[0]["D"]
    toys = [m.toy_speller(c["seed"], V, Tp, D) for c in cases]
    NL = toys[0]["NL"]
    dec_steps = [int(c["audiolen"] * c["convert_rate"]) for c in cases]
    Umax, N, selcap = max(dec_steps), n * beam, 3 * beam
    i32 = dict(dtype=torch.int32, device=dev)
    T = dict(logits=torch.zeros(n, beam, V, device=dev), score=torch.zeros(n, beam, device=dev), length=torch.zeros(n, beam, **i32),
             nlive=torch.full((n,), beam, **i32), nsel=torch.zeros(n, **i32), done=torch.zeros(n, **i32),
             dec_step=torch.tensor(dec_steps, **i32), step=torch.zeros(1, **i32),
             hist_parent=torch.zeros(Umax, n, beam, **i32), hist_token=torch.zeros(Umax, n, beam, **i32),
             hist_slot=torch.zeros(Umax, n, beam, **i32), hist_score=torch.zeros(Umax, n, beam, device=dev),
             hist_n=torch.zeros(Umax, n, **i32), sel_t=torch.zeros(n, selcap, **i32), sel_j=torch.zeros(n, selcap, **i32),
             src_row=torch.zeros(n, beam, **i32), next_token=torch.full((N,), 1, **i32))
    st_new = [torch.zeros(N, D, device=dev) for _ in range(NL)]
    st_prev = [torch.zeros(N, D, device=dev) for _ in range(NL)]
    al_new, al_prev = torch.zeros(N, Tp, device=dev), torch.zeros(N, Tp, device=dev)
    ba = _hip.BeamLoopArgs()
    for k, t in T.items():
        setattr(ba, k, t.data_ptr())
    ba.nutt, ba.beam, ba.V, ba.Umax, ba.selcap, ba.topn, ba.start_id, ba.end_id = n, beam, V, Umax, selcap, 64, 1, 2
    ba.ntens = NL + 1
    for k in range(NL):
        ba.state_in[k], ba.state_out[k], ba.state_width[k] = st_new[k].data_ptr(), st_prev[k].data_ptr(), D
    ba.state_in[NL], ba.state_out[NL], ba.state_width[NL] = al_new.data_ptr(), al_prev.data_ptr(), Tp
    al_hist, al_snap = torch.full((Umax, N, Tp), -1.0, device=dev), []       # the per-step record of the alignments (filing workgroups)
    ba.file_in, ba.file_out, ba.file_width = al_new.data_ptr(), al_hist.data_ptr(), Tp
    natt = np.zeros((Umax, N), np.int32)
    for t in range(Umax):
        # the "model": every row of every utterance through its utterance's toy step (host numpy, as in the goldens)
        tok = T["next_token"].cpu().numpy()
        prev_al = al_prev.cpu().numpy()
        states = [s.cpu().numpy() for s in st_prev]
        lg = np.zeros((n, beam, V), np.float32)
        for u in range(n):
            rows = slice(u * beam, (u + 1) * beam)
            l_u, new_u, a_u = m.toy_step(toys[u], tok[rows], prev_al[rows], np.stack([s[rows] for s in states]))
            lg[u] = l_u
            for k in range(NL):
                st_new[k][rows] = torch.tensor(new_u[k], device=dev)
            al_new[rows] = torch.tensor(a_u, device=dev)
        T["logits"].copy_(torch.tensor(lg))
        al_snap.append(al_new.clone())
        _hip.check(_hip.lib().las_beam_loop_step(ctypes.byref(ba), _hip.stream()), "las_beam_loop_step")
        assert int(T["step"][0]) == t + 1
        if bool(T["done"].all()):
            break
    hp, ht, hs = T["hist_parent"].cpu().numpy(), T["hist_token"].cpu().numpy(), T["hist_slot"].cpu().numpy()
    hsc, st_, sj_, ns_ = T["hist_score"].cpu().numpy(), T["sel_t"].cpu().numpy(), T["sel_j"].cpu().numpy(), T["nsel"].cpu().numpy()
    assert torch.equal(al_hist[:len(al_snap)], torch.stack(al_snap)) and bool((al_hist[len(al_snap):] == -1.0).all())
    # the device walk of the back pointers (las_beam_backtrack) against the host walk below
    w_ids, w_rows = torch.full((n * selcap, Umax), -7, **i32), torch.full((n * selcap, Umax), -7, **i32)
    w_len, w_sc = torch.full((n * selcap,), -7, **i32), torch.zeros(n * selcap, device=dev)
    _hip.check(_hip.lib().las_beam_backtrack(ctypes.byref(ba), w_ids.data_ptr(), w_rows.data_ptr(), w_len.data_ptr(), w_sc.data_ptr(),
                                             _hip.stream()), "las_beam_backtrack")
    w_ids, w_rows, w_len, w_sc = w_ids.cpu().numpy(), w_rows.cpu().numpy(), w_len.cpu().numpy(), w_sc.cpu().numpy()
    out = []
    for u in range(n):
        sel = []
        assert (w_len[u * selcap + min(int(ns_[u]), selcap):(u + 1) * selcap] == 0).all()
        for s_i in range(int(ns_[u])):
            tt, j = int(st_[u, s_i]), int(sj_[u, s_i])
            lp = np.float32(hsc[tt, u, j])
            ids = []
            while True:
                ids.append(int(ht[tt, u, j]))
                slot = int(hp[tt, u, j])
                if tt == 0:
                    break
                j = int(hs[tt - 1, u, slot])
                tt -= 1
            w = u * selcap + s_i
            assert w_len[w] == len(ids) and w_ids[w, :len(ids)].tolist() == ids[::-1] and w_sc[w] == lp
            assert (w_rows[w, :len(ids)] // beam == u).all() and w_rows[w, len(ids) - 1] == u * beam + int(hp[int(st_[u, s_i]), u, int(sj_[u, s_i])])
            sel.append(([1] + ids[::-1], lp))
        norm = np.asarray([lp / (len(ids) - 1) for ids, lp in sel])
        order = np.argsort(norm, kind="stable")[-beam:]
        out.append([sel[i] for i in order])
    return out


def _assert_matches(got, case):
    assert [ids for ids, _ in got] == [g["token_ids"] for g in case["hyps"]], case["seed"]
    for (ids, lp), g in zip(got, case["hyps"]):
        assert float(lp) == pytest.approx(g["log_prob"], rel=1e-6)
        assert len(ids) == g["n_att"]                       # one alignment per token incl. the initial zeros


def test_device_loop_reproduces_every_single_utterance_golden(golden):
    for c in golden["G5"]:
        _assert_matches(_run_loop([c])[0], c)


def test_device_loop_four_utterances_in_one_launch_reproduce_the_goldens(golden):
    got = _run_loop(golden["G6"])
    assert len(got) == 4
    for g, c in zip(got, golden["G6"]):
        _assert_matches(g, c)


@pytest.mark.parametrize("cell,mode", [("lstm", "add"), ("rnn", "loc")])
def test_decode_batch_equals_one_at_a_time_and_the_oracle(cell, mode):
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from oracle import las_oracle as O
    from utils.tokenizer import CharEncoder
    NL = 1 if cell == "lstm" else 2
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=NL, embedding_size=32, attention_size=32,
                     mode=mode, loc_kernel_size=7, loc_num_channels=3, beam_size=4, convert_rate=0.35, apply_lm=False)
    p0 = O.init_params(args, seed=21, cell=cell)
    p0["Speller/decode/dense/bias"][2] = 0.3          # let some hypotheses end
    L.set_cell(cell); L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, CharEncoder().token_to_id)
    bs = BeamSearch(args, las, CharEncoder().token_to_id, None)
    utts = []
    for k, T in enumerate((41, 29, 56)):               # different lengths -> different T', different step bounds
        xs, _ = synthetic_batch(1, T, 8, 30, seed=9 + k)
        utts.append(xs)
    batch = bs.decode_batch(None, utts)
    for xs, res in zip(utts, batch):
        one = bs.decode(None, xs)
        assert [b.token_ids for b in one] == [b.token_ids for b in res]
        assert [float(b.log_prob) for b in one] == [float(b.log_prob) for b in res]
        ref = oracle_decode(xs, p0, args, cell, 4)
        assert [b.token_ids for b in res] == [b.token_ids for b in ref]
        for a, b in zip(res, ref):
            assert float(a.log_prob) == pytest.approx(float(b.log_prob), abs=2e-3)
            assert len(a.att) == len(b.att)
            assert np.abs(a.att[-1].cpu().numpy() - b.att[-1]).max() < 1e-4


def test_beam16_with_2x512_char_rnnlm_matches_oracle():
    """configs[4]: beam 16 + char RNNLM shallow fusion (2 x LSTM-512, one-hot input, 28 symbols) at the product level."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from lang.char_rnn_model import CharRNN
    from oracle import las_oracle as O
    from utils.tokenizer import CharEncoder
    from helpers import lm_params as _lm_params, oracle_lm as _oracle_lm
    cell = "lstm"
    args = make_args(enc_units=64, num_enc_layers=2, dec_units=128, num_dec_layers=1, embedding_size=64, attention_size=64,
                     beam_size=16, convert_rate=0.3, apply_lm=True, lm_weight=0.5)
    p0 = O.init_params(args, seed=33, cell=cell)
    p0["Speller/decode/dense/bias"][2] = 0.5
    plm = _lm_params(np.random.RandomState(8), 28, 0, 512, 2)
    for k in plm:                                        # keep the 512-wide LM in a sane range
        plm[k] = (plm[k] * 0.3).astype(np.float32)
    L.set_cell(cell); L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0); st.load(plm)
    las = LAS(args, Listener, Speller, CharEncoder().token_to_id)
    lm = CharRNN(False, 1, 1, 28, 512, embedding_size=0, num_layers=2)
    bs = BeamSearch(args, las, CharEncoder().token_to_id, lm)
    utts = [synthetic_batch(1, T, 8, 30, seed=40 + k)[0] for k, T in enumerate((60, 47))]
    batch = bs.decode_batch(None, utts)
    olm = (_oracle_lm(plm, 0, 2), 512, 2)
    for xs, res in zip(utts, batch):
        ref = oracle_decode(xs, p0, args, cell, 16, lm=olm, lm_weight=0.5)
        assert len(res) == len(ref) and len(res) > 0
        assert [b.token_ids for b in res] == [b.token_ids for b in ref]
        for a, b in zip(res, ref):
            assert float(a.log_prob) == pytest.approx(float(b.log_prob), abs=5e-3)


@pytest.mark.parametrize("nutt,beam,V,k0,k1", [(3, 16, 30, 512, 512), (2, 4, 7, 64, 0), (2, 33, 30, 96, 32), (1, 16, 128, 64, 64)])
def test_projection_inside_the_beam_step_equals_projecting_first(nutt, beam, V, k0, k1):
    """las_beam_loop_step with proj_*: the logits it computes itself (bf16 operands, fp32 accumulation, written to `logits`) against torch,
    and its records against a second call from the same state that is GIVEN those logits."""
    import ctypes
    from las import _hip
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(nutt * 100 + beam + V)
    N, Umax, selcap = nutt * beam, 4, 3 * beam
    h0 = torch.randn(N, k0, generator=g).to(dev)
    h1 = torch.randn(N, k1, generator=g).to(dev) if k1 else None
    W = (torch.randn(k0 + k1, V, generator=g) * 0.2).to(dev)
    b = torch.randn(V, generator=g).to(dev)
    packed = _hip.skinny_pack(W, k0 + k1, V)
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    ref = rnd(h0).double() @ rnd(W[:k0]).double() + b.double()
    if k1:
        ref = ref + rnd(h1).double() @ rnd(W[k0:]).double()
    i32 = dict(dtype=torch.int32, device=dev)

    def state():
        return dict(logits=torch.zeros(nutt, beam, V, device=dev), score=torch.zeros(nutt, beam, device=dev), length=torch.zeros(nutt, beam, **i32),
                    nlive=torch.full((nutt,), beam, **i32), nsel=torch.zeros(nutt, **i32), done=torch.zeros(nutt, **i32),
                    dec_step=torch.full((nutt,), Umax, **i32), step=torch.ones(1, **i32),           # t = 1: every live hypothesis expands
                    hist_parent=torch.zeros(Umax, nutt, beam, **i32), hist_token=torch.zeros(Umax, nutt, beam, **i32),
                    hist_slot=torch.zeros(Umax, nutt, beam, **i32), hist_score=torch.zeros(Umax, nutt, beam, device=dev),
                    hist_n=torch.zeros(Umax, nutt, **i32), sel_t=torch.zeros(nutt, selcap, **i32), sel_j=torch.zeros(nutt, selcap, **i32),
                    src_row=torch.zeros(nutt, beam, **i32), next_token=torch.full((N,), 1, **i32))

    def run(T, proj):
        ba = _hip.BeamLoopArgs()
        for k, t in T.items():
            setattr(ba, k, t.data_ptr())
        ba.nutt, ba.beam, ba.V, ba.Umax, ba.selcap, ba.topn, ba.start_id, ba.end_id, ba.ntens = nutt, beam, V, Umax, selcap, 64, 1, 2, 0
        if proj:
            ba.proj_h0, ba.proj_k0, ba.proj_h1, ba.proj_k1 = h0.data_ptr(), k0, (h1.data_ptr() if k1 else None), k1
            ba.proj_w, ba.proj_b = packed.data_ptr(), b.data_ptr()
        _hip.check(_hip.lib().las_beam_loop_step(ctypes.byref(ba), _hip.stream()), "las_beam_loop_step")
        torch.cuda.synchronize()

    T1 = state()
    run(T1, True)
    got = T1["logits"].reshape(N, V)
    assert (got.double() - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max()))
    T2 = state()
    T2["logits"].copy_(T1["logits"])
    run(T2, False)
    for k in ("score", "length", "nlive", "nsel", "done", "hist_parent", "hist_token", "hist_slot", "hist_score", "hist_n", "sel_t", "sel_j",
              "src_row", "next_token", "step"):
        assert torch.equal(T1[k], T2[k]), k
    assert int(T1["hist_n"][1].min()) == min(beam, beam * (V - 1))


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_decode_batch_subword_vocabulary_v5000_location_aware(prec):
    """BASELINE configs[3]'s decode shape class: V = 5000 (beam x V = 20,000 candidates: the round-based ranking inside the loop kernel, the
    long form of the step -- the in-kernel projection serves <= 8 tiles) with location-aware attention, against the oracle's search."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from oracle import las_oracle as O
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=1, embedding_size=32, attention_size=32, unit="subword",
                     vocab_size=5000, mode="loc", loc_kernel_size=7, loc_num_channels=3, beam_size=4, convert_rate=0.3, apply_lm=False)
    p0 = O.init_params(args, seed=5, cell="lstm")
    p0["Speller/decode/dense/kernel"] = (p0["Speller/decode/dense/kernel"] * 6).astype(np.float32)   # spread the 5000 logits: no near ties
    p0["Speller/decode/dense/bias"][2] = 1.5
    L.set_cell("lstm"); L.set_precision(prec)
    st = V.reset_default_store(device="cuda"); st.load(p0)
    tok = {"<PAD>": 0, "<SOS>": 1, "<EOS>": 2}
    las = LAS(args, Listener, Speller, tok)
    bs = BeamSearch(args, las, tok, None)
    utts = [synthetic_batch(1, T, 8, 30, seed=3 + k)[0] for k, T in enumerate((37, 52))]
    got = bs.decode_batch(None, utts)
    for xs, res in zip(utts, got):
        ref = oracle_decode(xs, p0, args, "lstm", 4, prec=prec)
        assert len(res) == len(ref) > 0
        if prec == "f32":
            assert [b.token_ids for b in res] == [b.token_ids for b in ref]
        else:
            assert res[-1].token_ids == ref[-1].token_ids
        assert float(res[-1].log_prob) == pytest.approx(float(ref[-1].log_prob), abs=2e-3 if prec == "f32" else 5e-2)
