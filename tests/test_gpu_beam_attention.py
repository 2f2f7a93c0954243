"""K10 + decode-side API parity on the GPU.
 * las_beam_step (device pruning kernel) driven by the reference goldens' toy spellers must reproduce the
   REFERENCE's hypotheses (token ids, float32 scores) exactly -- fixtures G5.
 * BeamSearch.decode (product: fused Speller step + K10) vs oracle.beam_search with the oracle's Speller.
 * AdditiveAttention / LocationAwareAttention single-step objects vs the oracle (edge cases: len 0, len > T).
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


def test_beam_step_kernel_reproduces_reference_goldens(golden):
    from las import _hip
    m = _toy_mod()
    dev = "cuda"
    for c in golden["G5"]:
        V, Tp, D, beam = c["V"], c["Tp"], c["D"], c["beam"]
        toy = m.toy_speller(c["seed"], V, Tp, D)
        dec_step = int(c["audiolen"] * c["convert_rate"])
        # host bookkeeping identical to the reference; pruning on device
        hyps = [dict(ids=[1], lp=np.float32(0), att=np.zeros(Tp, np.float32),
                     st=tuple(np.zeros((1, D), np.float32) for _ in range(toy["NL"])))] * beam
        selected, t = [], 0
        bufs = dict(score=torch.zeros(1, beam, device=dev), length=torch.zeros(1, beam, dtype=torch.int32, device=dev),
                    nlive=torch.zeros(1, dtype=torch.int32, device=dev), par=torch.zeros(1, beam, dtype=torch.int32, device=dev),
                    tok=torch.zeros(1, beam, dtype=torch.int32, device=dev), osc=torch.zeros(1, beam, device=dev),
                    on=torch.zeros(1, dtype=torch.int32, device=dev), lg=torch.zeros(1, beam, V, device=dev))
        while t < dec_step and len(selected) < beam:
            N = len(hyps)
            st = np.stack([np.concatenate([h["st"][l] for h in hyps], 0) for l in range(toy["NL"])])
            logits, new, al = m.toy_step(toy, [h["ids"][-1] for h in hyps], np.stack([h["att"] for h in hyps]), st)
            bufs["lg"][0, :N] = torch.tensor(logits)
            bufs["score"][0, :N] = torch.tensor([float(h["lp"]) for h in hyps])
            bufs["length"][0, :N] = torch.tensor([len(h["ids"]) - 1 for h in hyps], dtype=torch.int32)
            bufs["nlive"][0] = N
            _hip.check(_hip.lib().las_beam_step(_hip.p(bufs["lg"]), _hip.p(bufs["score"]), _hip.p(bufs["length"]),
                                                _hip.p(bufs["nlive"]), 1, beam, V, 64, t, 1, _hip.p(bufs["par"]),
                                                _hip.p(bufs["tok"]), _hip.p(bufs["osc"]), _hip.p(bufs["on"]), _hip.stream()),
                       "las_beam_step")
            n = int(bufs["on"][0])
            nxt = []
            for j in range(n):
                i, v = int(bufs["par"][0, j]), int(bufs["tok"][0, j])
                h = dict(ids=hyps[i]["ids"] + [v], lp=np.float32(bufs["osc"][0, j].item()), att=al[i],
                         st=tuple(new[l][i:i + 1] for l in range(toy["NL"])))
                (selected if v == 2 else nxt).append(h)
            hyps = nxt
            t += 1
        if t == dec_step:
            selected.extend(hyps)
        norm = np.asarray([h["lp"] / (len(h["ids"]) - 1) for h in selected])
        order = np.argsort(norm, kind="stable")[-beam:]
        got = [selected[i] for i in order]
        assert [h["ids"] for h in got] == [g["token_ids"] for g in c["hyps"]], c["seed"]
        for h, g in zip(got, c["hyps"]):
            assert float(h["lp"]) == pytest.approx(g["log_prob"], rel=1e-6)


@pytest.mark.parametrize("cell,mode,NL", [("rnn", "add", 2), ("lstm", "add", 1), ("lstm", "loc", 1)])
def test_beam_search_decode_matches_oracle(cell, mode, NL):
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from oracle import las_oracle as O
    from utils.tokenizer import CharEncoder
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=NL, embedding_size=32, attention_size=32,
                     mode=mode, loc_kernel_size=7, loc_num_channels=3, beam_size=4, convert_rate=0.35, apply_lm=False)
    xs, _ = synthetic_batch(1, 41, 8, 30, seed=9)
    p0 = O.init_params(args, seed=21, cell=cell)
    p0["Speller/decode/dense/bias"][2] = 0.3          # let some hypotheses end
    L.set_cell(cell); L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, CharEncoder().token_to_id)
    bs = BeamSearch(args, las, CharEncoder().token_to_id, None)
    res = bs.decode(None, xs)
    # oracle: same control flow, oracle Speller as the step function
    po = O.to_torch(p0)
    with torch.no_grad():
        x = torch.tensor(xs[0]).reshape(1, -1, 39)
        h, el = O.pblstm_listener(x, xs[1], po, 2, cell)
        keys = h @ po["Speller/decode/attention/dense/kernel"]
        emb = po["embedding/embedding_matrix"]

        def step_fn(prev_ids, prev_al, states):
            N = len(prev_ids)
            stt = []
            for l in range(NL):
                if cell == "lstm":
                    stt.append((torch.cat([s[l][0] for s in states]), torch.cat([s[l][1] for s in states])))
                else:
                    stt.append(torch.cat([s[l] for s in states]))
            lg, ns, al = O.speller_decode(h.expand(N, -1, -1), el.repeat(N), stt, emb[torch.tensor(prev_ids)],
                                          torch.tensor(np.stack(prev_al), dtype=torch.float32), po, args, cell,
                                          keys.expand(N, -1, -1))
            outs = [tuple((ns[l][0][i:i + 1], ns[l][1][i:i + 1]) if cell == "lstm" else ns[l][i:i + 1] for l in range(NL))
                    for i in range(N)]
            return lg.numpy(), outs, al.numpy()
        z = torch.zeros(1, args.dec_units)
        init = tuple((z, z) if cell == "lstm" else z for _ in range(NL))
        ref = O.beam_search(step_fn, init, h.shape[1], int(xs[1][0] * args.convert_rate), 4, 1, 2)
    assert [b.token_ids for b in res] == [b.token_ids for b in ref]
    for a, b in zip(res, ref):
        assert float(a.log_prob) == pytest.approx(float(b.log_prob), abs=2e-3)
        assert len(a.att) == len(b.att)
        assert np.abs(a.att[-1].cpu().numpy() - b.att[-1]).max() < 1e-4


@pytest.mark.parametrize("mode", ["add", "loc"])
def test_attention_objects_match_oracle(mode):
    from las import layers as L, variables as V
    from oracle import las_oracle as O
    args = make_args(enc_units=24, dec_units=32, num_dec_layers=1, attention_size=16, mode=mode, loc_kernel_size=5,
                     loc_num_channels=4, embedding_size=8)
    p0 = O.init_params(args, seed=8, cell="rnn")
    L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0)
    if mode == "add":
        att = L.AdditiveAttention(48, 32, 16)
    else:
        att = L.LocationAwareAttention(48, 32, 16, 5, 4)
    g = torch.Generator().manual_seed(0)
    hidden = torch.randn(5, 9, 48, generator=g)
    state = torch.randn(5, 32, generator=g)
    align = torch.softmax(torch.randn(5, 9, generator=g), -1)
    seqlen = np.array([9.0, 4.0, 1.0, 0.0, 12.0])        # full, ragged, single frame, EMPTY, longer than T
    ctx, al = att(hidden.cuda(), state.cuda(), align.cuda(), seqlen)
    with torch.no_grad():
        ctx_o, al_o = O.attention_step(hidden, state, align, seqlen, O.to_torch(p0), mode)
    assert (al.cpu() - al_o).abs().max().item() < 1e-5
    assert (ctx.cpu() - ctx_o).abs().max().item() < 1e-4
    assert abs(float(al[3].sum()) - 1.0) < 1e-5 and float((al[3] - 1 / 9).abs().max()) < 1e-6   # all masked -> uniform
    m = att.mask(np.array([2, 3, 1]), 3)
    assert m.tolist() == [[1, 1, 0], [1, 1, 1], [1, 0, 0]]                                      # las/layers.py:182-186


@pytest.mark.parametrize("E", [0, 12])
def test_char_rnnlm_step_matches_oracle(E):
    """R1: CharRNN inference step (lang/char_rnn_model.py:54-142) vs oracle.lm_step over 3 chained steps."""
    from las import layers as L, variables as V
    from lang.char_rnn_model import CharRNN, create_vocab
    from oracle import las_oracle as O
    V_lm, H, NL, N = 28, 32, 2, 5
    assert create_vocab()[2] == 28 and create_vocab()[0]['A'] == 2
    rng = np.random.RandomState(3)
    p = lm_params(rng, V_lm, E, H, NL)
    L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p)
    lm = CharRNN(False, 1, 1, V_lm, H, embedding_size=E, num_layers=NL)
    olm = oracle_lm(p, E, NL)
    states = [lm.zero_state(1) for _ in range(N)]
    ostate = [(torch.zeros(N, H), torch.zeros(N, H)) for _ in range(NL)]
    for it in range(3):
        ids = rng.randint(0, V_lm, N)
        logits, states = lm.step(ids, states)
        with torch.no_grad():
            lo, ostate = O.lm_step(torch.tensor(ids), ostate, olm)
        assert (logits.cpu() - lo).abs().max().item() < 1e-4
        assert (states[2][1][1].cpu() - ostate[1][1][2]).abs().max().item() < 1e-5


def test_beam_search_with_lm_fusion_matches_oracle():
    """B6: shallow fusion (evident intent of las/beam_search.py:109-116): logits[:,2:] += lm_weight * lm_logits."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.beam_search import BeamSearch
    from lang.char_rnn_model import CharRNN
    from oracle import las_oracle as O
    from utils.tokenizer import CharEncoder
    cell, NL = "lstm", 1
    args = make_args(enc_units=48, num_enc_layers=2, dec_units=64, num_dec_layers=NL, embedding_size=32, attention_size=32,
                     beam_size=4, convert_rate=0.3, apply_lm=True, lm_weight=0.5)
    xs, _ = synthetic_batch(1, 41, 8, 30, seed=4)
    p0 = O.init_params(args, seed=31, cell=cell)
    p0["Speller/decode/dense/bias"][2] = 0.3
    plm = lm_params(np.random.RandomState(8), 28, 0, 24, 2)
    L.set_cell(cell); L.set_precision("f32")
    st = V.reset_default_store(device="cuda"); st.load(p0); st.load(plm)
    las = LAS(args, Listener, Speller, CharEncoder().token_to_id)
    lm = CharRNN(False, 1, 1, 28, 24, embedding_size=0, num_layers=2)
    res = BeamSearch(args, las, CharEncoder().token_to_id, lm).decode(None, xs)
    po = O.to_torch(p0); olm = oracle_lm(plm, 0, 2)
    with torch.no_grad():
        x = torch.tensor(xs[0]).reshape(1, -1, 39)
        h, el = O.pblstm_listener(x, xs[1], po, 2, cell)
        keys = h @ po["Speller/decode/attention/dense/kernel"]
        emb = po["embedding/embedding_matrix"]

        def step_fn(prev_ids, prev_al, states):
            N = len(prev_ids)
            stt = [(torch.cat([s[0][0] for s in states]), torch.cat([s[0][1] for s in states]))]
            lg, ns, al = O.speller_decode(h.expand(N, -1, -1), el.repeat(N), stt, emb[torch.tensor(prev_ids)],
                                          torch.tensor(np.stack(prev_al), dtype=torch.float32), po, args, cell, keys.expand(N, -1, -1))
            return lg.numpy(), [((ns[0][0][i:i + 1], ns[0][1][i:i + 1]),) for i in range(N)], al.numpy()

        def lm_fn(ids, states):
            stt = [(torch.stack([s[l][0] for s in states]), torch.stack([s[l][1] for s in states])) for l in range(2)]
            lo, ns = O.lm_step(torch.tensor(ids), stt, olm)
            return lo.numpy(), [tuple((ns[l][0][i], ns[l][1][i]) for l in range(2)) for i in range(len(ids))]
        z = torch.zeros(1, args.dec_units)
        lm0 = tuple((torch.zeros(24), torch.zeros(24)) for _ in range(2))
        ref = O.beam_search(step_fn, ((z, z),), h.shape[1], int(xs[1][0] * args.convert_rate), 4, 1, 2,
                            lm_fn=lm_fn, lm_init=lm0, lm_weight=0.5)
    assert [b.token_ids for b in res] == [b.token_ids for b in ref]
    for a, b in zip(res, ref):
        assert float(a.log_prob) == pytest.approx(float(b.log_prob), abs=2e-3)


@pytest.mark.parametrize("beam,V,t", [(16, 30, 3), (16, 30, 0), (4, 7, 2), (8, 64, 5), (16, 5000, 2), (3, 200, 1), (33, 15, 4)])
def test_beam_step_ordering_with_ties_matches_a_stable_sort(beam, V, t):
    """The ranking of las_beam_step on grids full of ties against a plain sort by the reference's key (score / len, hypothesis, logit,
    token -- the order a stable ascending sort of las/beam_search.py:119-152's candidate bank gives): equal logits inside a
    hypothesis, equal normalised scores across hypotheses, -inf, NaN (never ranks), +-0.  beam x V <= 512 takes the one-wave
    bisection path, larger grids the round-based path."""
    import numpy as np
    from las import _hip
    dev = "cuda"
    rng = np.random.RandomState(beam * 131 + V + t)
    nutt = 5
    lg = rng.choice(np.array([-2.5, -1.0, -0.5, 0.0, -0.0, 0.25, 1.0, 3.0], np.float32), size=(nutt, beam, V)).astype(np.float32)
    lg[0] = rng.randn(beam, V).astype(np.float32)                               # one utterance without engineered ties
    lg[1, :, : V // 2] = -np.inf
    lg[2, 0, 1::3] = np.nan
    if nutt > 3:
        lg[3] = 0.5                                                             # everything ties on the logit
    score = rng.choice(np.array([0.0, -1.0, -2.0, 2.0], np.float32), size=(nutt, beam)).astype(np.float32)
    length = rng.randint(0, 4, size=(nutt, beam)).astype(np.int32)
    nlive = np.array([beam, beam, max(beam - 1, 1), beam, 1], np.int32)
    start_id = 1
    d = lambda a: torch.tensor(a, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    outp, outt, outn = torch.full((nutt, beam), -1, **i32), torch.full((nutt, beam), -1, **i32), torch.zeros(nutt, **i32)
    outs = torch.zeros(nutt, beam, device=dev)
    d_lg, d_score, d_length, d_nlive = d(lg), d(score), d(length), d(nlive)      # (alive until the results are read)
    _hip.check(_hip.lib().las_beam_step(_hip.p(d_lg), _hip.p(d_score), _hip.p(d_length), _hip.p(d_nlive), nutt, beam, V, 64, t, start_id,
                                        _hip.p(outp), _hip.p(outt), _hip.p(outs), _hip.p(outn), _hip.stream()), "las_beam_step")
    outp, outt, outs, outn = outp.cpu().numpy(), outt.cpu().numpy(), outs.cpu().numpy(), outn.cpu().numpy()
    for u in range(nutt):
        nb = 1 if t == 0 else int(min(nlive[u], beam))
        cands = []
        for i in range(nb):
            for v in range(V):
                if t > 0 and v == start_id:
                    continue
                l = lg[u, i, v]
                norm = np.float32(np.float32(score[u, i] + l) / np.float32(length[u, i] + 1))
                if np.isnan(norm):
                    continue
                cands.append((float(norm), i, float(l), v))
        cands.sort()                                                             # -0.0 == 0.0 in the tuple compare, like the float compares of the kernel
        top = cands[-beam:]                                                      # ascending, best last
        assert outn[u] == len(top), (u, outn[u], len(top))
        assert [(int(outp[u, j]), int(outt[u, j])) for j in range(len(top))] == [(c[1], c[3]) for c in top], u
        for j, c in enumerate(top):
            assert outs[u, j] == np.float32(score[u, c[1]] + np.float32(c[2]))
