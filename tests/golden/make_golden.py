"""Generate golden vectors by IMPORTING the reference (build container only).

    python tests/golden/make_golden.py      # needs /root/reference

The reference cannot travel to the GPU box, so the outputs are committed as
small JSON fixtures next to this script.  Only data (inputs + expected outputs)
is written; no reference source text is stored.

What is pinned:
  G1  utils/tokenizer.py  CharEncoder.encode / tables
  G2  las/arguments.py    parse_args() defaults
  G3  las/utils.py        edit_distance
  G4  las/utils.py        convert_idx_to_string (char + subword modes)
  G5/G6 las/beam_search.py BeamSearch.decode control flow + _select_best_k, with an
                          injected numpy "toy speller" step function (apply_lm=False;
                          the LM branch of the reference is syntactically broken)
TensorFlow is absent: a stub module satisfies the imports; none of the pinned
functions touch it.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.path.insert(0, REF)
    import importlib
    tok = importlib.import_module("utils.tokenizer")
    argm = importlib.import_module("las.arguments")
    ut = importlib.import_module("las.utils")
    bs = importlib.import_module("las.beam_search")
    return tok, argm, ut, bs


def toy_speller(seed, V, Tp, D, NL=2):
    """Deterministic numpy step function: fixed random weights; the logits depend on
    previous token, previous alignment and the decoder state, so hypotheses diverge."""
    rng = np.random.RandomState(seed)
    W_tok = rng.randn(V, D).astype(np.float32) * 0.7
    W_al = rng.randn(Tp, D).astype(np.float32) * 0.5
    W_st = [rng.randn(D, D).astype(np.float32) * 0.4 for _ in range(NL)]
    W_out = rng.randn(D, V).astype(np.float32)
    W_att = rng.randn(D, Tp).astype(np.float32)
    # bias EOS so that hypotheses terminate at different steps
    b_out = np.zeros(V, np.float32)
    b_out[2] = 0.5
    return dict(W_tok=W_tok, W_al=W_al, W_st=W_st, W_out=W_out, W_att=W_att, b_out=b_out, NL=NL, D=D, Tp=Tp, V=V)


def toy_step(toy, prev_ids, prev_align, states):
    """states: float32 [NL, N, D].  Returns logits [N,V], tuple of NL [N,D], alphas [N,Tp]."""
    prev_ids = np.asarray(prev_ids)
    al = np.asarray(prev_align, np.float32)
    x = toy["W_tok"][prev_ids] + al @ toy["W_al"]
    new = []
    for l in range(toy["NL"]):
        h = np.tanh(x + states[l] @ toy["W_st"][l]).astype(np.float32)
        new.append(h)
        x = h
    logits = (x @ toy["W_out"] + toy["b_out"]).astype(np.float32)
    e = x @ toy["W_att"]
    e = e - e.max(-1, keepdims=True)
    a = np.exp(e)
    a = (a / a.sum(-1, keepdims=True)).astype(np.float32)
    return logits, tuple(new), a


def main():
    tok, argm, ut, bs = _import_reference()
    out = {}

    # ---- G1 CharEncoder
    enc = tok.CharEncoder()
    sents = ["HELLO WORLD", "A", "", "THE QUICK BROWN FOX", " LEADING SPACE", "TRAILING SPACE ",
             "DOUBLE  SPACE", "ZEBRA", "MISTER QUILTER IS THE APOSTLE OF THE MIDDLE CLASSES",
             "AND WE ARE GLAD TO WELCOME HIS GOSPEL", "X Y Z", "ABCDEFGHIJKLMNOPQRSTUVWXYZ"]
    out["G1"] = {
        "vocab_size": enc.get_vocab_size(),
        "token_to_id": enc.token_to_id,
        "id_to_token": {str(k): v for k, v in enc.id_to_token.items()},
        "cases": [{"s": s, "eos": e, "ids": enc.encode(s, e)} for s in sents for e in (True, False)],
    }

    # ---- G2 parse_args defaults
    argv, sys.argv = sys.argv, ["prog"]
    try:
        a = argm.parse_args()
    finally:
        sys.argv = argv
    out["G2"] = vars(a)
    out["G2_str2bool"] = {s: argm.str2bool(s) for s in ["yes", "True", "t", "Y", "1", "no", "FALSE", "f", "n", "0"]}

    # ---- G3 edit_distance
    rng = np.random.RandomState(7)
    words = ["A", "THE", "CAT", "DOG", "SAT", "ON", "MAT", "RAN", "FAR", "AWAY", "", "BIG"]
    cases = []
    for _ in range(60):
        n, m = rng.randint(0, 9), rng.randint(0, 9)
        s1 = [words[i] for i in rng.randint(0, len(words), n)]
        s2 = [words[i] for i in rng.randint(0, len(words), m)]
        e, ln = ut.edit_distance(s1, s2)
        cases.append({"s1": s1, "s2": s2, "e": float(e), "n": int(ln)})
    # the way test.py:127-136 calls it: "".split(" ") -> ['']
    for r, h in [("", ""), ("A B", ""), ("", "A B"), ("THE CAT SAT", "THE CAT SAT"), ("THE CAT", "THE BAT SAT")]:
        e, ln = ut.edit_distance(r.split(" "), h.split(" "))
        cases.append({"s1": r.split(" "), "s2": h.split(" "), "e": float(e), "n": int(ln)})
    out["G3"] = cases

    # ---- G4 convert_idx_to_string
    id2c = enc.id_to_token
    g4 = []
    idlists = [[11, 8, 15, 15, 18, 3, 26, 18, 21, 15, 7, 2], [2], [0, 0, 0], [1, 4, 3, 3, 5, 2, 6, 7], [3, 4, 3],
               [4, 5, 6], [], [4, 2, 5, 2, 6], [0, 4, 0, 5, 2, 0, 0], [3, 3, 3, 2]]
    for ids in idlists:
        g4.append({"unit": "char", "ids": ids, "out": ut.convert_idx_to_string(ids, id2c, "char")})
    sub = {0: "<PAD>", 1: "<SOS>", 2: "<EOS>", 3: "<unk>", 4: "the</w>", 5: "ca", 6: "t</w>", 7: "s", 8: "at</w>", 9: "</w>"}
    for ids in [[4, 5, 6, 7, 8, 2], [5, 6, 2, 4], [4, 4, 9, 4], [2], [], [3, 4, 0, 0]]:
        g4.append({"unit": "subword", "ids": ids, "table": {str(k): v for k, v in sub.items()},
                   "out": ut.convert_idx_to_string(ids, sub, "subword")})
    out["G4"] = g4

    # ---- G5 beam search control flow
    g5 = []
    confs = [(11, 30, 12, 16, 1, 0.5, 40), (12, 30, 9, 8, 4, 0.9, 33), (13, 30, 20, 16, 16, 0.4, 60),
             (14, 80, 15, 16, 8, 0.7, 30), (15, 30, 10, 8, 4, 0.12, 25), (16, 30, 14, 8, 10, 1.0, 18)]
    for seed, V, Tp, D, beam, rate, audiolen in confs:
        toy = toy_speller(seed, V, Tp, D)
        o = object.__new__(bs.BeamSearch)
        o.args = types.SimpleNamespace(convert_rate=rate, apply_lm=False, lm_weight=0.0)
        o.beam_size = beam
        o.start_id, o.end_id = 1, 2
        h = np.zeros((1, Tp, 4), np.float32)
        o._get_encode = lambda sess, audio, audiolen_, h=h: (h, np.array([Tp]))
        o._get_dec_init = lambda sess, D=D: (np.zeros((1, D), np.float32), np.zeros((1, D), np.float32))

        def _get_decode(sess, enc_out, enc_len, prev_ids, prev_align, packed, toy=toy):
            st = np.asarray(packed, np.float32)
            return toy_step(toy, prev_ids, prev_align, st)
        o._get_decode = _get_decode
        res = o.decode(None, (np.zeros((1, audiolen, 13, 3), np.float32), np.array([audiolen])))
        g5.append({"seed": seed, "V": V, "Tp": Tp, "D": D, "beam": beam, "convert_rate": rate, "audiolen": audiolen,
                   "hyps": [{"token_ids": [int(t) for t in b.token_ids], "log_prob": float(b.log_prob),
                             "n_att": len(b.att)} for b in res]})
    out["G5"] = g5

    # ---- G6: four utterances that share (V, beam, T', D) -- the batched device loop (las_beam_loop_step, nutt = 4)
    # must reproduce each of them; same reference control flow as G5
    g6 = []
    for seed, rate, audiolen in [(21, 0.5, 40), (22, 0.9, 22), (23, 0.3, 50), (24, 1.0, 9)]:
        V, Tp, D, beam = 30, 12, 12, 8
        toy = toy_speller(seed, V, Tp, D)
        o = object.__new__(bs.BeamSearch)
        o.args = types.SimpleNamespace(convert_rate=rate, apply_lm=False, lm_weight=0.0)
        o.beam_size = beam
        o.start_id, o.end_id = 1, 2
        h = np.zeros((1, Tp, 4), np.float32)
        o._get_encode = lambda sess, audio, audiolen_, h=h: (h, np.array([Tp]))
        o._get_dec_init = lambda sess, D=D: (np.zeros((1, D), np.float32), np.zeros((1, D), np.float32))

        def _get_decode6(sess, enc_out, enc_len, prev_ids, prev_align, packed, toy=toy):
            return toy_step(toy, prev_ids, prev_align, np.asarray(packed, np.float32))
        o._get_decode = _get_decode6
        res = o.decode(None, (np.zeros((1, audiolen, 13, 3), np.float32), np.array([audiolen])))
        g6.append({"seed": seed, "V": V, "Tp": Tp, "D": D, "beam": beam, "convert_rate": rate, "audiolen": audiolen,
                   "hyps": [{"token_ids": [int(t) for t in b.token_ids], "log_prob": float(b.log_prob),
                             "n_att": len(b.att)} for b in res]})
    out["G6"] = g6

    # ---- G7: char RNNLM host pieces (lang/char_rnn_model.py BatchGenerator, train_lm.py text_cleaning / create_vocab)
    import importlib
    if not hasattr(np, "float"):
        np.float = float          # numpy-1.17 alias used by the reference's BatchGenerator (requirements.txt pins numpy==1.17.4)
    crm = importlib.import_module("lang.char_rnn_model")
    # train_lm.py cannot be imported as a module without side effects on sys.argv only; its helpers are plain functions
    cwd = os.getcwd()
    import tempfile
    os.chdir(tempfile.mkdtemp())
    os.makedirs("data", exist_ok=True)                       # text_cleaning writes data/libri_cleaned.txt
    tl = importlib.import_module("train_lm")
    raw = "Hello, World!\n\nIt's 9 o'clock -- \"time\" to go?  Yes: go_now; (really)\nlast line"
    cleaned = tl.text_cleaning(raw)
    v2i, i2v, vs = tl.create_vocab()
    os.chdir(cwd)
    text = "THE QUICK BROWN FOX. JUMPS OVER THE LAZY DOG. AND RUNS AWAY"
    gen = crm.BatchGenerator(text, 4, 3, vs, v2i, i2v)
    batches = [[[int(x) for x in b] for b in gen.next()] for _ in range(3)]
    out["G7"] = {"raw": raw, "cleaned": cleaned, "vocab": v2i, "text": text, "batch_size": 4, "n_unrollings": 3, "batches": batches,
                 "strings": crm.batches2string(gen.next(), i2v)}

    with open(os.path.join(HERE, "reference_host_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "reference_host_golden.json"))


if __name__ == "__main__":
    main()
