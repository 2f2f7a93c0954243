#!/usr/bin/env python
"""Training-quality surrogate for the dev-clean WER target (BASELINE.json metric; SURVEY 8(d): "bf16: compare loss curve /
token agreement"; reference README.md:104-108, test.py:127-136).  LibriSpeech is not in the image, so the check that the
speed mode LEARNS like the parity mode is made on a learnable synthetic corpus:

  * `LearnableCorpus`: utterances are sentences over a small lexicon; every character occupies `frames_per_char` frames whose
    39 MFCC-like features are that character's fixed random code + noise (+ delta-like channels), i.e. the transcript is
    recoverable from the features the way it is from speech;
  * the bench architecture (3 x pBLSTM-256 listener, LSTM-512 speller, additive attention) is trained with LAS.train for
    `--steps` steps in f32 AND in bf16 mode from the same initial weights, on the same batches in the same order;
  * reported: the loss of every step in both modes, and the greedy-decoding WER (LAS.inference + test.py's corpus WER) on the
    training utterances at the end.

Prints one JSON object; tests/test_gpu_train_quality.py asserts on it and profiles/r3_train_quality.json keeps a run."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

LEXICON = ["THE", "CAT", "SAT", "ON", "A", "MAT", "DOG", "RAN", "TO", "SEE", "RED", "SUN", "WE", "GO", "UP", "IN"]


class LearnableCorpus:
    def __init__(self, n_utt, frames=200, frames_per_char=8, seed=0, noise=0.3, feat_dim=13):
        from utils.tokenizer import CharEncoder
        self.tok = CharEncoder()
        rng = np.random.RandomState(seed)
        self.codes = rng.randn(self.tok.get_vocab_size(), feat_dim, 3).astype(np.float32)
        self.codes[..., 1] *= 0.5
        self.codes[..., 2] *= 0.316
        max_chars = frames // frames_per_char
        self.texts, self.audio, self.audiolen, self.y, self.tokenlen = [], [], [], [], []
        for _ in range(n_utt):
            while True:
                words = [LEXICON[i] for i in rng.randint(0, len(LEXICON), size=rng.randint(2, 6))]
                text = " ".join(words)
                if len(text) <= max_chars:
                    break
            ids = self.tok.encode(text, with_eos=False)
            x = np.zeros((frames, feat_dim, 3), np.float32)
            n = len(ids) * frames_per_char
            x[:n] = np.repeat(self.codes[ids], frames_per_char, 0) + noise * rng.randn(n, feat_dim, 3).astype(np.float32)
            yy = np.zeros(max_chars + 1, np.int32)
            yy[:len(ids)] = ids
            yy[len(ids)] = 2
            self.texts.append(text); self.audio.append(x); self.audiolen.append(n); self.y.append(yy); self.tokenlen.append(len(ids) + 1)
        self.audio = np.stack(self.audio); self.y = np.stack(self.y)
        self.audiolen = np.asarray(self.audiolen, np.int32); self.tokenlen = np.asarray(self.tokenlen, np.int32)

    def batch(self, idx):
        return (self.audio[idx], self.audiolen[idx]), (self.y[idx], self.tokenlen[idx])


def arch(**over):
    from helpers import make_args
    kw = dict(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
              mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30, convert_rate=0.166)
    kw.update(over)
    return make_args(**kw)


def initial_weights(args, seed=5):
    """one draw of the model's own initialisers (las.variables: glorot / uniform, as TF's defaults), shared by both modes"""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    L.set_cell("lstm"); L.set_precision("f32")
    st = V.reset_default_store(device="cuda", seed=seed)
    LAS(args, Listener, Speller, {}).build_variables()
    return {n: st.vars[n].detach().cpu().numpy().copy() for n in st.order}


def window_means(losses, w=25):
    return [float(np.mean(losses[i:i + w])) for i in range(0, len(losses) - w + 1, w)]


def run_mode(prec, corpus, p0, args, steps, B, order_seed=0):
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from las.utils import convert_idx_to_string, edit_distance
    L.set_cell("lstm"); L.set_precision(prec)
    st = V.reset_default_store(device="cuda"); st.load(p0)
    las = LAS(args, Listener, Speller, corpus.tok.id_to_token)
    n = len(corpus.texts)
    rng = np.random.RandomState(order_seed)
    losses = []
    perm, pos = rng.permutation(n), 0
    for _ in range(steps):
        if pos + B > n:
            perm, pos = rng.permutation(n), 0
        idx = np.sort(perm[pos:pos + B]); pos += B
        xs, ys = corpus.batch(idx)
        losses.append(las.train(xs, ys)[0])
    torch.cuda.synchronize()
    las.check_status()
    losses = [float(v) for v in losses]
    hyp = []
    for lo in range(0, n, B):
        idx = np.arange(lo, min(lo + B, n))
        xs, _ = corpus.batch(idx)
        # greedy decoding runs int(convert_rate * max audiolen) steps (las/las.py:310-312): 8 frames per character -> 1.33 steps per
        # character of the longest utterance, enough for its <EOS>; the attention mask follows the true lengths, as in training
        _, y_hat = las.inference(xs)
        hyp += [convert_idx_to_string(r, corpus.tok.id_to_token, "char") for r in y_hat.cpu().numpy().tolist()]
    # corpus WER as test.py:127-136 computes it: summed word-level edit distance / summed reference words
    pairs = [edit_distance(t.split(" "), h.split(" ")) for t, h in zip(corpus.texts, hyp)]
    wer = sum(e for e, _ in pairs) / sum(n for _, n in pairs)
    exact = float(np.mean([h == t for h, t in zip(hyp, corpus.texts)]))
    return {"loss": losses, "wer": wer, "exact": exact, "examples": list(zip(corpus.texts[:3], hyp[:3]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--fpc", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--noise", type=float, default=0.3)
    ap.add_argument("--modes", default="f32,bf16,f32p", help="f32p = the CONTROL: parity mode from initial weights perturbed by 1e-6 "
                    "(relative): how far two correct runs drift apart by themselves")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    corpus = LearnableCorpus(a.utts, a.frames, a.fpc, seed=1, noise=a.noise)
    args = arch(lr=a.lr)
    p0 = initial_weights(args)
    out = {"config": vars(a), "modes": {}}
    for prec in a.modes.split(","):
        if prec == "f32p":
            rng = np.random.RandomState(99)
            pp = {n: (v * (1.0 + 1e-6 * rng.randn(*v.shape))).astype(np.float32) for n, v in p0.items()}
            out["modes"][prec] = run_mode("f32", corpus, pp, args, a.steps, a.batch)
        else:
            out["modes"][prec] = run_mode(prec, corpus, p0, args, a.steps, a.batch)
        r = out["modes"][prec]
        r["window_mean_25"] = window_means(r["loss"])
        print("%s: mean loss per 25 steps %s -> %.4f, WER %.4f, exact %.3f, e.g. %s" % (
            prec, ["%.3f" % v for v in r["window_mean_25"]], r["loss"][-1], r["wer"], r["exact"], r["examples"][:2]), file=sys.stderr)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f)
    print(json.dumps({k: {"window_mean_25": [round(v, 4) for v in r["window_mean_25"]], "final_loss": round(r["loss"][-1], 4), "wer": r["wer"],
                          "exact": r["exact"]} for k, r in out["modes"].items()}))


if __name__ == "__main__":
    main()
