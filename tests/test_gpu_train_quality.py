"""Training-quality surrogate for the dev-clean WER target of BASELINE.json's metric (VERDICT r2 Missing #2; SURVEY 8(d): "bf16:
compare loss curve / token agreement"; reference README.md:104-108, test.py:127-136).  LibriSpeech is not in the image, so
tools/train_quality.py trains the bench architecture on a learnable synthetic corpus (the transcript is encoded in the
MFCC-like features) from ONE set of initial weights in three runs: parity mode (f32), speed mode (bf16), and -- the control --
parity mode again from weights perturbed by 1e-6 (relative).

What can be asserted about two training runs of a recurrent attention model: their trajectories decorrelate (chaotically) after
~100 steps whatever the cause of the first difference, so "the curves agree to 2 % at every checkpoint" is not a property that
even two CORRECT runs have -- the control run measures exactly that.  Asserted:
  * the mean loss per 25 steps of bf16 stays within 15 % of f32 while the trajectories are still correlated (first 100 steps;
    measured r3: <= 10.3 %, control <= 5.4 %),
  * afterwards (measured r3: the two PARITY runs differ by up to 0.11 in mean loss at step 175-200, i.e. by 40 %) the speed mode
    never learns slower than the slower of the two parity runs: loss_bf16 <= 1.25 x max(loss_f32, loss_control) + 0.02 in every
    window -- it behaves like one more run of the parity mode, not like a different optimisation problem,
  * both modes drive the loss to the label-smoothing floor (< 0.12) and decode the training utterances greedily (LAS.inference,
    corpus WER as test.py computes it) with WER < 5 %.
The curves of the run are written to $LAS_TRAIN_QUALITY_OUT (profiles/r3_train_quality.json)."""
import json
import os
import sys

import numpy as np
import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu

STEPS, UTTS, BATCH = 600, 64, 16


def test_bf16_training_tracks_f32_and_both_learn_the_corpus():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_quality as tq
    corpus = tq.LearnableCorpus(UTTS, 200, 8, seed=1, noise=0.3)
    args = tq.arch(lr=1e-3)
    p0 = tq.initial_weights(args)
    rng = np.random.RandomState(99)
    pp = {n: (v * (1.0 + 1e-6 * rng.randn(*v.shape))).astype(np.float32) for n, v in p0.items()}
    runs = {"f32": tq.run_mode("f32", corpus, p0, args, STEPS, BATCH), "bf16": tq.run_mode("bf16", corpus, p0, args, STEPS, BATCH),
            "f32_perturbed_1e-6": tq.run_mode("f32", corpus, pp, args, STEPS, BATCH)}
    w = {k: tq.window_means(r["loss"]) for k, r in runs.items()}
    out = os.environ.get("LAS_TRAIN_QUALITY_OUT")
    if out:
        with open(out, "w") as f:
            json.dump({"steps": STEPS, "utterances": UTTS, "batch": BATCH, "lr": 1e-3,
                       "runs": {k: {"mean_loss_per_25_steps": [round(v, 4) for v in w[k]], "greedy_wer_on_training_set": r["wer"],
                                    "exact_sentences": r["exact"]} for k, r in runs.items()}}, f, indent=1)
    f32, bf, ctl = w["f32"], w["bf16"], w["f32_perturbed_1e-6"]
    print("f32 ", ["%.3f" % v for v in f32], runs["f32"]["wer"])
    print("bf16", ["%.3f" % v for v in bf], runs["bf16"]["wer"])
    print("ctl ", ["%.3f" % v for v in ctl], runs["f32_perturbed_1e-6"]["wer"])
    for k in range(4):                                   # steps 0..99: trajectories still correlated
        assert abs(bf[k] - f32[k]) <= 0.15 * f32[k], (k, bf[k], f32[k])
    for k in range(4, len(f32)):
        assert bf[k] <= 1.25 * max(f32[k], ctl[k]) + 0.02, (k, bf[k], f32[k], ctl[k])
    for name in ("f32", "bf16"):
        assert w[name][-1] < 0.12, (name, w[name][-1])
        assert runs[name]["wer"] < 0.05, (name, runs[name]["wer"], runs[name]["examples"])
