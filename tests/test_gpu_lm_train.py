"""F4: char RNNLM training (reference lang/char_rnn_model.py:110-244, train_lm.py) on the GPU vs the oracle restatement:
two consecutive batches (the state is carried), loss, every gradient and the Adam-updated weights; then the CLI end to end
(train_lm.py -> result.json / vocab.json / best model -> decode.py --apply_lm loads it)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import PKG

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("E", [0, 12])
def test_lm_train_step_matches_oracle(E):
    from las import layers as L, variables as V
    from lang.char_rnn_model import CharRNN
    from oracle import las_oracle as O
    Vn, H, NL, B, U = 28, 24, 2, 5, 7
    L.set_precision("f32")
    st = V.VariableStore(device="cuda", seed=3)
    lm = CharRNN(True, B, U, Vn, H, max_grad_norm=5.0, embedding_size=E, num_layers=NL, learning_rate=2e-3, store=st)
    lm.params()
    with torch.no_grad():                         # biases are zero-initialised: make them matter
        for n, v in st.vars.items():
            if n.endswith("bias") or n.endswith("softmax_b"):
                v.add_(torch.randn_like(v) * 0.1)
    def leaves():
        d = {"cells": [(st.vars["lm/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l].detach().cpu().clone().requires_grad_(True),
                        st.vars["lm/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l].detach().cpu().clone().requires_grad_(True)) for l in range(NL)],
             "softmax_w": st.vars["lm/softmax/softmax_w"].detach().cpu().clone().requires_grad_(True),
             "softmax_b": st.vars["lm/softmax/softmax_b"].detach().cpu().clone().requires_grad_(True)}
        if E > 0:
            d["embedding"] = st.vars["lm/embedding"].detach().cpu().clone().requires_grad_(True)
        return d
    names = {"embedding": "lm/embedding", "softmax_w": "lm/softmax/softmax_w", "softmax_b": "lm/softmax/softmax_b"}
    for l in range(NL):
        names["kernel%d" % l] = "lm/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l
        names["bias%d" % l] = "lm/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l
    rng = np.random.RandomState(0)
    state_g, state_o = None, [(torch.zeros(B, H), torch.zeros(B, H)) for _ in range(NL)]
    m, v = None, None
    for it in range(2):
        x = rng.randint(0, Vn, size=(B, U))
        y = rng.randint(0, Vn, size=(B, U))
        y[0, 0] = 0                                   # id 0 ('.') is a real target for the LM: it must NOT be masked like PAD
        lo_leaves = leaves()
        if m is None:
            keys = (["embedding"] if E > 0 else []) + [k for l in range(NL) for k in ("kernel%d" % l, "bias%d" % l)] + ["softmax_w", "softmax_b"]
            src = {"embedding": lo_leaves.get("embedding"), "softmax_w": lo_leaves["softmax_w"], "softmax_b": lo_leaves["softmax_b"]}
            for l in range(NL):
                src["kernel%d" % l], src["bias%d" % l] = lo_leaves["cells"][l]
            m = {k: torch.zeros_like(src[k]) for k in keys}
            v = {k: torch.zeros_like(src[k]) for k in keys}
        loss_o, state_o, g_o, newp, m, v = O.lm_train_step(lo_leaves, torch.tensor(x), torch.tensor(y), state_o, 2e-3, it + 1, m, v)
        loss, state_g = lm.train_step(x, y, state_g)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(loss_o)) < 2e-5
        for k, n in names.items():
            if k not in g_o:
                continue
            go, gg = g_o[k], st.vars[n].grad.cpu()
            assert (gg - go).abs().max().item() / max(go.abs().max().item(), 1e-3) < 2e-3, (it, k)
            assert (st.vars[n].detach().cpu() - newp[k]).abs().max().item() < 2e-5, (it, k)
        for l in range(NL):
            assert (state_g[l][1].cpu() - state_o[l][1]).abs().max().item() < 1e-5
    # evaluation mode (no update) returns the same loss as the oracle forward and leaves the weights alone
    before = st.flat.clone()
    l_eval, _ = lm.train_step(x, y, None, train=False)
    assert torch.equal(before, st.flat) and float(l_eval) > 0


def test_train_lm_cli_and_decode_with_the_trained_lm(tmp_path):
    tmp = str(tmp_path)
    rng = np.random.RandomState(0)
    words = ["THE", "CAT", "SAT", "ON", "A", "MAT", "AND", "RAN", "FAR", "AWAY"]
    text = ". ".join(" ".join(words[i] for i in rng.randint(0, len(words), 8)) for _ in range(60))
    data = os.path.join(tmp, "corpus.txt")
    open(data, "w").write(text.lower() + "!")
    out = os.path.join(tmp, "lm_out")
    cmd = [sys.executable, os.path.join(PKG, "train_lm.py"), "--data_file", data, "--output_dir", out, "--num_epochs", "3",
           "--hidden_size", "32", "--num_layers", "2", "--batch_size", "8", "--num_unrollings", "6", "--learning_rate", "0.01"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(os.path.join(out, "result.json")))
    assert res["params"]["vocab_size"] == 28 and os.path.exists(res["best_model"]) and os.path.exists(os.path.join(out, "vocab.json"))
    assert res["best_valid_ppl"] < 27.0 and res["test_ppl"] < 28.0          # below uniform over 28 symbols after 3 tiny epochs
    ppl = [float(l.split("Perplexity: ")[1].split(",")[0]) for l in r.stdout.splitlines() if "Perplexity:" in l]
    assert ppl[0] > ppl[-3]                                                  # training perplexity went down
    # decode.py --apply_lm picks the LM up from --lm_dir (reference decode.py:27-53,68-75)
    cmd = [sys.executable, os.path.join(PKG, "decode.py"), "--unit", "char", "--feat_dim", "13", "--enc_type", "pblstm", "--enc_units", "64",
           "--num_enc_layers", "2", "--dec_units", "64", "--num_dec_layers", "1", "--attention_size", "32", "--embedding_size", "32",
           "--cell", "lstm", "--synthetic", "True", "--save_dir", os.path.join(tmp, "nomodel"), "--beam_size", "4", "--max_steps", "3",
           "--apply_lm", "True", "--lm_dir", out + "/", "--lm_weight", "0.3", "--decode_batch", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Rnnlm restored" in r.stdout and "Dev WER:" in r.stdout
