"""End-to-end entry points on the GPU with real TFRecord input (SURVEY 3.1-3.3 call stacks + 8(f) rows F1/F3):
create_tfrecords -> train.py (2 steps, checkpoint) -> train.py resume -> test.py (greedy, WER artefacts) -> decode.py (beam)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "automatic-speech-recognition_amd")
sys.path.insert(0, PKG)

pytestmark = pytest.mark.gpu


def _run(script, extra, tmp):
    cmd = [sys.executable, os.path.join(PKG, script), "--unit", "char", "--feat_dim", "13", "--enc_type", "pblstm",
           "--enc_units", "64", "--num_enc_layers", "2", "--dec_units", "128", "--num_dec_layers", "2",
           "--attention_size", "64", "--embedding_size", "32", "--dropout_rate", "0", "--cell", "lstm",
           "--tfrecord_dir", os.path.join(tmp, "rec"), "--save_dir", os.path.join(tmp, "model"),
           "--log_dir", os.path.join(tmp, "log"), "--feat_dir", os.path.join(tmp, "nofeats")] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_train_test_decode_on_tfrecords(tmp_path):
    import tfrecord_data_loader as tdl
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, "rec"))
    rng = np.random.RandomState(0)
    lens = rng.randint(40, 120, size=10)
    feats = [rng.randn(n, 13, 3).astype(np.float32) for n in lens]
    toks = [np.r_[rng.randint(3, 30, size=max(2, n // 12)), 2].astype(np.int64) for n in lens]
    tdl.create_tfrecords(feats, toks, os.path.join(tmp, "rec", "train-100"), num_files=2)
    tdl.create_tfrecords(feats[:4], toks[:4], os.path.join(tmp, "rec", "dev"), num_files=1)

    out = _run("train.py", ["--max_steps", "2"], tmp)
    assert "Step: 1," in out and "Step: 2," in out
    assert os.path.exists(os.path.join(tmp, "model", "las_E1"))
    out = _run("train.py", ["--max_steps", "1"], tmp)                   # resumes: global step continues
    assert "Step: 3," in out

    out = _run("train.py", ["--max_steps", "1", "--stack", "2"], tmp)   # two bucket batches per step (here: the whole corpus in one)
    assert "Step: 4," in out

    out = _run("test.py", [], tmp)
    assert "total utterances: 4" in out
    assert len(open(os.path.join(tmp, "log", "test_gt.txt")).read().split("\n")) == 4

    out = _run("decode.py", ["--beam_size", "4", "--max_steps", "2"], tmp)
    assert "LAS restored" in out and "Dev WER:" in out
