"""The input side of the step loop on the GPU (VERDICT r2 Missing #3; reference tfrecord_data_loader.py:87-105 prefetch, train.py:114-117):
DeviceFeeder must deliver exactly the source's batches -- through its pinned ring, the copy stream and the device ring with slot
reuse -- for the C++ TFRecord reader (las_input_upload straight out of its pinned slots) and for a Python source; the lagged loss
log must report every step's loss; `train.py --synthetic` runs on it."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import PKG

pytestmark = pytest.mark.gpu


def _files(tmp_path, n_files=3, per=60):
    import tfrecord_data_loader as tdl
    rng = np.random.RandomState(0)
    files = []
    for i in range(n_files):
        lens = rng.randint(100, 1500, size=per)
        feats = [rng.randn(n, 13, 3).astype(np.float32) for n in lens]
        toks = [np.r_[rng.randint(3, 30, size=rng.randint(2, 150)), 2] for _ in lens]
        fn = str(tmp_path / ("train-%d.tfrecord" % i))
        tdl.write_tfrecord(fn, feats, toks)
        files.append(fn)
    return files


@pytest.mark.parametrize("native", [True, False])
def test_feeder_delivers_the_sources_batches(tmp_path, native):
    import tfrecord_data_loader as tdl
    from las.input_pipeline import feeder_for
    files = _files(tmp_path)
    ref = tdl._BucketedIterator(files, tdl.data_parser, 13, True, seed=4, shuffle_buffer=2)
    src = tdl.NativeReader(files, 13, True, seed=4, shuffle_buffer=2) if native else \
        tdl._BucketedIterator(files, tdl.data_parser, 13, True, seed=4, shuffle_buffer=2)
    feed = feeder_for(src, "cuda:0", 13, depth=2)          # 2 device slots, 9 batches: every slot is reused several times
    junk = torch.empty(64 << 20, device="cuda")
    for k in range(9):
        (audio, audiolen), (y, tokenlen) = next(feed)
        (ra, ral), (ry, rtl) = next(ref)
        assert audio.is_cuda and y.is_cuda and audio.dtype == torch.float32 and y.dtype == torch.int32
        junk.normal_()                                      # work on the training stream between the batches
        assert np.array_equal(audio.cpu().numpy(), ra), k
        assert np.array_equal(y.cpu().numpy(), ry) and np.array_equal(audiolen, ral) and np.array_equal(tokenlen, rtl)
    feed.close()


def test_lagged_log_reports_every_step_in_order():
    from las.input_pipeline import LaggedLog
    got = []
    log = LaggedLog(lambda info, v: got.append((info, v)))
    for i in range(20):
        log.push(i, torch.full((1,), float(i), device="cuda") * 2)
    log.drain(True)
    assert got == [(i, 2.0 * i) for i in range(20)]


def test_train_py_synthetic_runs_on_the_feeder(tmp_path):
    cmd = [sys.executable, os.path.join(PKG, "train.py"), "--unit", "char", "--feat_dim", "13", "--enc_type", "pblstm",
           "--enc_units", "64", "--num_enc_layers", "2", "--dec_units", "128", "--num_dec_layers", "1", "--attention_size", "64",
           "--embedding_size", "32", "--dropout_rate", "0", "--cell", "lstm", "--dtype", "bf16", "--synthetic", "True",
           "--max_steps", "4", "--save_dir", str(tmp_path / "model"), "--log_dir", str(tmp_path / "log")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(1, 5):
        assert "Step: %d," % k in r.stdout
    assert "train loop: 4 steps" in r.stdout and "utterances/s" in r.stdout
