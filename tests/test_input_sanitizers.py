"""Host sanitizers on the native input reader (sanitizer row; GPU AddressSanitizer is not available on the pool,
so this is the CPU build only): csrc/input.hip compiled host-only with -fsanitize=address,undefined together with
tests/native/input_asan_main.cpp, run over TFRecord files written here -- threads, mmap, protobuf parsing, slot ring, close with
work in flight, lock-step shards, an evaluation pass, a corrupted file."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import PKG, ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_input_reader_under_address_and_ub_sanitizers(tmp_path):
    sys.path.insert(0, PKG)
    import tfrecord_data_loader as tdl
    exe = str(tmp_path / "input_asan")
    cmd = [HIPCC, "--offload-host-only", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-Wno-unused-result", os.path.join(PKG, "csrc", "input.hip"),
           os.path.join(ROOT, "tests", "native", "input_asan_main.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.RandomState(0)
    files = []
    for i in range(4):
        lens = rng.randint(60, 1700, size=30)
        feats = [rng.randn(n, 13, 3).astype(np.float32) for n in lens]
        toks = [np.r_[rng.randint(3, 30, size=rng.randint(1, 150)), 2] for _ in lens]
        fn = str(tmp_path / ("train-%d.tfrecord" % i))
        tdl.write_tfrecord(fn, feats, toks)
        files.append(fn)
    bad = str(tmp_path / "bad.tfrecord")
    raw = bytearray(open(files[0], "rb").read())
    raw[len(raw) // 2] ^= 0x5A                              # somewhere inside a record: framing or payload
    raw[3] ^= 1                                             # and the first length field for certain
    open(bad, "wb").write(raw)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([exe, bad] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "INPUT_SANITIZER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
