"""Data-parallel train step on the GPU: 2 ranks (one process each; gloo here because the test box has ONE GPU --
RCCL refuses two ranks on one device) vs a single process on the concatenated batch.  Same flat-bucket all-reduce,
global token normalisation and replicated clip+Adam as the multi-GPU bench path (las/parallel.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import make_args, synthetic_batch, PKG, ROOT

pytestmark = pytest.mark.gpu


def _setup(cell, prec="f32"):
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_units=64, num_enc_layers=1, dec_units=64, num_dec_layers=1, embedding_size=32, attention_size=32, lr=1e-3)
    p0 = O.init_params(args, seed=7, cell=cell)
    L.set_cell(cell); L.set_precision(prec)
    st = V.reset_default_store(device="cuda:0"); st.load(p0)
    return args, LAS(args, Listener, Speller, {}), st


def _worker(rank, world, port, out_path, prec="f32"):
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from las.parallel import DataParallel
    from las import _hip
    # two PROCESSES share the one GPU of the test box: the one-launch Speller loop needs all of its 256 workgroups co-resident
    # (one per compute unit), which two concurrent launches cannot both have -- per-step kernels here (one process per GPU, the
    # real deployment, never shares)
    _hip.speller_flags = _hip.SPELLER_NO_FUSED_STEP
    args, las, st = _setup("lstm", prec)
    las.dp = DataParallel()
    las.build_variables()
    las.dp.broadcast_(st.flat)
    calls = []
    if prec == "bf16":                                   # the early exchange must really have been taken (speed-mode pBLSTM stack)
        real = las.dp.all_reduce_
        las.dp.all_reduce_ = lambda t, async_op=False: (calls.append((t.numel(), async_op)), real(t, async_op=async_op))[1]
    losses = []
    for k in range(3):                                   # three optimiser steps, a different (lock-step) global batch each
        xs, ys = synthetic_batch(6, 40 + 8 * k, 8, 30, seed=11 + k)
        sl = slice(rank, None, world)                    # the reader's sharding rule: rows rank, rank + world, ...
        losses.append(float(las.train((xs[0][sl], xs[1][sl]), (ys[0][sl], ys[1][sl]))[0]))
    torch.cuda.synchronize()
    las.check_status()
    if prec == "bf16":
        assert len(calls) == 6 and [c[1] for c in calls] == [True, False] * 3, calls
        assert calls[0][0] + calls[1][0] == st.grad_bucket.numel() and calls[0][0] > calls[1][0]
    torch.save({"flat": st.flat.cpu(), "m": st.adam_m.cpu(), "v": st.adam_v.cpu(), "loss": losses}, out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_two_rank_steps_equal_single_rank_full_batch(tmp_path, prec):
    """three steps: token count + gradient bucket all-reduced (the bucket in two parts: everything but the bottom layer under the
    end-of-step tail, then guard + bottom layer), replicated clip + Adam -- parameters AND Adam moments equal single-process
    training on the concatenated batches, and the two replicas stay bit-identical"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, 29600 + os.getpid() % 300, out, prec), nprocs=2, join=True)
    got, got1 = torch.load(out + ".0"), torch.load(out + ".1")
    for k in ("flat", "m", "v"):
        assert torch.equal(got[k], got1[k]), k
    args, las, st = _setup("lstm", prec)
    losses = []
    for k in range(3):
        xs, ys = synthetic_batch(6, 40 + 8 * k, 8, 30, seed=11 + k)
        # the two ranks' shards as TWO stacked batches of one step (LAS.train_stacked: the single-GPU form of the same update)
        shards = [((xs[0][r::2], xs[1][r::2]), (ys[0][r::2], ys[1][r::2])) for r in range(2)]
        losses.append(float(las.train_stacked(shards)[0]))
    torch.cuda.synchronize()
    tol = 1.0 if prec == "f32" else 20.0                 # bf16: near-zero gradients whose Adam steps are sign-like, then 2 more steps
    assert max(abs(a - b) for a, b in zip(got["loss"], losses)) < 2e-4 * tol
    # Adam's first steps are ~lr*sign(g): compare the UPDATE against lr, not the weights against each other
    assert (got["flat"] - st.flat.cpu()).abs().max().item() < 6e-4 * (1 if prec == "f32" else 10)
    assert (got["m"] - st.adam_m.cpu()).abs().max().item() < 1e-5 * tol * max(1.0, st.adam_m.abs().max().item())
    assert (got["v"] - st.adam_v.cpu()).abs().max().item() < 1e-6 * tol * max(1.0, st.adam_v.abs().max().item())


_RCCL_SCRIPT = r'''
import os, sys
sys.path[:0] = [%(pkg)r, %(root)r, %(tests)r]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
from las.parallel import DataParallel
from test_gpu_dp import _setup
from helpers import synthetic_batch
args, las, st = _setup("lstm")
las.dp = DataParallel()
las.build_variables()
ref = st.flat.clone()
las.dp.broadcast_(st.flat)                                                  # RCCL broadcast of the flat parameter bucket
assert torch.equal(ref, st.flat)
g = torch.arange(st.flat.numel(), device=dev, dtype=torch.float32)
las.dp.all_reduce_(g)                                                       # RCCL all-reduce over one flat fp32 bucket
assert torch.equal(g, torch.arange(st.flat.numel(), device=dev, dtype=torch.float32))
xs, ys = synthetic_batch(6, 40, 8, 30, seed=11)
loss = float(las.train(xs, ys)[0])                                          # a full step with the collectives in it
torch.cuda.synchronize()
print("RCCL_OK %%.6f" %% loss)
dist.destroy_process_group()
'''


def test_rccl_world1_flat_bucket_all_reduce_and_train_step():
    """RCCL communicator set-up, broadcast and all-reduce of the flat gradient bucket on hardware (one rank: the test box
    has one GPU), and the data-parallel train step through them == the plain single-process step."""
    import subprocess
    script = _RCCL_SCRIPT % dict(pkg=PKG, root=ROOT, tests=os.path.join(ROOT, "tests"), port=str(29900 + os.getpid() % 90))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_OK")]
    assert line, r.stdout[-2000:]
    args, las, st = _setup("lstm")
    xs, ys = synthetic_batch(6, 40, 8, 30, seed=11)
    loss = float(las.train(xs, ys)[0])
    assert abs(float(line[0].split()[1]) - loss) < 1e-5


def _worker_lost_step(rank, world, port, out_path, lose):
    """six lock-step steps; with `lose`, rank 1's status word is set in front of step 1 (a timed-out Speller loop / sweep on ONE rank)"""
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from las.parallel import DataParallel
    from las import _hip
    _hip.speller_flags = _hip.SPELLER_NO_FUSED_STEP      # (two processes share the test box's one GPU: see _worker)
    args, las, st = _setup("lstm", "bf16")
    las.dp = DataParallel()
    las.build_variables()
    las.dp.broadcast_(st.flat)
    warnings.simplefilter("ignore")
    for k in range(6):
        if lose and k == 1 and rank == 1:
            torch.cuda.synchronize()
            _hip.status_word("cuda")[0] = 3
        xs, ys = synthetic_batch(6, 40 + 8 * k, 8, 30, seed=11 + k)
        sl = slice(rank, None, world)
        las.train((xs[0][sl], xs[1][sl]), (ys[0][sl], ys[1][sl]))
    las.check_status()
    torch.cuda.synchronize()
    torch.save({"flat": st.flat.cpu(), "m": st.adam_m.cpu(), "gs": st.global_step, "recovered": las.recovered_steps}, out_path + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_a_step_lost_on_one_rank_is_rerun_by_all_ranks_in_lock_step(tmp_path):
    """Round 6: the step recovery under data parallelism.  A time-out on ONE rank reaches every rank through the all-reduced guard slot
    (every rank's las_clip_adam skips, and keeps skipping: the word is sticky).  The ranks must also DECIDE together: each looks at the
    guard slot of the step three steps back -- the same number on every rank at the same call -- rewinds to that step and re-runs its own
    held shards with the collectives in lock step.  Result: both replicas bit-identical to each other and to an undisturbed run (at this
    size the fall-back schedule changes no arithmetic: the same per-step kernels, hand-overs not in play)."""
    import torch.multiprocessing as mp
    out = {}
    for lose in (False, True):
        path = str(tmp_path / ("dp_lost%d.pt" % lose))
        mp.spawn(_worker_lost_step, args=(2, 29300 + (os.getpid() + 17 * lose) % 300, path, lose), nprocs=2, join=True)
        out[lose] = (torch.load(path + ".0"), torch.load(path + ".1"))
    for lose in (False, True):
        a, b = out[lose]
        assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["m"], b["m"]) and a["gs"] == b["gs"] == 6
        assert a["recovered"] == b["recovered"] == (4 if lose else 0), (a["recovered"], b["recovered"])
    assert torch.equal(out[False][0]["flat"], out[True][0]["flat"]) and torch.equal(out[False][0]["m"], out[True][0]["m"])
