"""Data-parallel path on CPU: world_size 2 over gloo.  Each rank computes the oracle's gradients on its own
utterance shard, normalised by the GLOBAL non-PAD token count, and all-reduces one flat bucket through
las.parallel.DataParallel; the result must equal single-process training on the concatenated batch
(the reference divides by the batch's own token count, las/las.py:329-331)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from oracle import las_oracle as O


def _grads(p, xs, ys, args, n_total):
    x = torch.tensor(xs[0]).reshape(len(xs[1]), -1, 39)
    U = int(ys[1].max())
    h, el = O.pblstm_listener(x, xs[1], p, args.num_enc_layers, "lstm")
    logits, _ = O.speller_forward(h, el, U, p, args, "lstm", teacher=torch.tensor(ys[0]))
    y = torch.tensor(ys[0])[:, :U].long()
    y1 = O.label_smoothing(torch.nn.functional.one_hot(y, 30).float())
    ce = -(y1 * torch.log_softmax(logits, -1)).sum(-1)
    loss = (ce * (y != 0)).sum() / (n_total + 1e-9)
    names = sorted(p)
    g = torch.autograd.grad(loss, [p[n] for n in names])
    return loss.detach(), torch.cat([t.reshape(-1) for t in g])


def _worker(rank, world, port, ret):
    sys.path.insert(0, helpers.PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from las.parallel import DataParallel, shard
    dp = DataParallel()
    args = helpers.make_args(enc_units=8, num_enc_layers=1, dec_units=8, num_dec_layers=1, embedding_size=6, attention_size=8)
    xs, ys = helpers.synthetic_batch(4, 10, 5, 30, seed=2)
    p = O.to_torch(O.init_params(args, seed=1, cell="lstm"), requires_grad=True)
    sl = slice(rank * 2, rank * 2 + 2)
    U = int(ys[1].max())                      # same dec_steps on both ranks (same bucket)
    ys_l = (ys[0][sl], np.array([U, U]))
    n_local = torch.tensor(float((ys[0][sl][:, :U] != 0).sum()))
    n_total = dp.all_reduce_scalar(n_local)
    loss, flat = _grads(p, (xs[0][sl], xs[1][sl]), ys_l, args, n_total)
    dp.all_reduce_(flat)
    loss = dp.all_reduce_scalar(loss)
    assert shard(list(range(5)), rank, world) == ([0, 2, 4] if rank == 0 else [1, 3])
    if rank == 0:
        ret["flat"], ret["loss"], ret["n"] = flat.numpy(), float(loss), float(n_total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_rank_full_batch():
    port = 29500 + (os.getpid() % 1000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    args = helpers.make_args(enc_units=8, num_enc_layers=1, dec_units=8, num_dec_layers=1, embedding_size=6, attention_size=8)
    xs, ys = helpers.synthetic_batch(4, 10, 5, 30, seed=2)
    p = O.to_torch(O.init_params(args, seed=1, cell="lstm"), requires_grad=True)
    U = int(ys[1].max())
    n = float((ys[0][:, :U] != 0).sum())
    loss, flat = _grads(p, xs, (ys[0], np.array([U] * 4)), args, torch.tensor(n))
    assert ret["n"] == n
    assert abs(ret["loss"] - float(loss)) < 1e-6
    assert np.abs(ret["flat"] - flat.numpy()).max() < 1e-6


def _eval_worker(rank, world, port, ret):
    sys.path.insert(0, helpers.PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LAS_DIST_BACKEND="gloo")
    from las import parallel
    dp = parallel.init_from_env()
    assert dp is not None and dp.rank == rank and dp.world == world
    utts = list(range(11))                                   # length-sorted utterance indices (decode.py:122-124)
    mine = parallel.shard(utts, rank, world)
    errors, words = float(sum(mine)), float(10 * len(mine))  # stand-ins for this replica's edit distances / word counts
    e, n = parallel.reduce_error_counts(dp, errors, words)
    ret[rank] = (mine, e, n)
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_only_evaluation_shards_and_reduces_counts():
    """SURVEY 8(e) decode / eval: replicas only -- round-robin shard of the utterance list, (errors, words) all-reduced."""
    port = 29700 + (os.getpid() % 200)
    ret = mp.Manager().dict()
    mp.spawn(_eval_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0][0] == [0, 2, 4, 6, 8, 10] and ret[1][0] == [1, 3, 5, 7, 9]
    assert ret[0][1:] == ret[1][1:] == (55.0, 110.0)


# ---- lock-step data parallelism over the TFRecord pipeline, three optimiser steps ------------------------
def _tiny_pipeline(tdl):
    tdl.TRAIN_BOUNDARIES[:] = [12, 24]
    tdl.BUCKET_BATCH_LIMIT[:] = [3, 2, 2]


def _tiny_corpus(d):
    sys.path.insert(0, helpers.PKG)
    import tfrecord_data_loader as tdl
    rng = np.random.RandomState(11)
    files = []
    for i in range(3):
        lens = rng.randint(4, 24, size=10)
        feats = [rng.randn(n, 13, 3).astype(np.float32) for n in lens]
        toks = [np.r_[rng.randint(3, 30, size=rng.randint(1, 4)), 2] for _ in lens]
        fn = os.path.join(d, "train-%d.tfrecord" % i)
        tdl.write_tfrecord(fn, feats, toks)
        files.append(fn)
    return files


def _adam_steps(p, batches, args, dp=None, steps=3):
    """`steps` optimiser steps of the oracle (clip + TF Adam) on this process' batches; under dp the token count and the flat
    gradient are all-reduced exactly as LAS.train does (las/parallel.py)."""
    names = sorted(p)
    m = {n: torch.zeros_like(p[n]) for n in names}
    v = {n: torch.zeros_like(p[n]) for n in names}
    Ts = []
    for step in range(steps):
        (audio, audiolen), (y, tokenlen) = next(batches)
        Ts.append(audio.shape[1])
        U = int(tokenlen.max())
        if dp is not None:                                   # same dec_steps on every rank: the global batch's longest transcript
            U = int(_max_all(dp, U))
        n_local = torch.tensor(float((y[:, :U] != 0).sum()))
        n_total = dp.all_reduce_scalar(n_local) if dp is not None else n_local
        pl = {n: p[n].detach().clone().requires_grad_(True) for n in names}
        _, flat = _grads(pl, (audio, audiolen), (y, np.full(len(tokenlen), U)), args, n_total)
        if dp is not None:
            dp.all_reduce_(flat)
        gs, o = [], 0
        for n in names:
            gs.append(flat[o:o + p[n].numel()].view(p[n].shape))
            o += p[n].numel()
        gs, _ = O.clip_by_global_norm(gs, 5.0)
        for n, g in zip(names, gs):
            p[n], m[n], v[n] = O.adam_tf(p[n].detach(), g, m[n], v[n], step + 1, 1e-3)
    cat = lambda d: torch.cat([d[n].reshape(-1) for n in names])
    return Ts, cat(p), cat(m), cat(v)


def _max_all(dp, x):
    t = torch.tensor([float(x)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=dp.group)
    return float(t[0])


def _lockstep_worker(rank, world, port, d, ret):
    sys.path.insert(0, helpers.PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tfrecord_data_loader as tdl
    from las.parallel import DataParallel
    _tiny_pipeline(tdl)
    files = sorted(os.path.join(d, f) for f in os.listdir(d))
    it = tdl._BucketedIterator(files, tdl.data_parser, 13, True, seed=3, shuffle_buffer=2, cycle_length=2, rank=rank, world=world)
    args = helpers.make_args(enc_units=8, num_enc_layers=1, dec_units=8, num_dec_layers=1, embedding_size=6, attention_size=8)
    p = O.to_torch(O.init_params(args, seed=1, cell="lstm"))
    Ts, th, m, v = _adam_steps(p, it, args, DataParallel())
    ret[rank] = (Ts, th.numpy(), m.numpy(), v.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_lock_step_two_ranks_train_like_one_process_on_the_global_batches(tmp_path):
    """Both ranks draw the same bucket (same T) at every step, and three clip+Adam steps on the sharded global batches -- token
    count and gradient bucket all-reduced -- leave parameters AND Adam moments equal to single-process training on the same
    global batches (per-bucket limits x 2)."""
    d = str(tmp_path)
    files = _tiny_corpus(d)
    ret = mp.Manager().dict()
    mp.spawn(_lockstep_worker, args=(2, 29800 + (os.getpid() % 150), d, ret), nprocs=2, join=True)
    assert ret[0][0] == ret[1][0] and len(ret[0][0]) == 3            # equal T per step across the ranks
    for a, b in zip(ret[0][1:], ret[1][1:]):
        assert np.array_equal(a, b)                                   # replicas stay identical (Adam state included)
    import tfrecord_data_loader as tdl
    saved = (list(tdl.TRAIN_BOUNDARIES), list(tdl.BUCKET_BATCH_LIMIT))
    try:
        _tiny_pipeline(tdl)
        tdl.BUCKET_BATCH_LIMIT[:] = [2 * x for x in tdl.BUCKET_BATCH_LIMIT]
        it = tdl._BucketedIterator(files, tdl.data_parser, 13, True, seed=3, shuffle_buffer=2, cycle_length=2)
        args = helpers.make_args(enc_units=8, num_enc_layers=1, dec_units=8, num_dec_layers=1, embedding_size=6, attention_size=8)
        p = O.to_torch(O.init_params(args, seed=1, cell="lstm"))
        Ts, th, m, v = _adam_steps(p, it, args, None)
    finally:
        tdl.TRAIN_BOUNDARIES[:], tdl.BUCKET_BATCH_LIMIT[:] = saved
    assert Ts == ret[0][0]
    # first Adam steps are ~ lr * sign(g): compare the accumulated moments tightly, the parameters against the step size
    assert np.abs(ret[0][2] - m.numpy()).max() < 1e-6 and np.abs(ret[0][3] - v.numpy()).max() < 1e-8
    assert np.abs(ret[0][1] - th.numpy()).max() < 2e-4
