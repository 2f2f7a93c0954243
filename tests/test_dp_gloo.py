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
