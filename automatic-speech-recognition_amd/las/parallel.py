"""las.parallel -- data-parallel exchange for the train step.

The reference is single-device (train.py:23).  Here every rank (one process per GPU) runs the same
step on its own utterance shard; the ONLY exchanges are
  * a 1-float all-reduce of the non-PAD token count before backward (so the loss/gradient is
    normalised by the GLOBAL count, which makes N-rank training equal single-rank training on the
    concatenated batch: the reference divides by the batch's own count, las/las.py:329-331), and
  * ONE all-reduce(sum) over the flat fp32 gradient bucket (RCCL over xGMI; backend 'nccl' on ROCm,
    'gloo' in the CPU tests).
Clip + Adam then run replicated on every rank from identical inputs."""
import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    # The collectives are issued for every world size, 1 included: a one-rank group still goes through the backend
    # (RCCL on a GPU box), which is how the single-GPU test box exercises communicator set-up and the bucket exchange.
    def all_reduce_(self, flat, async_op=False):
        """in-place sum over the ranks; async_op=True returns the work handle (the backend's own stream runs it; `wait()` orders
        the current stream behind it)"""
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return work if async_op else flat

    def all_reduce_scalar(self, x):
        x = x.detach().clone().reshape(1)
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x[0]

    def broadcast_(self, flat, src=0):
        dist.broadcast(flat, src=src, group=self.group)
        return flat

    def sampling_seed(self, global_step):
        """Seed of this rank's scheduled-sampling draws at a step: distinct per rank (only the coin is
        shared), reproducible for a given (step, rank)."""
        return sampling_seed(global_step, self.rank)


def sampling_seed(global_step, rank=0):
    return (977 + int(global_step)) * 1000003 + 7919 * int(rank)


def shard(items, rank, world):
    """Round-robin shard of a (length-sorted) utterance list -- replicas-only decode/eval."""
    return items[rank::world]


def init_from_env(device=None):
    """One process per GPU under torch.distributed.run: initialise RCCL from RANK / WORLD_SIZE / MASTER_* and return a
    DataParallel, or None for a plain single-process run (the evaluation entry points use it as 'replicas only')."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("LAS_DIST_BACKEND", "nccl")               # "nccl" IS RCCL on ROCm
    if not dist.is_initialized():
        if backend == "nccl" and device is not None:
            dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world)
    return DataParallel()


def reduce_error_counts(dp, errors, words, device=None):
    """(errors, reference words) summed over the evaluation replicas -> corpus WER is the same on every rank."""
    if dp is None:
        return errors, words
    t = torch.tensor([float(errors), float(words)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=dp.group)
    return float(t[0]), float(t[1])
