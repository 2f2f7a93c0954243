"""Variable store: the eager stand-in for TF-1.13 variable scopes (`tf.get_variable`, AUTO_REUSE).

The reference creates its weights lazily by name the first time a layer runs
(reference las/layers.py:68,80,246; las/las.py:73,146,202) and shares them through
`reuse=tf.AUTO_REUSE`.  Here a `VariableStore` maps the same TF names
(SURVEY.md section 8(a) parameter inventory) to fp32 torch tensors.  Before the first optimiser step
the store is *flattened*: every parameter becomes a view into ONE contiguous fp32 bucket, with
matching flat buckets for the gradient and the Adam slots, so that

  * the data-parallel exchange is a single RCCL all-reduce over one buffer,
  * clip_by_global_norm + Adam is one K9 kernel pair over one buffer (include/las_hip.h).
"""
import math

import numpy as np
import torch

_default_store = None


class VariableStore:
    def __init__(self, device=None, seed=0):
        self.device = torch.device(device) if device is not None else None
        self.vars = {}            # name -> tensor (leaf, requires_grad)
        self.buffers = {}         # name -> tensor (non-trainable state: batch-norm moving statistics)
        self.order = []           # creation order
        self.rng = np.random.RandomState(seed)
        self.flat = self.flat_grad = self.adam_m = self.adam_v = self.grad_bucket = self.guard = None
        self.global_step = 0
        self.shadows = {}         # bf16 copies of weights for the speed-mode products (las.layers._shadow); cleared when weights change
        self.shadow_recipes = {}  # key -> (tensor, ShadowDesc): how to rebuild every shadow in ONE launch (cleared when storage moves)
        self.shadow_table = None
        self.seq_recipes = {}     # (W_hh pointers, cell, H, pass) -> how to prepare that sweep's workspace (las.layers._prepare_sweeps)
        self.seq_ready = {}       # ... -> batch rows it has been prepared for since the weights last changed (consumed by ONE sweep)
        self.seq_prep_done = None  # event behind the side-stream prepare launch until the launch stream has waited for it
        self.weights_epoch = 0    # counts weights_changed() calls: an ordering event is only good for the epoch it was recorded in
        self.applied = None       # device int: optimiser updates the device really applied (las_clip_adam skips a step whose status word is set)
        self.adam_launches = []   # global_step of the las_clip_adam launches number adam_base, adam_base + 1, ... (bounded: LAS._apply_adam)
        self.adam_base = 0

    def weights_changed(self, storage_moved=False):
        """Everything derived from the parameter VALUES is stale (optimiser step, load); storage_moved: their addresses too (flatten)."""
        self.weights_epoch += 1
        self.shadows.clear()
        self.seq_ready.clear()
        if self.seq_prep_done is not None and storage_moved:
            # a side-stream prepare may still be writing the workspaces that are dropped below: the launch stream (whose later
            # allocations may reuse those blocks) waits for it first
            torch.cuda.current_stream().wait_event(self.seq_prep_done)
        self.seq_prep_done = None
        if storage_moved:
            self.shadow_recipes.clear()
            self.shadow_table = None
            self.seq_recipes.clear()

    # ---- creation ----------------------------------------------------------------------------
    def get(self, name, shape=None, init="glorot", fan=None):
        """tf.get_variable with AUTO_REUSE.  init: 'glorot' (tf default / tf.layers kernels,
        SURVEY App. A.3), 'zeros', 'uniform1' (U(-1,1): las/las.py:206, las/layers.py:248), or a
        callable(rng, shape) -> ndarray."""
        v = self.vars.get(name)
        if v is not None:
            if shape is not None and tuple(v.shape) != tuple(shape):
                raise ValueError("variable %s exists with shape %s, requested %s" % (name, tuple(v.shape), tuple(shape)))
            return v
        if shape is None:
            raise KeyError("variable %s does not exist" % name)
        if self.flat is not None:
            raise RuntimeError("variable %s requested after the store was flattened" % name)
        shape = tuple(int(s) for s in shape)
        if init == "glorot":
            fi, fo = fan if fan is not None else (shape[0], shape[-1])
            lim = math.sqrt(6.0 / (fi + fo))
            arr = self.rng.uniform(-lim, lim, size=shape)
        elif init == "zeros":
            arr = np.zeros(shape)
        elif init == "uniform1":
            arr = self.rng.uniform(-1.0, 1.0, size=shape)
        elif callable(init):
            arr = init(self.rng, shape)
        else:
            raise ValueError(init)
        t = torch.tensor(np.asarray(arr, np.float32), device=self.device)
        t.requires_grad_(True)
        self.vars[name] = t
        self.order.append(name)
        return t

    def get_buffer(self, name, shape, fill=0.0):
        b = self.buffers.get(name)
        if b is None:
            b = torch.full(tuple(shape), float(fill), device=self.device)
            self.buffers[name] = b
        return b

    def load(self, params):
        """Install externally supplied values {name: array} (parity tests, checkpoints)."""
        self.weights_changed(storage_moved=True)
        for name, val in params.items():
            val = torch.as_tensor(np.asarray(val, np.float32) if not torch.is_tensor(val) else val,
                                  dtype=torch.float32, device=self.device)
            if "moving_" in name:                      # batch-norm statistics are buffers, not parameters
                self.buffers[name] = val.clone()
                continue
            if name in self.vars:
                with torch.no_grad():
                    self.vars[name].copy_(val)
            else:
                if self.flat is not None:
                    raise RuntimeError("cannot add %s after flatten()" % name)
                t = val.clone().contiguous()
                t.requires_grad_(True)
                self.vars[name] = t
                self.order.append(name)

    # ---- flat buckets ------------------------------------------------------------------------
    def flatten(self):
        if self.flat is not None:
            return
        self.weights_changed(storage_moved=True)
        names = list(self.order)
        sizes = [self.vars[n].numel() for n in names]
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += (s + 3) // 4 * 4          # keep every view 16-byte aligned
        total = max(o, 4)
        dev = self.vars[names[0]].device if names else self.device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # the gradient bucket carries ONE extra slot IN FRONT of the gradients (4 floats, keeps 16-byte granularity): the
        # data-parallel exchange all-reduces `grad_bucket`, so a rank whose sweep timed out tells every rank to skip the update
        # (guard != 0).  It sits next to the bottom layer's gradients -- the part of the bucket that is exchanged LAST.
        self.grad_bucket = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        self.guard = self.grad_bucket[0:1]
        self.flat_grad = self.grad_bucket[4:]
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        self.offsets = dict(zip(names, offs))
        for n, off, sz in zip(names, offs, sizes):
            old = self.vars[n]
            view = self.flat[off:off + sz].view(old.shape)
            with torch.no_grad():
                view.copy_(old)
            # re-point the existing leaf (identity preserved: layers may hold references)
            old.data = view
            old.grad = self.flat_grad[off:off + sz].view(old.shape)

    def zero_grad(self):
        if self.flat_grad is not None:
            self.grad_bucket.zero_()
            for n in self.order:                     # autograd may have replaced .grad
                off, v = self.offsets[n], self.vars[n]
                g = self.flat_grad[off:off + v.numel()].view(v.shape)
                if v.grad is None or v.grad.data_ptr() != g.data_ptr():
                    v.grad = g
        else:
            for v in self.vars.values():
                v.grad = None

    def num_params(self):
        return int(sum(v.numel() for v in self.vars.values()))

    def state_dict(self):
        sd = {"params": {n: self.vars[n].detach().cpu() for n in self.order}, "global_step": self.global_step,
              "buffers": {n: b.detach().cpu() for n, b in self.buffers.items()}}
        if self.flat is not None:
            sd["adam_m"] = {n: self.adam_m[self.offsets[n]:self.offsets[n] + self.vars[n].numel()].view(self.vars[n].shape).cpu()
                            for n in self.order}
            sd["adam_v"] = {n: self.adam_v[self.offsets[n]:self.offsets[n] + self.vars[n].numel()].view(self.vars[n].shape).cpu()
                            for n in self.order}
        return sd

    def load_state_dict(self, sd):
        self.load(sd["params"])
        for n, b in sd.get("buffers", {}).items():
            self.buffers[n] = b.to(self.device)
        self.global_step = int(sd.get("global_step", 0))
        if "adam_m" in sd:
            self.flatten()
            with torch.no_grad():
                for n in self.order:
                    off, k = self.offsets[n], self.vars[n].numel()
                    self.adam_m[off:off + k].copy_(sd["adam_m"][n].reshape(-1))
                    self.adam_v[off:off + k].copy_(sd["adam_v"][n].reshape(-1))


def default_store():
    global _default_store
    if _default_store is None:
        _default_store = VariableStore()
    return _default_store


def reset_default_store(device=None, seed=0):
    """tf.reset_default_graph() analogue."""
    global _default_store
    _default_store = VariableStore(device=device, seed=seed)
    return _default_store
