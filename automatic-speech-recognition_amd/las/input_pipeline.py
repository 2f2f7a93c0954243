"""las.input_pipeline -- keeps the train loop fed without ever making it wait for the host.

The reference's loop `sess.run`s a tf.data iterator with prefetch (tfrecord_data_loader.py:87-105, train.py:114-117): reading,
parsing, bucketing and the host->device copy all happen on TensorFlow's runtime threads while the previous step computes.
`DeviceFeeder` is that stage here: ONE background thread takes batches from a source, stages them in pinned host memory, copies
them to a small ring of device buffers on a dedicated copy stream, and hands the consumer device tensors whose copy is ordered
before their first use by an event -- the train step never sees pageable memory, a synchronous copy or an idle device.

Sources:
  * `tfrecord_data_loader.NativeReader` -- batches already sit in pinned slots of the C++ reader (csrc/input.hip); the copy is
    `las_input_upload` straight out of the slot, the host never touches the payload twice;
  * any Python iterator of ((audio, audiolen), (y, tokenlen)) numpy batches (`data.SyntheticBatches`, the pure-Python TFRecord
    iterator): staged through a pinned ring owned by the feeder.

Lifetime of a device slot: batch i lives in slot i % depth; the consumer's NEXT `__next__` call records an event on the training
stream ("everything that read batch i has been enqueued before this point"); the copy stream waits for that event before it
overwrites the slot `depth` batches later.
"""
import queue
import threading

import numpy as np
import torch


class DeviceFeeder:
    def __init__(self, source, device, feat_dim, max_batch_floats, max_batch, max_tokenlen, depth=3):
        self.source, self.dev, self.depth = source, torch.device(device), int(depth)
        self.native = hasattr(source, "next_slot")
        self.F = int(feat_dim)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.d_feat = [torch.empty(max_batch_floats, dtype=torch.float32, device=self.dev) for _ in range(self.depth)]
        self.d_tok = [torch.empty(max_batch * max_tokenlen, dtype=torch.int32, device=self.dev) for _ in range(self.depth)]
        if not self.native:
            self.h_feat = [torch.empty(max_batch_floats, dtype=torch.float32).pin_memory() for _ in range(self.depth)]
            self.h_tok = [torch.empty(max_batch * max_tokenlen, dtype=torch.int32).pin_memory() for _ in range(self.depth)]
        self.ready = [torch.cuda.Event() for _ in range(self.depth)]          # copy of the slot's current batch has executed
        self.consumed = [None] * self.depth                                  # training stream is past the slot's previous batch
        self.free = [threading.Semaphore(1) for _ in range(self.depth)]
        self.q = queue.Queue(maxsize=self.depth)
        self._prev = None
        self._stop = False
        self._err = None
        self._th = threading.Thread(target=self._produce, name="las-feeder", daemon=True)
        self._th.start()

    # ---- producer thread ---------------------------------------------------------------------------------------------------
    def _produce(self):
        try:
            torch.cuda.set_device(self.dev)
            i = 0
            while not self._stop:
                s = i % self.depth
                self.free[s].acquire()                                      # the consumer has moved past the slot's previous batch
                if self._stop:
                    return
                if self.native:
                    b = self.source.next_slot()
                    if b is None:
                        self.q.put(None)
                        return
                    B, T, Ut = b.B, b.T, b.max_tokenlen
                    (_, fl), (_, tl) = self.source.arrays(b)
                    audiolen, tokenlen = fl.astype(np.int32), tl.astype(np.int32)     # (copies: the slot is released below)
                else:
                    try:
                        (audio, audiolen), (y, tokenlen) = next(self.source)
                    except StopIteration:
                        self.q.put(None)
                        return
                    audio = np.ascontiguousarray(audio, np.float32)
                    y = np.ascontiguousarray(y, np.int32)
                    B, T, Ut = audio.shape[0], audio.shape[1], y.shape[1]
                    np.copyto(self.h_feat[s].numpy()[:audio.size], audio.reshape(-1))          # (releases the GIL for the memcpy)
                    np.copyto(self.h_tok[s].numpy()[:y.size], y.reshape(-1))
                    audiolen, tokenlen = np.asarray(audiolen, np.int32).copy(), np.asarray(tokenlen, np.int32).copy()
                nf, nt = B * T * self.F * 3, B * Ut
                # The step that last read this device slot must be done before the copy overwrites it.  THIS THREAD waits for that
                # (two steps back with depth 3: normally no wait at all), not the copy stream: r4 -- with the copy stream on a hardware
                # queue of its own (GPU_MAX_HW_QUEUES = 8, or the auxiliary streams in the high-priority pool) a device-side
                # `copy_stream.wait_event(consumed)` in front of the DMA stretched EVERY launch of the training thread (loop_trace.py:
                # 14.0 -> 17.5 ms per step, each phase longer, clip + Adam 0.06 -> 0.19 ms); a copy whose dependencies are already
                # met when it is enqueued goes to the DMA engine directly.
                if self.consumed[s] is not None:
                    self.consumed[s].synchronize()
                with torch.cuda.stream(self.copy_stream):
                    if self.native:
                        self.source.upload(b, self.d_feat[s].data_ptr(), self.d_tok[s].data_ptr(), self.copy_stream.cuda_stream)
                        self.source.release(b)                              # (the reader waits for the copy before it refills the slot)
                    else:
                        self.d_feat[s][:nf].copy_(self.h_feat[s][:nf], non_blocking=True)
                        self.d_tok[s][:nt].copy_(self.h_tok[s][:nt], non_blocking=True)
                    self.ready[s].record(self.copy_stream)
                self.q.put((s, B, T, Ut, audiolen, tokenlen))
                i += 1
        except BaseException as e:                                          # surfaces in the consumer
            self._err = e
            self.q.put(None)

    # ---- consumer ----------------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        """-> ((audio [B,T,F,3] float32 on the device, audiolen numpy), (y [B,Ut] int32 on the device, tokenlen numpy))"""
        if self._prev is not None:
            s = self._prev
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self.consumed[s] = ev
            self.free[s].release()
            self._prev = None
        item = self.q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        s, B, T, Ut, audiolen, tokenlen = item
        torch.cuda.current_stream(self.dev).wait_event(self.ready[s])
        self._prev = s
        audio = self.d_feat[s][:B * T * self.F * 3].view(B, T, self.F, 3)
        y = self.d_tok[s][:B * Ut].view(B, Ut)
        return (audio, audiolen), (y, tokenlen)

    def close(self):
        self._stop = True
        for f in self.free:
            f.release()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self._th.join(timeout=5)
        if self.native and hasattr(self.source, "close"):
            self.source.close()


def feeder_for(source, device, feat_dim=13, is_training=True, depth=3, batch_scale=1):
    """DeviceFeeder sized for the reference's bucket table (tfrecord_data_loader.py:75-83); batch_scale: train.py --stack."""
    from tfrecord_data_loader import BUCKET_BATCH_LIMIT, EVAL_BOUNDARIES, TRAIN_BOUNDARIES
    bounds = TRAIN_BOUNDARIES if is_training else EVAL_BOUNDARIES
    k = max(int(batch_scale), 1)
    cap = max(k * BUCKET_BATCH_LIMIT[i] * (b - 1) * feat_dim * 3 for i, b in enumerate(bounds))
    return DeviceFeeder(source, device, feat_dim, cap, k * max(BUCKET_BATCH_LIMIT), 219 if is_training else 227, depth=depth)


class LaggedLog:
    """Per-step scalars (the loss) reach the host through asynchronous copies into pinned memory and are reported when their
    copy has completed -- one or two steps late -- so that logging never synchronises the device (the reference's
    `sess.run([loss, ...])` returns the loss of every step, train.py:114-125; here the same line is printed, lagged)."""

    def __init__(self, emit):
        self.emit = emit                     # emit(step_info, value)
        self.pending = []

    def push(self, info, scalar):
        pin = torch.empty(1, dtype=torch.float32).pin_memory()
        pin.copy_(scalar.detach().reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((info, pin, ev))
        self.drain(False)

    def drain(self, wait):
        while self.pending and (wait or self.pending[0][2].query()):
            info, pin, ev = self.pending.pop(0)
            if wait:
                ev.synchronize()
            self.emit(info, float(pin[0]))
