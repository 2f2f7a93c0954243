"""Batch sources for train.py / test.py.

`SyntheticBatches` reproduces the SHAPES of the reference's tf.data pipeline
(tfrecord_data_loader.py:75-94: bucket boundaries [639,1062,1275,1377,1449,1506,1563,1710] padded to
boundary-1, batch sizes [96,48,48,...], features [B,T,feat_dim,3] float32, tokens padded to 219) with the
synthetic content of SURVEY.md section 8(d).  The TFRecord reader itself is SURVEY 8(f) row F1 (next)."""
import numpy as np

BUCKET_BOUNDARIES = [639, 1062, 1275, 1377, 1449, 1506, 1563, 1710]      # tfrecord_data_loader.py:75
BUCKET_BATCH_SIZES = [96, 48, 48, 48, 48, 48, 48, 48]                    # tfrecord_data_loader.py:83
TOKEN_PAD = 219                                                           # tfrecord_data_loader.py:76


class SyntheticBatches:
    """Endless synthetic batches of the reference's bucket shapes.  The bucket of every step is drawn from a stream that depends
    on `seed` only, so all data-parallel ranks train on the SAME shape at every step (lock step); lengths, tokens and
    feature values are per rank.  The feature values are windows of one pre-drawn Gaussian pool (drawing 2.4 M fresh normals per
    batch would cost 40 ms of host time -- three train steps)."""

    def __init__(self, feat_dim=13, vocab_size=30, seed=0, rank=0, batch_scale=1.0, max_frames=None, buckets=None):
        self.feat_dim, self.vocab_size = feat_dim, vocab_size
        self.buckets = buckets                    # optional subset of bucket indices (bench.py: the bucket its headline is quoted on)
        self.shape_rng = np.random.RandomState(1234 + seed)
        self.rng = np.random.RandomState(1234 + seed + 7919 * rank)
        self.batch_scale = batch_scale
        self.max_frames = max_frames
        self._pool = None

    def __iter__(self):
        return self

    def _noise(self, n):
        if self._pool is None or self._pool.size < 2 * n:
            self._pool = self.rng.randn(2 * n + 4099).astype(np.float32)
        off = int(self.rng.randint(0, self._pool.size - n))
        return self._pool[off:off + n]

    def __next__(self):
        ks = list(self.buckets) if self.buckets else list(range(len(BUCKET_BOUNDARIES)))
        if self.max_frames:
            ks = [i for i in ks if BUCKET_BOUNDARIES[i] - 1 <= self.max_frames] or [0]
        k = ks[self.shape_rng.randint(0, len(ks))]
        T = BUCKET_BOUNDARIES[k] - 1
        lo = BUCKET_BOUNDARIES[k - 1] if k else 100
        B = max(1, int(BUCKET_BATCH_SIZES[k] * self.batch_scale))
        F = self.feat_dim
        audio = self._noise(B * T * F * 3).reshape(B, T, F, 3) * np.asarray([1.0, 0.5, 0.316], np.float32)
        audiolen = self.rng.randint(lo, T + 1, size=B).astype(np.int32)
        for b in range(B):
            audio[b, audiolen[b]:] = 0.0
        tokenlen = np.clip(np.round(0.12 * audiolen).astype(np.int32), 2, TOKEN_PAD)
        y = self.rng.randint(3, self.vocab_size, size=(B, TOKEN_PAD)).astype(np.int32)
        y[np.arange(TOKEN_PAD)[None, :] >= tokenlen[:, None]] = 0
        y[np.arange(B), tokenlen - 1] = 2
        return (audio, audiolen), (y, tokenlen)
