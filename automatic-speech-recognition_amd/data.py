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
    def __init__(self, feat_dim=13, vocab_size=30, seed=0, rank=0, batch_scale=1.0, max_frames=None):
        self.feat_dim, self.vocab_size = feat_dim, vocab_size
        self.rng = np.random.RandomState(1234 + seed + 7919 * rank)
        self.batch_scale = batch_scale
        self.max_frames = max_frames

    def __iter__(self):
        return self

    def __next__(self):
        k = self.rng.randint(0, len(BUCKET_BOUNDARIES))
        if self.max_frames:
            ks = [i for i, b in enumerate(BUCKET_BOUNDARIES) if b - 1 <= self.max_frames] or [0]
            k = ks[self.rng.randint(0, len(ks))]
        T = BUCKET_BOUNDARIES[k] - 1
        lo = BUCKET_BOUNDARIES[k - 1] if k else 100
        B = max(1, int(BUCKET_BATCH_SIZES[k] * self.batch_scale))
        F = self.feat_dim
        audio = np.zeros((B, T, F, 3), np.float32)
        audio[..., 0] = self.rng.randn(B, T, F)
        audio[..., 1] = self.rng.randn(B, T, F) * 0.5
        audio[..., 2] = self.rng.randn(B, T, F) * 0.316
        audiolen = self.rng.randint(lo, T + 1, size=B).astype(np.int32)
        for b in range(B):
            audio[b, audiolen[b]:] = 0.0
        tokenlen = np.clip(np.round(0.12 * audiolen).astype(np.int32), 2, TOKEN_PAD)
        y = np.zeros((B, TOKEN_PAD), np.int32)
        for b in range(B):
            n = tokenlen[b]
            y[b, :n - 1] = self.rng.randint(3, self.vocab_size, size=n - 1)
            y[b, n - 1] = 2
        return (audio, audiolen), (y, tokenlen)
