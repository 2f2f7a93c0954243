"""tfrecord_data_loader -- the reference's tf.data input pipeline without TensorFlow.

Counterpart of the reference's tfrecord_data_loader.py (`get_num_records` :17, `data_parser` :25,
`tfrecord_iterator` :54) and of the writer in create_tfrecord.py:44-95.  Pure host code (numpy): this is I/O on
either side of the hot path, not arithmetic.

Wire format (public TFRecord / tf.train.Example formats):
  record  = uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload)
  payload = Example{ features = 1: Features{ feature = 1: map<string, Feature> } }
  Feature = oneof { bytes_list = 1, float_list = 2 (packed floats), int64_list = 3 (packed varints) }
  keys    = 'feat' (flattened float32 [T, feat_dim, 3]), 'shape' (3 x int64), 'token' (int64 ids)
Pipeline semantics restated from tfrecord_data_loader.py:70-105 / tf.data.experimental.bucket_by_sequence_length:
  buckets  [0,639) [639,1062) ... [1563,1710) (eval: last boundary 3600), batch sizes 96,48,48,...;
  pad_to_bucket_boundary=True -> frames padded to boundary-1, tokens padded to 219 (train) / 227 (eval);
  a batch is emitted when its bucket fills, leftovers at end of data; training: shuffle(64) over batches, repeat().
"""
import glob
import struct

import numpy as np

# ------------------------------------------------------------------------------------------------
# crc32c (Castagnoli), masked as TFRecord does
# ------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab[i] = c
        _CRC_TABLE = tab
    return _CRC_TABLE


def crc32c(data):
    if len(data) > 4096:                              # payloads (150 KB of floats): the library's table-driven loop when it is there
        try:
            from las import _hip
            return int(_hip.lib().las_crc32c(bytes(data), len(data)))
        except Exception:
            pass
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# protobuf (just what Example needs)
# ------------------------------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    """Yield (field_number, wire_type, value) over a serialized message; length-delimited values are memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fn, wt, v


def _parse_feature(buf):
    for fn, wt, v in _fields(buf):
        if fn == 2:                                   # FloatList
            vals = []
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    vals.append(np.frombuffer(bytes(v2), dtype="<f4"))
                elif f2 == 1 and w2 == 5:
                    vals.append(np.frombuffer(bytes(v2), dtype="<f4"))
            return np.concatenate(vals) if vals else np.zeros(0, np.float32)
        if fn == 3:                                   # Int64List
            vals = []
            for f2, w2, v2 in _fields(v):
                if f2 == 1 and w2 == 2:
                    p = 0
                    while p < len(v2):
                        x, p = _varint(v2, p)
                        vals.append(x - (1 << 64) if x >= (1 << 63) else x)
                elif f2 == 1 and w2 == 0:
                    vals.append(v2 - (1 << 64) if v2 >= (1 << 63) else v2)
            return np.asarray(vals, np.int64)
        if fn == 1:                                   # BytesList
            return [bytes(v2) for f2, w2, v2 in _fields(v) if f2 == 1]
    return None


def parse_example(payload):
    """serialized tf.train.Example -> {key: ndarray}"""
    out = {}
    buf = memoryview(payload)
    for fn, wt, feats in _fields(buf):
        if fn != 1:
            continue
        for f2, w2, entry in _fields(feats):          # map<string, Feature> entries
            if f2 != 1:
                continue
            key, val = None, None
            for f3, w3, v3 in _fields(entry):
                if f3 == 1:
                    key = bytes(v3).decode()
                elif f3 == 2:
                    val = _parse_feature(v3)
            out[key] = val
    return out


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(feat, token):
    """feat float32 [T, feat_dim, 3], token int ids -> serialized Example (what create_tfrecord.py:83-87 builds)."""
    feat = np.asarray(feat, np.float32)
    entries = b""
    for key, feature in (
            ("feat", _ld(2, _ld(1, feat.reshape(-1).astype("<f4").tobytes()))),
            ("shape", _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in feat.shape)))),
            ("token", _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in token))))):
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feature))
    return _ld(1, entries)


# ------------------------------------------------------------------------------------------------
# TFRecord framing
# ------------------------------------------------------------------------------------------------
def tf_record_iterator(path, verify_payload_crc=False):
    """Yield the payload bytes of every record (tf.python_io.tf_record_iterator).  The length CRC is always
    checked; the payload CRC on request (pure-Python crc32c is slow on 200 KB feature blocks)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header in %s" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc32c(head[:8]) != lcrc:
                raise IOError("corrupted TFRecord length in %s" % path)
            payload = f.read(length)
            tail = f.read(4)
            if len(payload) < length or len(tail) < 4:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify_payload_crc and masked_crc32c(payload) != struct.unpack("<I", tail)[0]:
                raise IOError("corrupted TFRecord payload in %s" % path)
            yield payload


def write_tfrecord(path, feats, tokens):
    """TFRecordWriter over (feat, token) pairs, as create_tfrecord.py:44-95 does per file."""
    with open(path, "wb") as f:
        for feat, token in zip(feats, tokens):
            payload = serialize_example(feat, token)
            head = struct.pack("<Q", len(payload))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + payload + struct.pack("<I", masked_crc32c(payload)))


MAXLEN = 1710                      # create_tfrecord.py:28 -- training utterances at or above this are dropped


def create_tfrecords(X, y, filename, num_files=5, file_start_index=1):
    """create_tfrecord.py:44-95: split (X, y) into `num_files` shards `{filename}-{i}.tfrecord`, i counted from
    `file_start_index`; the remainder goes into the last shard.  Returns the record count."""
    assert len(X) == len(y)
    per = len(X) // num_files
    total = 0
    for i in range(num_files):
        st = i * per
        ed = (i + 1) * per if i != num_files - 1 else len(y)
        write_tfrecord("%s-%d.tfrecord" % (filename, i + file_start_index), X[st:ed], y[st:ed])
        total += ed - st
    return total


def get_num_records(files):
    """reference tfrecord_data_loader.py:17-22"""
    return sum(1 for fn in files for _ in tf_record_iterator(fn))


def data_parser(record):
    """reference tfrecord_data_loader.py:25-52 -> ((feat [T,feat_dim,3] f32, featlen), (token int32, tokenlen))"""
    ex = parse_example(record)
    shape = ex["shape"].astype(np.int32)
    feat = np.asarray(ex["feat"], np.float32).reshape(int(shape[0]), int(shape[1]), 3)
    token = ex["token"].astype(np.int32)
    return (feat, int(shape[0])), (token, int(token.shape[0]))


# ------------------------------------------------------------------------------------------------
# the dataset
# ------------------------------------------------------------------------------------------------
TRAIN_BOUNDARIES = [639, 1062, 1275, 1377, 1449, 1506, 1563, 1710]      # :75
EVAL_BOUNDARIES = [639, 1062, 1275, 1377, 1449, 1506, 1563, 3600]       # :80
BUCKET_BATCH_LIMIT = [96, 48, 48, 48, 48, 48, 48, 48, 48]               # :83


class _Rng:
    """The pipeline's shuffle stream: splitmix64, restated bit for bit in csrc/input.hip (`Rng`) so that the native reader
    and this iterator produce the same batch order from the same seed.  (The reference's order comes from TensorFlow's own
    generators and is not reproducible outside it.)"""
    M = (1 << 64) - 1

    def __init__(self, seed):
        self.s = int(seed) & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def randint(self, n):
        return self.next() % n

    def shuffle(self, v):
        for i in range(len(v), 1, -1):                                    # Fisher-Yates from the top
            j = self.randint(i)
            v[i - 1], v[j] = v[j], v[i - 1]


class _BucketedIterator:
    """list_files(shuffle) -> interleave(16) -> parse -> bucket_by_sequence_length(pad_to_bucket_boundary)
    [-> shuffle(64) -> repeat()].

    rank / world (data parallel, lock step): every rank walks the SAME record stream; a bucket emits when it holds
    world x batch-limit utterances and this rank keeps rows rank, rank + world, ... of that global batch (a leftover smaller
    than the world is dropped on every rank) -- all ranks train on the same bucket shape at every step."""

    def __init__(self, files, record_parser, feat_dim, is_training, seed=0, shuffle_buffer=64, cycle_length=16, rank=0, world=1, batch_scale=1):
        self.files = list(files)
        self.batch_scale = max(int(batch_scale), 1)                       # train.py --stack: k bucket batches per step and rank
        self.parser = record_parser
        self.feat_dim = feat_dim
        self.is_training = is_training
        self.bounds = TRAIN_BOUNDARIES if is_training else EVAL_BOUNDARIES
        self.max_tokenlen = 219 if is_training else 227                   # :76,:81
        self.rng = _Rng(seed)
        self.shuffle_buffer = shuffle_buffer if is_training else 0
        self.cycle_length = cycle_length
        self.rank, self.world = int(rank), max(int(world), 1)
        if not is_training and self.world > 1:
            # an end-of-data leftover smaller than the world would be skipped on every rank: WER / loss from an incomplete set
            raise ValueError("evaluation data is not sharded by the reader (is_training=False needs world=1): read every batch "
                             "and take this rank's share (las.parallel.shard; test.py / decode.py do)")
        self.initializer = None                                           # API parity with make_initializable_iterator
        self._gen = self._batches()

    def _records(self):
        files = list(self.files)
        self.rng.shuffle(files)
        active = []
        it = iter(files)
        while True:
            while len(active) < self.cycle_length:
                try:
                    active.append(tf_record_iterator(next(it), verify_payload_crc=True))     # as TFRecordDataset does (and csrc/input.hip)
                except StopIteration:
                    break
            if not active:
                return
            for r in list(active):                                        # block_length 1 round robin
                try:
                    yield next(r)
                except StopIteration:
                    active.remove(r)

    def _emit(self, k, items):
        T = self.bounds[k] - 1 if k < len(self.bounds) else max(x[0][1] for x in items)
        items = items[self.rank::self.world]
        B = len(items)
        feat = np.zeros((B, T, self.feat_dim, 3), np.float32)
        featlen = np.zeros(B, np.int32)
        tok = np.zeros((B, self.max_tokenlen), np.int32)
        toklen = np.zeros(B, np.int32)
        for i, ((f, fl), (t, tl)) in enumerate(items):
            feat[i, :fl] = f
            featlen[i] = fl
            tok[i, :tl] = t
            toklen[i] = tl
        return (feat, featlen), (tok, toklen)

    def _one_pass(self):
        buckets = [[] for _ in range(len(self.bounds) + 1)]
        for rec in self._records():
            xs, ys = self.parser(rec)
            n = xs[1]
            k = int(np.searchsorted(self.bounds, n, side="right"))
            if k >= len(self.bounds):
                # bucket_by_sequence_length(pad_to_bucket_boundary=True) rejects elements >= the last boundary
                raise ValueError("utterance of %d frames exceeds the last bucket boundary %d" % (n, self.bounds[-1]))
            if ys[1] > self.max_tokenlen:
                raise ValueError("token sequence of %d exceeds the padded length %d" % (ys[1], self.max_tokenlen))
            buckets[k].append((xs, ys))
            if len(buckets[k]) == BUCKET_BATCH_LIMIT[k] * self.batch_scale * self.world:
                yield self._emit(k, buckets[k])
                buckets[k] = []
        for k, items in enumerate(buckets):                               # leftovers at end of data
            if len(items) >= self.world:
                yield self._emit(k, items)

    def _batches(self):
        while True:
            buf = []
            for batch in self._one_pass():
                if self.shuffle_buffer:
                    buf.append(batch)
                    if len(buf) > self.shuffle_buffer:
                        yield buf.pop(self.rng.randint(len(buf)))
                else:
                    yield batch
            while buf:
                yield buf.pop(self.rng.randint(len(buf)))
            if not self.is_training:
                return                                                    # dataset.repeat(1)

    def get_next(self):
        """-> ((feat [B,T,feat_dim,3], featlen [B]), (token [B,219|227], tokenlen [B])); StopIteration at the end of an
        evaluation pass (tf.errors.OutOfRangeError in the reference)."""
        return next(self._gen)

    __next__ = get_next

    def __iter__(self):
        return self


class NativeReader:
    """The same pipeline through liblas_hip.so's reader (csrc/input.hip: one C++ producer thread, mmap'ed files, pinned batch
    ring): same constructor arguments, same batch order as `_BucketedIterator`.  `get_next()` returns numpy COPIES (API parity);
    the train loop uses `next_slot()` / `upload()` (las.input_pipeline.DeviceFeeder) and never copies on the host."""

    def __init__(self, files, feat_dim, is_training, seed=0, shuffle_buffer=64, cycle_length=16, rank=0, world=1, slots=4, batch_scale=1):
        import ctypes
        from las import _hip
        self._hip, self._ct = _hip, ctypes
        self.files = list(files)
        self.feat_dim, self.is_training = int(feat_dim), bool(is_training)
        bounds = TRAIN_BOUNDARIES if is_training else EVAL_BOUNDARIES
        self.max_tokenlen = 219 if is_training else 227
        cfg = _hip.InputConfig()
        cfg.feat_dim, cfg.is_training, cfg.n_bounds = self.feat_dim, int(self.is_training), len(bounds)
        for i, b in enumerate(bounds):
            cfg.bounds[i] = b
        for i, b in enumerate(BUCKET_BATCH_LIMIT):
            cfg.batch_limit[i] = b * max(int(batch_scale), 1)                 # (train.py --stack)
        cfg.max_tokenlen = self.max_tokenlen
        cfg.shuffle_buffer = shuffle_buffer if is_training else 0
        cfg.cycle_length, cfg.seed, cfg.rank, cfg.world, cfg.slots = cycle_length, int(seed) & ((1 << 64) - 1), int(rank), max(int(world), 1), slots
        arr = (ctypes.c_char_p * len(self.files))(*[f.encode() for f in self.files])
        self._h = _hip.lib().las_input_open(arr, len(self.files), ctypes.byref(cfg))
        if not self._h:
            raise IOError("las_input_open: %s" % _hip.lib().las_last_error().decode())
        self.initializer = None

    def next_slot(self):
        """-> las_input_batch (pointers into a pinned slot; release it with `release(slot)`), or None at the end of an evaluation pass"""
        b = self._hip.InputBatch()
        rc = self._hip.lib().las_input_next(self._h, self._ct.byref(b))
        if rc == 1:
            return None
        if rc != 0:
            raise IOError(self._hip.lib().las_last_error().decode())
        return b

    def arrays(self, b):
        """numpy views of a slot (valid until it is released)"""
        ct = self._ct
        F = self.feat_dim
        feat = np.ctypeslib.as_array(ct.cast(b.feat, ct.POINTER(ct.c_float)), shape=(b.B, b.T, F, 3))
        tok = np.ctypeslib.as_array(ct.cast(b.token, ct.POINTER(ct.c_int)), shape=(b.B, b.max_tokenlen))
        fl = np.ctypeslib.as_array(ct.cast(b.featlen, ct.POINTER(ct.c_int)), shape=(b.B,))
        tl = np.ctypeslib.as_array(ct.cast(b.tokenlen, ct.POINTER(ct.c_int)), shape=(b.B,))
        return (feat, fl), (tok, tl)

    def upload(self, b, d_feat, d_token, stream):
        self._hip.check(self._hip.lib().las_input_upload(self._h, b.slot, d_feat, d_token, stream), "las_input_upload")

    def release(self, b):
        self._hip.check(self._hip.lib().las_input_release(self._h, b.slot), "las_input_release")

    def get_next(self):
        b = self.next_slot()
        if b is None:
            raise StopIteration
        (feat, fl), (tok, tl) = self.arrays(b)
        out = (feat.copy(), fl.astype(np.int32)), (tok.astype(np.int32), tl.astype(np.int32))
        self.release(b)
        return out

    __next__ = get_next

    def __iter__(self):
        return self

    def records(self):
        return int(self._hip.lib().las_input_records(self._h))

    def close(self):
        if self._h:
            self._hip.lib().las_input_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tfrecord_iterator(filenames, record_parser, feat_dim=13, is_training=True, seed=0, rank=0, world=1, native=False, batch_scale=1):
    """reference tfrecord_data_loader.py:54-109.  `filenames`: a glob pattern or a list of paths.
    Returns (iterator, output_types, output_shapes).  native=True: the C++ reader of liblas_hip.so (same batches)."""
    files = sorted(glob.glob(filenames)) if isinstance(filenames, str) else list(filenames)
    if not files:
        raise IOError("no TFRecord files match %r" % (filenames,))
    if native:
        it = NativeReader(files, feat_dim, is_training, seed=seed, rank=rank, world=world, batch_scale=batch_scale)
    else:
        it = _BucketedIterator(files, record_parser, feat_dim, is_training, seed=seed, rank=rank, world=world, batch_scale=batch_scale)
    max_tok = it.max_tokenlen
    types = ((np.float32, np.int32), (np.int32, np.int32))
    shapes = (([None, None, feat_dim, 3], [None]), ([None, max_tok], [None]))
    return it, types, shapes
