#!/usr/bin/env python
"""bench.py -- utterances/sec of one full LAS train step (fwd + bwd + clip + Adam [+ all-reduce]).

Workload (BASELINE.json configs[1]): LibriSpeech-100 char LAS, 3 x pBLSTM-256 Listener (+ first BLSTM),
1 x LSTM-512 Speller, additive attention, bf16 contractions, on the reference's own bucket shape
B=48, T=1274 (tfrecord_data_loader.py:75,83), synthetic MFCC-39 cube and labels per SURVEY.md 8(d).

    python bench.py --gpus N --steps K --warmup W
With N > 1 and no WORLD_SIZE in the environment this process starts N fresh ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, one rank per GPU over RCCL), relays rank 0's
JSON line and exits with the launcher's code; under a launcher (the driver's form) it is one of the ranks.
Every rank trains on its own 48-utterance shard (weak scaling), one flat-bucket all-reduce per step.

Prints ONE JSON line (rank 0):
  roofline      dominant kernel family (the recurrent sweep), measured live with HIP events on the launch stream, plus the
                WHOLE-STEP algorithmic rates against the HBM and MFMA roofs (BASELINE.md section 3 per-utterance figures x utt/s)
  cpu_baseline  the oracle restatement of the reference graph AS WRITTEN (un-hoisted key projection, per-step cells) on
                the host cores: 1 warm-up + 3 timed steps on a bounded sample
  decode        (N = 1) device-resident beam search, beam 16 + 2 x 512 char RNNLM on synthetic T=1274 utterances
                (BASELINE configs[4]): utterances/s
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "automatic-speech-recognition_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak
# BASELINE.md section 3 (config 2, lstm cells, T=1274 -> T'=160, U=200): algorithmic work of one TRAINED utterance
STEP_BYTES_PER_UTT = 115e6
STEP_FLOPS_PER_UTT = 34.3e9


def bench_args(cell, config=1):
    """config 1 (default): BASELINE configs[1] -- char units, additive attention (the headline).
    config 3: BASELINE configs[3] -- subword vocabulary (V = 5000, train_subword.py) + location-aware attention at the reference
    defaults K = 201, C = 10 (las/arguments.py:130-137), same listener / speller sizes: the per-step row kernels with the conv1d over
    the previous alignment (the one-launch loop serves additive attention only)."""
    from helpers import make_args
    kw = dict(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128,
              attention_size=128, mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True,
              scheduled_sampling=True, vocab_size=30)
    if config == 3:
        kw.update(mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=5000, unit="subword")
    if config == "run_sh":
        # the reference's own recipe (run.sh:59-76 + the defaults it leaves alone, las/arguments.py:109-137): CNN listener with four
        # BLSTM-512, location-aware attention K = 201 / C = 10, two 1024-unit decoder cells, subword vocabulary
        kw.update(enc_type="cnn", enc_units=512, num_enc_layers=4, num_enc_channels=32, dec_units=1024, num_dec_layers=2,
                  embedding_size=256, attention_size=128, mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=5000,
                  unit="subword", lr=1e-4, scheduled_sampling=False)
    return make_args(**kw)


def sweep_bytes(B, T, H, G, bwd, f=2):
    """ALGORITHMIC HBM bytes of one recurrent sweep launch (both directions); f = bytes per stored activation (2: speed mode
    keeps the listener's activations in HBM as bf16 -- SURVEY 8(d)'s figure; 4: parity mode):
    fwd: read x-projection, write activated gates + cell state + h; W_hh fragments once.
    bwd: read activated gates, c_t, c_prev, dout; write d(pre-activation)."""
    gates = B * T * 2 * G * H * f
    h = B * T * 2 * H * f
    w = 2 * H * G * H * 2
    if not bwd:
        return gates * 2 + (h if G == 4 else 0) + h + w
    return gates * 2 + (2 * h if G == 4 else h) + h + w


def usable_cores():
    """Host cores this process may really use: scheduler affinity, cut by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_child(cell, budget_s=45.0):
    """CPU-only child process.  BASELINE.md section 2 protocol: one oracle train step = fwd + bwd + clip + Adam of the
    reference graph as written; 1 warm-up + 3 timed steps, median; batch as large as fits the time budget; all usable
    host cores (count stated).  TensorFlow 1.13 itself cannot be run (not installable offline)."""
    from helpers import synthetic_batch
    from oracle import las_oracle as O
    args = bench_args(cell)
    T = 1274

    def one(Bs):
        xs, ys = synthetic_batch(Bs, T, 256, args.vocab_size, seed=0, min_frac=0.834)
        po = O.to_torch(O.init_params(args, seed=0, cell=cell), requires_grad=True)
        z1 = {k: torch.zeros_like(v) for k, v in po.items()}
        z2 = {k: torch.zeros_like(v) for k, v in po.items()}
        t0 = time.time()
        O.train_step(po, z1, z2, 0, (torch.tensor(xs[0]), xs[1]), (torch.tensor(ys[0]), ys[1]), args, cell, hoist=False)
        return time.time() - t0

    n_all = usable_cores()
    # torch's intra-op pool does not always scale to every core on these small per-step products: probe two pool sizes
    # on a small batch and keep the faster one (both stated)
    probes = {}
    for n in sorted({n_all, min(n_all, 16)}, reverse=True):
        torch.set_num_threads(n)
        one(2)
        probes[n] = one(2)
    ncores = min(probes, key=probes.get)
    torch.set_num_threads(ncores)
    # batch as large as fits the budget: grow from B=4 while a step scales about linearly (a step at the next size that
    # takes more than 1.5x the linear prediction -- allocator / cache pathologies seen at B=8 on some hosts -- ends the search)
    Bs, t_prev = 4, one(4)
    for cand in (8, 16, 32, 48):
        if 4.5 * t_prev * cand / Bs > budget_s:
            break
        t_c = one(cand)
        if t_c > 1.5 * t_prev * cand / Bs:
            break
        Bs, t_prev = cand, t_c
    one(Bs)                                                 # warm-up at the chosen size
    ts = sorted(one(Bs) for _ in range(3))
    med = ts[1]
    # decode leg (BASELINE configs[4]): the oracle's beam search, beam 16 + 2x512 char RNNLM, ONE synthetic T=1274 utterance
    dec = None
    try:
        from helpers import lm_params, oracle_lm, oracle_decode
        dargs = bench_args(cell)
        dargs.beam_size, dargs.apply_lm, dargs.lm_weight, dargs.convert_rate = 16, True, 0.5, 0.166
        p0 = O.init_params(dargs, seed=0, cell=cell)
        p0["Speller/decode/dense/bias"][2] = -1e4           # no hypothesis ends early: all int(T * convert_rate) = 211 steps, like the GPU leg
        olm = (oracle_lm(lm_params(np.random.RandomState(8), 28, 0, 512, 2), 0, 2), 512, 2)
        xs1, _ = synthetic_batch(1, T, 8, 30, seed=100)
        t0 = time.time()
        res = oracle_decode(xs1, p0, dargs, cell, 16, lm=olm, lm_weight=0.5, hoist=False)       # as written: keys re-projected per step
        dt = time.time() - t0
        dec = {"value": round(1.0 / dt, 4), "unit": "utterances/s", "cores": ncores,
               "sample": "oracle beam search (las/beam_search.py:61-158 restated as written: key projection recomputed at every step for "
                         "every hypothesis row, torch-CPU fp32), 1 utterance, %d steps, %.1f s"
                         % (max(len(r.token_ids) - 1 for r in res), dt)}
    except Exception as e:                                  # the train baseline must not depend on this leg
        dec = {"value": None, "sample": "oracle decode failed: %s" % e}
    print(json.dumps({"value": round(Bs / med, 4), "unit": "utterances/s", "cores": ncores, "kind": "port", "decode": dec,
                      "sample": "reference-equivalent CPU path (restated; TF unavailable offline): oracle train step of the "
                                "reference graph as written (un-hoisted key projection, torch-CPU fp32, %s cells), B=%d of the "
                                "same T=1274 workload, 1 warm-up + 3 timed steps, median %.2f s/step; %d of %d usable cores "
                                "(pool probe: %s)" % (cell, Bs, med, ncores, n_all,
                                                      ", ".join("%d thr %.2f s" % (k, v) for k, v in sorted(probes.items())))}))


def cpu_baseline(cell, timeout=300.0):
    """Timed in a child process that never touches the GPU, with a hard time limit so the bench line always appears."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cell", cell],
                           capture_output=True, text=True, timeout=timeout, env=env)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                          # timeout / parse failure: say so instead of hanging the bench
        return {"value": None, "unit": "utterances/s", "cores": usable_cores(), "kind": "port",
                "sample": "oracle train step did not finish within %.0f s (%s)" % (timeout, type(e).__name__)}


def recorded_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes recorded under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_summary.py).
    Only a recording made from THIS kernel source counts: the file stores the sha of csrc/rnn_seq.hip; else null."""
    import glob
    try:
        src = os.path.join(ROOT, "automatic-speech-recognition_amd", "csrc", "rnn_seq.hip")
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("rnn_seq_sha16") != sha:
                continue
            for k, v in rec["kernels"].items():
                if k.startswith(kernel_prefix):
                    return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def recorded_decode_traffic():
    """{kernel: memory-side bytes per launch} of the decode step's Speller launches from the newest profiles/*_decode_pmc.json that was
    recorded from THIS csrc/speller.hip (FETCH_SIZE x2 + WRITE_SIZE, separate passes; tools/pmc_summary.py), else None."""
    import glob
    try:
        src = os.path.join(ROOT, "automatic-speech-recognition_amd", "csrc", "speller.hip")
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_decode_pmc.json")), reverse=True):
            rec = json.load(open(path))
            if rec.get("speller_sha16") != sha:
                continue
            return {k: v["hbm_bytes_per_launch"] for k, v in rec["kernels"].items()
                    if k.startswith(("dec_step_fwd_pf", "dec_beam_rows4", "lstm_cell_rows", "beam_")) and "hbm_bytes_per_launch" in v}
    except Exception:
        pass
    return None


def decode_bench(dev, cell, dtype, nutt=16, beam=16, T=1274):
    """BASELINE configs[4]: beam search (beam 16) + char RNNLM shallow fusion on synthetic T=1274 utterances, bench
    architecture, device-resident batched loop (BeamSearch.decode_batch).  Random-init weights: hypotheses do not end
    early, every utterance runs its full int(T * convert_rate) = 211 steps."""
    from helpers import synthetic_batch
    from las import layers as L, variables as V
    from las.beam_search import BeamSearch
    from las.las import LAS, Listener, Speller
    from lang.char_rnn_model import CharRNN
    from utils.tokenizer import CharEncoder
    L.set_cell(cell)
    L.set_precision(dtype)
    st = V.reset_default_store(device=dev, seed=0)
    args = bench_args(cell)
    args.beam_size, args.apply_lm, args.lm_weight, args.convert_rate = beam, True, 0.5, 0.166
    tok = CharEncoder()
    las = LAS(args, Listener, Speller, tok.token_to_id)
    lm = CharRNN(False, 1, 1, 28, 512, embedding_size=0, num_layers=2, store=st)
    lm.params()
    las.build_variables()
    bs = BeamSearch(args, las, tok.token_to_id, lm)
    utts = []
    for k in range(nutt):
        xs, _ = synthetic_batch(1, T, 8, 30, seed=100 + k)
        utts.append(xs)
    bs.decode_batch(None, utts[:2])                      # warm-up (library init, allocator)
    bs.decode_batch(None, utts)                          # ... and once at the timed geometry (graph capture, workspaces)
    torch.cuda.synchronize()
    # One timing = `groups` consecutive decode_batch calls of `nutt` utterances (groups * nutt >= 64 utterances, ~0.1 s); `reps` timings,
    # the MEDIAN is `value` and the spread is stated (a single 25 ms call -- rounds 1-3 -- moved by 20 % between boxes / runs).
    groups, reps = max(1, -(-64 // nutt)), 7
    rates = []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        for _g in range(groups):
            res = bs.decode_batch(None, utts)
        torch.cuda.synchronize()
        if rep:                                          # (the first repetition is cold -- r4's driver run: 433 against 745 utt/s -- and not part of the spread)
            rates.append(groups * nutt / (time.perf_counter() - t0))
    rates.sort()
    dt = nutt / rates[len(rates) // 2]                   # seconds per `nutt` utterances at the median rate
    steps = max(len(r[-1].token_ids) - 1 for r in res)
    # a second, instrumented pass (device syncs between the phases, then the three parts of a step timed alone with HIP events):
    # where the time of a decode step goes, and the dominant part against the HBM roof
    bs.measure = True
    bs.decode_batch(None, utts)
    tm = bs.last_timing or {}
    bs.measure = False
    # what a real test set looks like: every utterance its own length (the reference encoder has no length mask, so nothing is padded):
    # ONE encoder pass over rows of different lengths (las_rnn_seq_fwd_rows; BeamSearch.ragged_encoder), and for comparison 16 encoders
    # side by side on several streams (parallel_encoders) and one after the other
    ragged = []
    for k in range(nutt):
        Tk = T - 18 * k                                  # 1274 ... 1004 frames
        xs, _ = synthetic_batch(1, Tk, 8, 30, seed=300 + k)
        ragged.append(xs)
    rag = {}
    for name, rg, par in (("one_ragged_encoder_pass", True, True), ("parallel_encoders", False, True), ("one_encoder_at_a_time", False, False)):
        bs.ragged_encoder, bs.parallel_encoders = rg, par
        bs.decode_batch(None, ragged[:3])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        bs.decode_batch(None, ragged)
        torch.cuda.synchronize()
        rag[name] = round(nutt / (time.perf_counter() - t1), 1)
    bs.ragged_encoder = bs.parallel_encoders = True
    # more utterances per device-resident batch (decode.py --decode_batch, default 64): the step's kernels are latency-bound at 256 rows,
    # so 512 / 1024 rows per step cost 1.4x / 2.2x the step time for 2x / 4x the utterances
    larger, big, stream = {}, None, None
    for nb, nrep in ((32, 3), (64, 7)):
        try:
            more = []
            for k in range(nb):
                xs, _ = synthetic_batch(1, T, 8, 30, seed=100 + k)
                more.append(xs)
            bs.decode_batch(None, more)
            torch.cuda.synchronize()
            rs = []
            for rep in range(nrep + 1):
                t1 = time.perf_counter()
                bs.decode_batch(None, more)
                torch.cuda.synchronize()
                if rep:
                    rs.append(nb / (time.perf_counter() - t1))
            rs.sort()
            larger[str(nb)] = round(rs[len(rs) // 2], 1)
            if nb == 64:
                big = {"value": round(rs[len(rs) // 2], 2), "min": round(rs[0], 1), "max": round(rs[-1], 1), "repetitions": nrep,
                       "spread": round((rs[-1] - rs[0]) / rs[len(rs) // 2], 4)}
                # decode.py's loop over a test set (BeamSearch.decode_batches): the encoders of batch k+1 on a second stream under the
                # search of batch k -- 6 batches per timing, the median of 5
                list(bs.decode_batches(None, [more] * 2))
                torch.cuda.synchronize()
                ss = []
                for rep in range(5):
                    t1 = time.perf_counter()
                    for _r in bs.decode_batches(None, [more] * 6):
                        pass
                    torch.cuda.synchronize()
                    ss.append(6 * nb / (time.perf_counter() - t1))
                ss.sort()
                stream = {"value": round(ss[2], 2), "min": round(ss[0], 1), "max": round(ss[-1], 1), "batches_per_timing": 6, "repetitions": 5,
                          "note": "BeamSearch.decode_batches (what decode.py runs): a stream of 64-utterance batches, the encoders of the next "
                                  "batch overlapped with the search of the current one; `value` stays one decode_batch call at a time"}
        except Exception as e:
            larger[str(nb)] = "%s: %s" % (type(e).__name__, str(e)[:120])
    parts = tm.get("parts_us", {})
    N, Tp = nutt * beam, tm.get("frames", 160)
    D, A, Hd, E = args.dec_units, args.attention_size, 2 * args.enc_units, args.embedding_size
    f = 2 if dtype == "bf16" else 4
    # algorithmic HBM bytes of one decode step = every DISTINCT operand once (all rows of an utterance share its keys / encoder
    # rows, all rows share the weights; re-reads are L2 hits by construction): Speller = Ws + per-utterance keys and encoder rows
    # + the cell weights + the rows' state in and out; LM = its two cell kernels (fp32 masters, converted in the loader) + state
    sp_bytes = D * A * f + nutt * Tp * (A + Hd) * f + (E + Hd + D) * 4 * D * f + N * (2 * 2 * D * 4 + Tp * 4 * 2)
    lm_bytes = ((28 + 512) * 2048 + (512 + 512) * 2048) * 4 + N * 2 * 2 * 2 * 512 * 4
    roof = None
    if parts:
        dom = max(parts, key=parts.get)
        byts = {"speller": sp_bytes, "lm": lm_bytes}.get(dom, 0)
        roof = {"bound": "hbm", "part": dom, "kernels": {"speller": "dec_step_fwd_pf_kernel<1,10> (attention rows) + lstm_cell_rows_big_kernel<fast, bf16 rows, 32> (cell product + gate math; in the search itself one launch with the LM's first layer: lstm_cell_rows_big_pair_kernel); 256 rows",
                                                          "lm": "lstm_cell_rows_big_kernel<exact, fp32 rows, 32> (layer 1) + lstm_cell_rows_kernel<exact, fp32> (layer 2): las_lstm_cell_rows, one launch per LM layer",
                                                          "beam": "beam_loop_kernel (both vocabulary projections + ranking + bookkeeping + alignment filing) + beam_gather_kernel"}[dom],
                "us_per_decode_step": parts[dom], "algorithmic_bytes_per_step": int(byts),
                "achieved": round(byts / (parts[dom] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(byts / (parts[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                # counter bytes (FETCH_SIZE x2 + WRITE_SIZE per launch) of the step's Speller / LM / pruning kernels, beside the algorithmic
                # figure above: what the launches really pull through the fabric (the attention rows moved 12-16x their
                # distinct operands; round 6 keeps an utterance's hypothesis rows on one XCD)
                "traffic": recorded_decode_traffic(),
                "note": "latency-bound: one decode step is a chain of 5 dependent launches over %d rows (8-13 us each in the replayed graph); "
                        "traffic = memory-side bytes per launch from the recorded rocprofv3 --pmc passes when they were made from this "
                        "csrc/speller.hip, else null" % N}
    # `value`: the rate of what decode.py does by default since round 4 -- 64 utterances per device-resident batch (1024 hypothesis rows per
    # step); `at_16_utterances`: rounds 1-3's geometry (256 rows), with the per-step parts and the roofline that were measured there
    at16 = round(nutt / dt, 2)
    return {"value": big["value"] if big else at16, "unit": "utterances/s", "beam": beam, "lm": "2x512 char RNNLM, lm_weight 0.5",
            # `value` changed geometry in round 4 (16 -> 64 utterances per device-resident batch = decode.py's default); both
            # geometries under explicit keys so that rounds compare like with like: value_b16 is rounds 1-3's `value`
            "value_b16": at16, "value_b64": big["value"] if big else None, "value_b64_stream": stream,
            "utterances_per_batch_of_value": 64 if big else nutt, "value_timing": big, "at_16_utterances": at16,
            "utterances": nutt, "frames": T, "decode_steps": steps, "dtype": dtype, "seconds": round(dt, 4),
            "timing": {"utterances_per_timing": groups * nutt, "repetitions": reps, "value_is": "median",
                       "min": round(rates[0], 1), "max": round(rates[-1], 1), "spread": round((rates[-1] - rates[0]) / rates[len(rates) // 2], 4)},
            "us_per_decode_step": round((tm.get("searched", 0.0)) / max(tm.get("steps", 1), 1) * 1e6, 1) if tm else None,
            "phases_s": {k: tm[k] for k in ("encoded", "searched", "done") if k in tm}, "step_parts_us": parts, "roofline": roof,
            "ragged": dict(rag, unit="utterances/s", frames="%d utterances of %d ... %d frames, all different" % (nutt, T - 18 * (nutt - 1), T)),
            "utterances_per_batch": dict(larger, unit="utterances/s", note="the same search with 32 / 64 utterances (512 / 1024 hypothesis rows) per "
                                         "device-resident batch; `timing`, `us_per_decode_step`, `step_parts_us`, `roofline`, `ragged` are at %d" % nutt),
            "note": "utterances of equal length share one encoder launch (rows are independent; the reference encoder has no length "
                    "mask, so utterances are never padded to a common length); the search runs all utterances x beam rows per "
                    "step on the device, one captured step replayed as a HIP graph"}


def train_loop_bench(las, dev, a, value, steps=20, warmup=4):
    """utterances/s of the LOOP train.py runs (VERDICT r2 Missing #3): batches produced by a background reader, staged in pinned
    memory, copied host->device on a copy stream while the previous step computes (las.input_pipeline.DeviceFeeder), loss logged
    from asynchronous copies -- on the SAME bucket the headline is quoted on, so the two are comparable:
      synthetic  data.SyntheticBatches (what `train.py --synthetic True` reads)
      tfrecord   TFRecord files of that bucket's shapes written to a temporary directory, read back by the C++ reader of
                 liblas_hip.so (csrc/input.hip: mmap, parse, bucket, pinned slots) -- the path `train.py` takes on real data."""
    import shutil
    import tempfile
    from data import BUCKET_BOUNDARIES, SyntheticBatches
    from las.input_pipeline import LaggedLog, feeder_for
    import tfrecord_data_loader as tdl
    T, B = a.frames, a.batch
    k = BUCKET_BOUNDARIES.index(T + 1) if T + 1 in BUCKET_BOUNDARIES else None
    if k is None or B != tdl.BUCKET_BATCH_LIMIT[k]:
        return {"skipped": "not one of the reference's bucket shapes"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="las_tfr_")
    try:
        rng = np.random.RandomState(3)
        lo = BUCKET_BOUNDARIES[k - 1] if k else 100
        files = []
        for i in range(4):                                   # 4 files x 96 utterances of this bucket's length range (~55 MB)
            lens = rng.randint(max(lo, int(0.834 * T)), T + 1, size=96)
            feats = [rng.randn(n, 13, 3).astype(np.float32) for n in lens]
            toks = [np.r_[rng.randint(3, 30, size=int(0.15 * n) - 1), 2] for n in lens]
            fn = os.path.join(tmp, "train-%d.tfrecord" % i)
            tdl.write_tfrecord(fn, feats, toks)
            files.append(fn)
        for name in ("synthetic", "tfrecord"):
            if name == "synthetic":
                src = SyntheticBatches(13, 30, seed=0, buckets=[k])
            else:
                src = tdl.tfrecord_iterator(files, tdl.data_parser, 13, seed=0, native=True)[0]
            feed = feeder_for(src, dev, 13)
            seen = []
            log = LaggedLog(lambda info, v: seen.append(v))
            n_utt = 0
            for i in range(warmup + steps):
                if i == warmup:
                    torch.cuda.synchronize()
                    t0, n_utt = time.perf_counter(), 0
                xs, ys = next(feed)
                loss = las.train(xs, ys)[0]
                log.push(i, loss)
                n_utt += int(xs[0].shape[0])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            log.drain(True)
            las.check_status()
            feed.close()
            out[name] = {"value": round(n_utt / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
                         "vs_resident_batch": round(n_utt / dt / value, 4), "losses_logged": len(seen)}
        rd = tdl.tfrecord_iterator(files, tdl.data_parser, 13, seed=0, native=True)[0]
        t0, n_rd = time.perf_counter(), 0
        for _ in range(60):                                     # the reader alone: parse + bucket + assemble into pinned slots
            bslot = rd.next_slot()
            n_rd += bslot.B
            rd.release(bslot)
        out["reader_alone"] = {"value": round(n_rd / (time.perf_counter() - t0), 0), "note": "csrc/input.hip producer thread, no device copy"}
        rd.close()
        out["unit"] = "utterances/s"
        out["note"] = ("the loop of train.py on the headline's bucket: reader thread + pinned staging + host->device copy on a copy "
                       "stream included, loss logged without synchronising; tfrecord = the C++ reader of liblas_hip.so on files "
                       "of this bucket's shapes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def side_step_bench(dev, cell, dtype, config, B, T, steps=5, warmup=2, stack=1, seed=0, global_step=None):
    """ms per train step of ANOTHER configuration than the headline's, reported beside it (same protocol: synthetic batch resident in
    HBM, warm-up, timed steps between synchronisations).  stack = k: k bucket batches of B rows as one step (LAS.train_stacked)."""
    from helpers import synthetic_batch
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    L.set_cell(cell)
    L.set_precision(dtype)
    V.reset_default_store(device=dev, seed=0)
    args = bench_args(cell, config)
    las = LAS(args, Listener, Speller, {})
    las.build_variables()
    batches = []
    for k in range(stack):
        xs, ys = synthetic_batch(B, T, 256, args.vocab_size, seed=seed + k, min_frac=0.834)
        batches.append(((torch.tensor(xs[0], device=dev), xs[1]), (torch.tensor(ys[0], device=dev), ys[1])))
    if stack > 1:                                        # (stacked on the device once: the timed region holds the step, not the concatenation)
        W = max(int(b[1][0].shape[1]) for b in batches)
        audio = torch.cat([b[0][0] for b in batches], 0)
        y = torch.cat([torch.nn.functional.pad(b[1][0], (0, W - b[1][0].shape[1])) for b in batches], 0)
        batches = [((audio, np.concatenate([b[0][1] for b in batches])), (y, np.concatenate([b[1][1] for b in batches])))]
    xs, ys = batches[0]
    st = V.default_store()
    if global_step is not None:                          # scheduled sampling ACTIVE: the schedule reads the global step (las/las.py:177-183)
        st.global_step = global_step
    for _ in range(warmup):
        las.train(xs, ys)
    torch.cuda.synchronize()
    if global_step is not None:
        st.global_step = global_step
    t0 = time.perf_counter()
    for _ in range(steps):
        las.train(xs, ys)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    las.check_status()
    n = int(xs[0].shape[0])
    out = {"cell": cell, "dtype": dtype, "rows": n, "frames": T, "dec_steps": int(ys[1].max()), "ms_per_step": round(dt / steps * 1e3, 3),
           "value": round(n * steps / dt, 1), "unit": "utterances/s", "steps": steps, "schedule": dict(las.last_variants)}
    if global_step is not None:
        tok = las.speller.last_tokens_in
        out["teacher_forcing_rate"] = round(float(las.speller._scheduled_sampling(global_step)), 4)
        out["global_step"] = int(global_step)
        out["sampled_steps_last"] = int((tok[1:, 0] != ys[0][0, :tok.shape[0] - 1].to(tok.dtype)).sum()) if tok is not None else None
    sv = _hip_speller_variant()
    if sv is not None:
        out["speller_kernels"] = sv
    return out


def _hip_speller_variant():
    """Which kernel family served the Speller of the last step (las_speller_last_variant: a bit mask the library keeps per call)."""
    try:
        from las import _hip
        return _hip.speller_last_variant()
    except Exception:
        return None


def side_legs(dev, a, value):
    """The other configurations BASELINE.json names / VERDICT r3 asked a number for, each as one object inside the headline line:
      config3   BASELINE configs[3] on one rank (V = 5000 subword + location-aware attention K = 201, C = 10), same bucket
      cell_rnn  the cell the reference really builds (BasicRNNCell, las/layers.py:31, las/las.py:194), speed and parity mode
      stacked   k = 2, 4 bucket batches per step on ONE GPU (LAS.train_stacked = the update of k data-parallel ranks): what the
                latency-bound sweeps leave of the machine, used; also the single-GPU stand-in for the unmeasured scaling curve"""
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:
            out[name] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    B, T = a.batch, a.frames
    leg("config3", lambda: dict(side_step_bench(dev, a.cell, a.dtype, 3, B, T), workload="BASELINE configs[3], one rank: V=5000, location-aware K=201 C=10"))
    leg("cell_rnn", lambda: {"bf16": side_step_bench(dev, "rnn", "bf16", 1, B, T), "f32": side_step_bench(dev, "rnn", "f32", 1, B, T, steps=3, warmup=1),
                             "note": "BasicRNNCell recurrences (the reference as written); the headline is BasicLSTMCell, as BASELINE.json's north star names"})

    def run_sh():
        r = {"workload": "the reference's recipe run.sh:59-76: --enc_type cnn (default) --enc_units 512 --num_enc_layers 4 --dec_units 1024 "
                         "--num_dec_layers 2 --embedding_size 256 --attention_size 128 --mode loc (K=201, C=10), V=5000; bucket B=%d T=%d (T'=319)" % (B, T)}
        r["rnn"] = side_step_bench(dev, "rnn", a.dtype, "run_sh", B, T)      # BasicRNNCell: what las/las.py:191-199 builds
        r["lstm"] = side_step_bench(dev, "lstm", a.dtype, "run_sh", B, T)
        r["ms_per_step"] = r["rnn"]["ms_per_step"]
        return r
    leg("run_sh", run_sh)

    def sampling():
        args0 = bench_args(a.cell)
        mid = (args0.warmup_step + args0.max_step) // 2
        r = dict(side_step_bench(dev, a.cell, a.dtype, 1, B, T, steps=10, warmup=3, global_step=mid),
                 workload="BASELINE configs[2] on one rank with scheduled sampling ACTIVE: the global step set into the middle of the linear decay "
                          "(las/las.py:177-183), so that a share of the decode steps draw their input token on the device (in-loop logits + Gumbel arg-max)")
        r["vs_teacher_forcing"] = round(r["ms_per_step"] / (1e3 * B / value), 3)
        return r
    leg("config2_sampling", sampling)

    def stacked():
        r = {}
        for k in (2, 4):
            r["k%d" % k] = dict(side_step_bench(dev, a.cell, a.dtype, 1, B, T, stack=k), vs_one_batch=None)
            r["k%d" % k]["vs_one_batch"] = round(r["k%d" % k]["value"] / value, 3)
        r["note"] = ("k batches of the bucket stacked along the batch axis in ONE step: loss normalised by the token count of all k*B rows = the "
                     "update k data-parallel ranks compute (tests/test_gpu_dp.py); NOT the headline (that is the reference's per-GPU batch)")
        return r
    leg("stacked", stacked)
    return out


def parity_mode_bench(dev, a, xs, ys, steps=5, warmup=2):
    """The same step in the PARITY mode (--dtype f32: fp32 storage and arithmetic everywhere, the mode the f32 rows of the parity
    table are measured in): reported beside the speed-mode headline so that the price of exactness is on record."""
    from las import layers as L, variables as V
    from las.las import LAS, Listener, Speller
    L.set_precision("f32")
    try:
        V.reset_default_store(device=dev, seed=0)
        las = LAS(bench_args(a.cell, a.config), Listener, Speller, {})
        las.build_variables()
        for _ in range(warmup):
            las.train(xs, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            las.train(xs, ys)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        las.check_status()
        return {"dtype": "f32", "ms_per_step": round(dt / steps * 1e3, 3), "value": round(xs[0].shape[0] * steps / dt, 1),
                "unit": "utterances/s", "steps": steps}
    finally:
        L.set_precision(a.dtype)


def self_launch(n, argv):
    """Run `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child (one rank per GPU over RCCL).
    Called before anything initialises the GPU in this process (torch.cuda.device_count() does not)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py: --gpus %d requested but only %d device(s) are visible\n" % (n, have))
        return 2
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif r.returncode == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no result line\n")
        return 3
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cell", default="lstm", choices=["lstm", "rnn"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--frames", type=int, default=1274)
    ap.add_argument("--config", type=int, default=1, choices=[1, 3], help="BASELINE.json configs[] index (1 = headline, 3 = subword + location-aware)")
    ap.add_argument("--only-leg", default=None, choices=["run_sh", "run_sh_rnn", "run_sh_lstm", "config2_sampling"],
                    help="run ONE side leg alone and print its JSON object (for rocprofv3 kernel traces of that leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-train-loop", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the config3 / cell_rnn / stacked objects")
    ap.add_argument("--decode-only", action="store_true", help="only the decode leg (BASELINE configs[4]); prints its JSON object")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_only:
        cpu_baseline_child(a.cell)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this parent never touches the GPU; it starts N fresh ranks
        # under torch.distributed.run, relays rank 0's JSON line and exits with the launcher's code
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (a.gpus, world, a.gpus))
    if world > 1:
        # N ranks share the host: a rank's few CPU-side torch ops must not each fork a pool as wide as the machine (the launch thread of
        # every rank competes with them; cf. tests/conftest.py)
        torch.set_num_threads(max(1, min(8, usable_cores() // world)))
    # the driver stack may print to fd 1 while the device is initialised (libdrm's "amdgpu.ids" notice): keep stdout
    # for the ONE JSON line by pointing fd 1 at stderr until the result is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    from helpers import synthetic_batch
    from las import _hip
    from las import layers as L
    from las import variables as V
    from las.las import LAS, Listener, Speller
    from las.parallel import DataParallel

    dp = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LAS_DIST_BACKEND", "nccl")           # "nccl" IS RCCL on ROCm; gloo only for 1-GPU dry runs
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dp = DataParallel()

    if a.only_leg:
        if a.only_leg.startswith("run_sh"):
            cells = ("rnn", "lstm") if a.only_leg == "run_sh" else (a.only_leg.split("_")[-1],)
            out = {c: side_step_bench(dev, c, a.dtype, "run_sh", a.batch, a.frames, steps=a.steps, warmup=a.warmup) for c in cells}
        else:
            args0 = bench_args(a.cell)
            out = side_step_bench(dev, a.cell, a.dtype, 1, a.batch, a.frames, steps=a.steps, warmup=a.warmup,
                                  global_step=(args0.warmup_step + args0.max_step) // 2)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({a.only_leg: out}), flush=True)
        return
    if a.decode_only:
        out = decode_bench(dev, a.cell, a.dtype)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"decode": out}), flush=True)
        return
    L.set_cell(a.cell)
    L.set_precision(a.dtype)
    torch.manual_seed(1000003 + rank)
    V.reset_default_store(device=dev, seed=0)
    args = bench_args(a.cell, a.config)
    las = LAS(args, Listener, Speller, {})
    las.dp = dp
    las.build_variables()
    st = V.default_store()
    if dp is not None:
        dp.broadcast_(st.flat)

    B, T = a.batch, a.frames
    xs, ys = synthetic_batch(B, T, 256, args.vocab_size, seed=rank, min_frac=0.834)   # lengths within the bucket
    xs = (torch.tensor(xs[0], device=dev), xs[1])
    ys = (torch.tensor(ys[0], device=dev), ys[1])
    U = int(ys[1].max())

    def sync():
        torch.cuda.synchronize()
        if dp is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        las.train(xs, ys)
    sync()
    _hip.prof_begin()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = las.train(xs, ys)[0]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    las.check_status()                              # a sweep exchange timeout would invalidate the measurement
    if dp is not None:
        dist.barrier()
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
    prof = _hip.prof_end()
    loss = float(loss)

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = world * B * a.steps / elapsed
        G = 4 if a.cell == "lstm" else 1
        H = args.enc_units
        per = {k: (sum(v) / len(v), len(v)) for k, v in prof.items()}
        if os.environ.get("LAS_PHASES"):            # spans of the step's phases and sweeps on the launch stream (no profiler needed)
            for k in sorted(per):
                print("  %-34s %8.3f ms  x%d" % (k, per[k][0], per[k][1] // a.steps), file=sys.stderr)
        # dominant kernel = the recurrent-sweep kernel family (fwd or bwd) with the largest total time in the timed
        # region.  achieved = algorithmic bytes of ALL its launches / their total duration (HIP events on the launch
        # stream), avg_launch_ms = mean over the same launches -> directly comparable with rocprofv3's per-kernel average.
        fam = {}
        for k, (avg, n) in per.items():
            if not k.startswith("rnn_seq"):
                continue
            name = k.split("[")[0]
            Tl = int(k.split("T=")[1].split(",")[0])
            f = fam.setdefault(name, {"ms": 0.0, "n": 0, "bytes": 0, "steps": 0})
            f["ms"] += avg * n
            f["n"] += n
            f["bytes"] += sweep_bytes(B, Tl, H, G, name.endswith("bwd"), 2 if a.dtype == "bf16" else 4) * n
            f["steps"] += Tl * n
        roof = None
        if fam:
            dom = max(fam, key=lambda k: fam[k]["ms"])
            f = fam[dom]
            bwd = dom.endswith("bwd")
            ach = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            if a.dtype == "bf16":
                # template arguments <CELL, UT, P, RB[, CH]>: LSTM = 1, 4 unit tiles per wave, clusters of P = 4 CUs, 8-row batch tiles
                kname = "rnn_seq_bwd_ks_kernel<1,4,4,8,CH>" if (bwd and a.cell == "lstm") else \
                    ("rnn_seq_bwd_bf16_kernel" if bwd else "rnn_seq_fwd_hw_kernel<1,4,4,8>")
            else:
                kname = ("rnn_seq_bwd" if bwd else "rnn_seq_fwd") + "_f32_kernel"
            traffic = None
            if a.cell == "lstm" and a.dtype == "bf16" and B == 48 and T == 1274:
                traffic = recorded_traffic("rnn_seq_bwd" if bwd else "rnn_seq_fwd")
            # whole-step algorithmic rates: BASELINE.md section 3 per-utterance figures x utterances/s of one GPU
            per_gpu = value / world
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "launches_per_step": f["n"] // a.steps,
                    "algorithmic_bytes_per_launch": f["bytes"] // f["n"],
                    "avg_launch_ms": round(f["ms"] / f["n"], 4),
                    "us_per_recurrent_step": round(f["ms"] * 1e3 / f["steps"], 3),
                    "whole_step": {"hbm_GBps": round(STEP_BYTES_PER_UTT * per_gpu / 1e9, 1),
                                   "frac_hbm": round(STEP_BYTES_PER_UTT * per_gpu / 1e9 / HBM_PEAK_GBS, 5),
                                   "mfma_TFLOPps": round(STEP_FLOPS_PER_UTT * per_gpu / 1e12, 2),
                                   "frac_mfma": round(STEP_FLOPS_PER_UTT * per_gpu / 1e12 / MFMA_PEAK_TFLOPS, 5),
                                   "per_utterance": "115 MB, 34.3 GFLOP (BASELINE.md section 3; lstm, T=1274, U=200)"},
                    "note": "sequential-dependency bound: every launch is T dependent steps (SURVEY 8(d)); since round 5 the W_hh packs and "
                            "exchange-state clears of a step's eight sweeps are ONE side-stream launch at the start of the step "
                            "(las_rnn_seq_prepare), no longer part of a launch's span; traffic = FETCH_SIZE x2 + WRITE_SIZE of the "
                            "recorded rocprofv3 --pmc passes when they were made from this kernel source, else null"}
        out = {
            "metric": "utterances/sec (train step) LibriSpeech-360 char-LAS @1/2/4/8 GPU; dev-clean WER",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype if a.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ("LibriSpeech-100 char LAS (BASELINE configs[1]): 3xpBLSTM-256 listener + 1x%s-512 "
                                    "speller, additive attention, MFCC-39, bucket B=%d T=%d, U=%d, V=30; full train "
                                    "step fwd+bwd+clip+Adam" % (a.cell.upper(), B, T, U)) if a.config == 1 else
                                   ("LibriSpeech-360 subword LAS (BASELINE configs[3], one rank of it): 3xpBLSTM-256 listener + "
                                    "1x%s-512 speller, location-aware attention K=201 C=10, V=5000, MFCC-39, bucket B=%d T=%d, "
                                    "U=%d; full train step fwd+bwd+clip+Adam" % (a.cell.upper(), B, T, U)),
                       "cell": a.cell, "per_gpu_batch": B, "global_batch": B * world, "frames": T, "dec_steps": U,
                       "parallelism": "dp%d" % world, "params": st.num_params()},
            "roofline": roof,
            # which cross-stream hand-overs the timed steps really ran (las.layers.VARIANTS; the GPU tests assert the same dict at this
            # geometry): chunked x-projections / chunked upstream gradients (-> rnn_seq_bwd_ks_kernel<...,CH=true>) / held weight gradients
            "schedule": dict(las.last_variants, aux_stream_priority=_hip.AUX_PRIORITY),
            "loss": round(loss, 4),
            "kernel_ms": {k: [round(v[0], 3), v[1] // a.steps] for k, v in sorted(per.items())},
        }
        out["scale_measured"] = world > 1           # (no multi-GPU node was available to rounds 1-3: until a line with n_gpus > 1 exists, the
                                                    #  data-parallel path is covered by gloo / RCCL-world-1 tests only)
        if a.config != 1:
            a.no_train_loop = a.no_decode = a.no_cpu_baseline = True       # side legs belong to the headline configuration
        if world == 1 and a.dtype == "bf16" and a.config == 1 and not a.no_train_loop:
            try:
                out["parity_mode"] = parity_mode_bench(dev, a, xs, ys)
            except Exception as e:
                out["parity_mode"] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if not a.no_side_legs:
                out.update(side_legs(dev, a, value))
            L.set_cell(a.cell)
            L.set_precision(a.dtype)
            V.reset_default_store(device=dev, seed=0)           # (the legs below build their own models)
            las = LAS(args, Listener, Speller, {})
            las.build_variables()
        if world == 1 and not a.no_train_loop:
            try:
                out["train_loop"] = train_loop_bench(las, dev, a, value)
            except Exception as e:                       # the train metric must still be printed
                out["train_loop"] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not a.no_decode:
            try:
                out["decode"] = decode_bench(dev, a.cell, a.dtype)
            except Exception as e:                       # the train metric must still be printed
                out["decode"] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cell)
            cpu_dec = out["cpu_baseline"].pop("decode", None)          # the CPU decode leg rides along in the same child
            if cpu_dec is not None and isinstance(out.get("decode"), dict):
                out["decode"]["cpu_baseline"] = cpu_dec
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dp is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
