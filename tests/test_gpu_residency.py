"""Residency beside a co-resident kernel (VERDICT r4 "Next" #8).  The recurrent sweeps and the one-launch Speller loops exchange data between
workgroups that must all be resident at once; a data-parallel run puts a collective's kernels on the same device.  No 8-GPU node has been
available to any round, so this is the stand-in: a persistent "foreign" kernel (las_occupy: 256-thread workgroups, workgroup L on XCD L % 8,
some LDS, 32 or 64 VGPRs per lane -- the footprint of a collective's channels) is resident while train steps of the timed geometry run.

What must hold in every case: NOTHING HANGS, and a step either completes with the undisturbed bits or reports the documented status (never
silent garbage: las_clip_adam skips the update on the device).

* WHERE A COLLECTIVE REALLY RUNS in the data-parallel schedule (las/las.py): beside the Listener's BPTT sweeps at the earliest -- the early
  part of the bucket is issued behind the last sweep (layers.BEFORE_TAIL_HOOK), the token count beside the first forward sweep -- never
  beside a Speller loop.  Case "bptt": 32 workgroups x 16 KB of LDS x 64 VGPRs resident from the first BPTT sweep of a step to its end.
  The sweeps take 60 of the 256 CUs: the steps must complete, bit-identical.
* The Speller's loop kernels are the ones that need EVERY CU (8 x 32 workgroups of 1024 threads, <= 120 VGPRs; the forward rows keep 156 KB
  of a CU's 160 KB of LDS).  Case "whole step, registers only" (32 VGPRs, no LDS: what that register budget leaves room for --
  tests/test_cabi_and_host.py): must complete, bit-identical.  Case "whole step, 4 KB of LDS": a loop workgroup cannot be placed on a CU
  that has lost 4 KB -- the outcome is RECORDED: it must be the documented status error (or identical bits), not a hang and not garbage."""
import json
import os
import time

import pytest
import torch

from helpers import make_args, synthetic_batch

pytestmark = pytest.mark.gpu
B, T = 48, 1274


def _independent_streams(_hip, main):
    """Two streams for the foreign kernel and for the word that releases it, on hardware queues of their own: a stream is mapped to one of
    four queues by creation order, and in a process that has created many (the whole suite) a fresh one can share its queue with the
    launch stream, an auxiliary stream of the step or the other of the two -- the foreign kernel would then hold back the step's own
    kernels, or its release would queue up behind it, until its 20 s bound (seen once in a suite run).  Probed with the library's own
    overlap probe: a bounded waiter on one stream, the store it waits for on the other."""
    aux = list(_hip.aux_streams("cuda").values())

    def overlaps(a, b):
        with torch.cuda.stream(a):
            return _hip._probe_overlap("cuda", b) < 2.0
    picked = []
    for _ in range(24):
        s = torch.cuda.Stream()
        if all(overlaps(q, s) for q in [main] + aux[:2] + picked):
            picked.append(s)
            if len(picked) == 2:
                return picked
    pytest.skip("no two streams with hardware queues of their own in this process")


def _steps(n, foreign=None, window="step", fallback=False):
    from las import _hip, layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
                     mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30)
    xs, ys = synthetic_batch(B, T, 256, 30, seed=0, min_frac=0.834)
    L.set_cell("lstm"); L.set_precision("bf16")
    st = V.reset_default_store(device="cuda"); st.load(O.init_params(args, seed=0, cell="lstm"))
    las = LAS(args, Listener, Speller, {})
    las.train(xs, ys)                                   # (workspaces, shadows, registered sweeps)
    torch.cuda.synchronize()
    if fallback:                                        # the steps below on las.layers.fallback_schedule: what LAS._recover re-runs lost steps on
        las._fallback_until = 10 ** 9
    lib = _hip.lib()
    words = torch.zeros(2 * (n + 1), dtype=torch.int32, device="cuda")  # per launch of the foreign kernel: [stop, resident]
    main = torch.cuda.current_stream()
    other, third = _independent_streams(_hip, main)
    launched = [0]

    def occupy():
        w = words[2 * launched[0]:]
        launched[0] += 1
        nwg, lds, vg = foreign
        _hip.check(lib.las_occupy(_hip.p(w), _hip.p(w[1:]), nwg, lds, vg, 20000, _hip.stream()), "las_occupy")

    def release(k, after_main):
        if after_main:
            third.wait_stream(main)
        with torch.cuda.stream(third):
            _hip.set_word(words[2 * k:], 1)

    err, losses = None, []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    real_bwd = _hip.rnn_seq_bwd
    try:
        if foreign is not None and window == "step":
            with torch.cuda.stream(other):
                occupy()
            t0 = time.time()
            while int(words[1].item()) < foreign[0]:    # every foreign workgroup is on the machine before the steps start
                assert time.time() - t0 < 10, "the foreign kernel did not become resident"
        if foreign is not None and window == "bptt":
            state = {"armed": False}

            def bwd(*a, **k):                           # in front of a step's FIRST BPTT sweep: the foreign kernel starts behind everything
                if state["armed"]:                      # enqueued so far (both Speller loops) and stays until the step's last kernel
                    state["armed"] = False
                    other.wait_stream(main)
                    with torch.cuda.stream(other):
                        occupy()
                return real_bwd(*a, **k)
            _hip.rnn_seq_bwd = bwd
        e0.record()
        for i in range(n):
            if foreign is not None and window == "bptt":
                state["armed"] = True
            losses.append(las.train(xs, ys)[0])
            if foreign is not None and window == "bptt":
                release(i, after_main=True)
        e1.record()
        main.synchronize()
        las.check_status()
    except RuntimeError as e:
        err = str(e)
    finally:
        _hip.rnn_seq_bwd = real_bwd
        for k in range(n + 1):
            release(k, after_main=False)                # (whatever is still resident)
        torch.cuda.synchronize()
        try:
            _hip.check_status()
        except RuntimeError as e:
            err = err or str(e)
    ms = e0.elapsed_time(e1) / n if err is None else float("nan")
    _steps.recovered = las.recovered_steps
    return [float(v) for v in losses], st.flat.clone(), ms, err


def test_train_steps_beside_a_resident_foreign_kernel():
    base_l, base_p, base_ms, err = _steps(3)
    assert err is None
    rec = {"test": "residency", "undisturbed_ms_per_step": round(base_ms, 3)}
    must = (("bptt_32wg_64vgpr_16KB", (32, 16384, 64), "bptt"),          # a collective where the data-parallel schedule puts it
            ("step_8wg_32vgpr_noLDS", (8, 0, 32), "step"),               # registers only, one per XCD, the whole step
            ("step_32wg_32vgpr_noLDS", (32, 0, 32), "step"))
    for name, foreign, window in must:
        l, p, ms, err = _steps(3, foreign, window)
        rec[name] = {"ms_per_step": None if err else round(ms, 3), "status": err}
        assert err is None, (name, err)
        assert l == base_l and torch.equal(p, base_p), name
        assert ms < 2.0 * base_ms, (name, ms, base_ms)          # (measured 1.06-1.15 x; the bar only excludes a stalled schedule)
    # Footprints the Speller's loop kernels cannot share a CU with (round 5: status 3, the host raised and training stopped).  Round 6: the
    # device still skips the update of the step whose loop timed out, but LAS.train RE-RUNS the lost steps on the fall-back schedule
    # (per-step Speller launches, no cross-stream hand-overs) and stays on it: the three steps must complete, with the bits of three steps
    # run on that schedule on an idle device.  (Losses are compared through the parameters: a lost step's returned loss is garbage.)
    fb_l, fb_p, fb_ms, err = _steps(3, fallback=True)
    assert err is None
    rec["fallback_schedule_ms_per_step"] = round(fb_ms, 3)
    assert max(abs(a - b) for a, b in zip(fb_l, base_l)) < 2e-3 * max(1.0, abs(base_l[0]))        # same model, other kernels
    for name, foreign in (("step_8wg_32vgpr_4KB", (8, 4096, 32)), ("step_8wg_64vgpr_noLDS", (8, 0, 64))):
        l, p, ms, err = _steps(3, foreign, "step")
        rec[name] = {"status": err[:160] if err else None, "recovered_steps": _steps.recovered}
        assert err is None, (name, err)
        if _steps.recovered:
            assert torch.equal(p, fb_p), name           # every step was re-run (or run) on the fall-back schedule
        else:
            assert l == base_l and torch.equal(p, base_p), name         # (the loop kernels found room after all)
    print("residency:", json.dumps(rec))
    path = os.environ.get("LAS_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


def test_a_lost_step_is_rerun_on_the_fallback_schedule_and_training_continues():
    """LAS._recover without a neighbour: the status word is set by hand in front of the third of five steps (what a timed-out loop kernel
    does), so the device skips that step's update and -- the word is sticky -- the two behind it; the host notices at a later poll or at
    check_status, finds the first lost step from the device's count of applied updates (las_clip_adam `applied`), re-runs the three steps
    from the batches it kept and stays on the fall-back schedule.  The parameters must equal, bit for bit, a run whose last three steps
    were put on that schedule from the start; the step counter, Adam's bias correction and the sampling seeds follow the re-run."""
    import warnings
    from las import _hip, layers as L, variables as V
    from las.las import LAS, Listener, Speller
    from oracle import las_oracle as O
    # (64 units, 8 rows of a 16-row tile, on-device sampling: the geometry at which this test found the forward sweep's duplicate stores of
    #  rows past the end of the batch -- csrc/rnn_seq.hip, DESIGN section 8 -- which made the second step of any run differ by 2e-6)
    args = make_args(enc_units=64, num_enc_layers=2, dec_units=128, num_dec_layers=1, embedding_size=64, attention_size=64, mode="add",
                     lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30, scheduled_sampling=True, warmup_step=0, max_step=8)
    batches = [synthetic_batch(8, 96, 24, 30, seed=40 + k, min_frac=0.8) for k in range(5)]
    p0 = O.init_params(args, seed=2, cell="lstm")

    def run(lose):
        L.set_cell("lstm"); L.set_precision("bf16")
        st = V.reset_default_store(device="cuda"); st.load(p0)
        las = LAS(args, Listener, Speller, {})
        for k, (xs, ys) in enumerate(batches):
            if k == 2:
                torch.cuda.synchronize()
                if lose:
                    _hip.status_word("cuda")[0] = 3          # "a Speller loop of this step timed out"
                else:
                    las._fallback_until = 10 ** 9            # the reference run: steps 2.. on the fall-back schedule
            xd = (torch.tensor(xs[0], device="cuda"), xs[1])     # device tensors the caller overwrites afterwards (a feeder's ring slot)
            yd = (torch.tensor(ys[0], device="cuda"), ys[1])
            las.train(xd, yd)
            xd[0].fill_(7.0); yd[0].fill_(3)
        with warnings.catch_warnings(record=True) as wr:
            warnings.simplefilter("always")
            las.check_status()
        torch.cuda.synchronize()
        return st.flat.clone(), st.global_step, las.recovered_steps, las.speller._scheduled_sampling()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        f_ref, gs_ref, rec_ref, rate_ref = run(False)
        f_new, gs_new, rec_new, rate_new = run(True)
    assert rec_ref == 0 and rec_new in (1, 2, 3), (rec_ref, rec_new)        # (1-3: depends on which poll saw the word first; the rest ran on the fall-back schedule anyway)
    assert gs_ref == gs_new == 5 and rate_ref == rate_new
    assert torch.equal(f_ref, f_new), "max |d theta| %.3e" % (f_ref - f_new).abs().max().item()
