"""Which torch streams get a hardware queue of their own?  (round 4: VERDICT r3 weak #1)

The cross-stream hand-overs of the train step (x-projection chunks, backward hand-over, held side stream) need kernels of the
launch stream and of an auxiliary stream to RUN AT THE SAME TIME.  HIP multiplexes streams onto a few hardware queues; two
streams that share one serialise.  This tool measures, for several ways of creating the auxiliary streams, whether a bounded
waiter on stream i sees a store made on stream j while it waits (las_wait_word / las_set_word of liblas_hip.so).

  python tools/probe_streams.py            # runs every scenario in a child process, prints one line each
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))


def child(order, prio, nother):
    import torch
    from las import _hip
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    lib = _hip.lib()

    def touch(s):
        w = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.stream(s):
            _hip.check(lib.las_set_word(_hip.p(w), 1, _hip.stream()))
        torch.cuda.synchronize()

    def mk_aux():
        ss = [torch.cuda.Stream(priority=prio) for _ in range(3)]
        for s in ss:
            touch(s)
        return ss

    def mk_other():
        ss = [torch.cuda.Stream() for _ in range(nother)]
        for s in ss:
            touch(s)
        return ss

    if order == "aux_first":
        aux = mk_aux()
        other = mk_other()
    else:
        other = mk_other()
        aux = mk_aux()

    def overlap(sa, sb):
        """ms a bounded (4 ms) waiter on sa needs when the store is issued on sb right behind it"""
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            e0.record()
            _hip.check(lib.las_wait_word(_hip.p(flag), 1, 4000, _hip.stream()))
            e1.record()
        with torch.cuda.stream(sb):
            _hip.check(lib.las_set_word(_hip.p(flag), 1, _hip.stream()))
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    main = torch.cuda.current_stream()
    names = ["main"] + ["aux%d" % i for i in range(3)] + ["o%d" % i for i in range(nother)]
    streams = [main] + aux + other
    row_main = [round(overlap(main, s), 2) for s in streams[1:]]
    rows_aux = [[round(overlap(a, s), 2) for s in streams if s is not a] for a in aux]
    # a second pass after all of them have been busy
    row_main2 = [round(overlap(main, s), 2) for s in streams[1:]]
    print(json.dumps({"order": order, "prio": prio, "nother": nother, "names": names, "main_vs": row_main, "main_vs_again": row_main2,
                      "aux_vs": rows_aux}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    scen = []
    for env in ({}, {"GPU_MAX_HW_QUEUES": "8"}, {"GPU_MAX_HW_QUEUES": "16"}, {"DEBUG_HIP_DYNAMIC_QUEUES": "0"}, {"DEBUG_HIP_DYNAMIC_QUEUES": "1"}):
        for order in ("aux_first", "aux_last"):
            for prio in (0, -1):
                scen.append((env, order, prio))
    for env, order, prio in scen:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", order, str(prio), "9"], env=e, capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps(env), out[-1] if out else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:]), flush=True)
