"""Where the parity-mode step's launch stream idles: does the host stall (device allocations, synchronisations) inside a step?
Prints, per step, the caching allocator's device-allocation counters and the host time of the step's enqueue, then the host-side
time stamps of the phases (no device synchronisation inside)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
import bench
from las import layers as L, variables as V
from las.las import LAS, Listener, Speller

dev = torch.device("cuda:0")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synthetic_batch
prec = os.environ.get("PROBE_PREC", "f32")
L.set_cell(os.environ.get("PROBE_CELL", "lstm"))
L.set_precision(prec)
V.reset_default_store(device=dev, seed=0)
args = bench.bench_args("lstm", 1)
las = LAS(args, Listener, Speller, {})
las.build_variables()
xs, ys = synthetic_batch(48, 1274, 256, args.vocab_size, seed=0, min_frac=0.834)
xs = (torch.tensor(xs[0], device=dev), xs[1])
ys = (torch.tensor(ys[0], device=dev), ys[1])
for _ in range(2):
    las.train(xs, ys)
torch.cuda.synchronize()
for s in range(4):
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    las.train(xs, ys)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print("step %d: host enqueue %.2f ms, device done after %.2f ms; device allocs +%d frees +%d retries +%d; reserved %.2f GB allocated peak %.2f GB" % (
        s, (t1 - t0) * 1e3, (t2 - t0) * 1e3, st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"],
        st1["num_alloc_retries"] - st0["num_alloc_retries"], st1["reserved_bytes.all.current"] / 2**30, st1["allocated_bytes.all.peak"] / 2**30), flush=True)

for mode in ("back-to-back", "synchronised after every step", "back-to-back"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(10):
        las.train(xs, ys)
        if mode.startswith("sync"):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%-32s %.3f ms per step" % (mode, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
las.check_status()
