"""Phase stamps of the wide Speller path's attention kernels (speller_wide.h; workgroup (0, 0), the last launch of each) in one train step of
run.sh's recipe at B = 48, T = 1274:
`make -C automatic-speech-recognition_amd/csrc ablf F=speller D=-DLAS_ROW_STAMPS S=rowst`, LAS_LIB_PATH=.../liblas_hip_rowst.so."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
import bench
CELL = os.environ.get("CELL", "rnn")
out = bench.side_step_bench("cuda:0", CELL, "bf16", "run_sh", 48, 1274, steps=2, warmup=2)
print("run.sh recipe, %s cells: %.3f ms per step; Speller kernels %s" % (CELL, out["ms_per_step"], out.get("speller_kernels")))
from las import _hip
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(_hip.LIB_PATH)
lib.las_dev_wide_stamps.argtypes = [ctypes.c_void_p]
assert lib.las_dev_wide_stamps(buf) == 0
v = list(buf)
SECTIONS = [
    ("wide_energy_kernel", 0, ["q / alpha_{t-1} / filter / Wf staged", "conv of alpha_{t-1} (8 tap slices + sum)", "f kept for the gradient loop",
                                "energies of the slice", "slice max / sum of exp, statistics stored"]),
    ("wide_context_kernel", 10, ["softmax over all frames -> LDS", "context columns (partials)", "partials summed, row written"]),
    ("wide_dalpha_kernel", 20, ["d context staged", "d alpha of the slice's frames", "alpha . d alpha summed, stored"]),
    ("wide_energy_bwd_kernel", 30, ["q / Wf staged", "f rows, d energy of the slice", "tanh recomputed, dq / du / d f partials",
                                     "partials through LDS, stored", "d f stored"]),
    # (the reverse loop's LAST launch, step 0, has no conv transpose: its stamps 42.. are those of the launch before, printed relative to that launch's 41)
    ("wide_dq_kernel", 40, ["dq / du summed over the slices", "d f rows + filter staged", "conv transpose (tap slices)", "summed, stored"]),
]
for name, base, labels in SECTIONS:
    print("%s; us since its first stamp:" % name)
    for i, lab in enumerate(labels):
        a, b = v[base + i], v[base + i + 1]
        if a and b and b >= a and a >= v[base]:
            print("  %-52s %6.2f  (+%.2f)" % (lab, (b - v[base]) / 100.0, (b - a) / 100.0))
        elif a and b and b >= a:
            print("  %-52s         (+%.2f, in the launch before)" % (lab, (b - a) / 100.0))
        elif a and b:
            print("  %-52s         (stale: stamped by an earlier launch)" % lab)
