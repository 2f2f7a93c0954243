"""Ablation timing of the forward sweep: runs tools/bench_rnn.run(lstm) once per liblas_hip_abl<mask>.so found in lib/
(built by `make -C automatic-speech-recognition_amd/csrc abl ABL=<mask>`; results of those builds are wrong by construction)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(ROOT, "automatic-speech-recognition_amd", "lib", "liblas_hip_abl*.so")))
code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from las import _hip; _hip.LIB_PATH = sys.argv[1]; import bench_rnn; bench_rnn.run(1)"
for lib in [os.path.join(ROOT, "automatic-speech-recognition_amd", "lib", "liblas_hip.so")] + libs:
    out = subprocess.run([sys.executable, "-c", code % (os.path.join(ROOT, "tools"), os.path.join(ROOT, "automatic-speech-recognition_amd")), lib],
                         capture_output=True, text=True)
    print(os.path.basename(lib), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
