"""Phase stamps of the beam search's shared attention rows (dec_beam_rows4_kernel, workgroup 0) in a replayed step at 64 utterances:
build with `make -C automatic-speech-recognition_amd/csrc ablf F=speller D=-DLAS_ROW_STAMPS S=rowst`, run with LAS_LIB_PATH=.../liblas_hip_rowst.so."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(ROOT, "tools", "probe_decode_stream.py")).read()
exec(src[:src.index("for NUTT in")])
NUTT = int(os.environ.get("NUTT", "64"))
utts = [synthetic_batch(1, 1274, 8, 30, seed=100 + k)[0] for k in range(NUTT)]
bs.decode_batch(None, utts[:2]); bs.decode_batch(None, utts)
torch.cuda.synchronize()
from las import _hip
out = (ctypes.c_ulonglong * 32)()
lib = ctypes.CDLL(_hip.LIB_PATH)
lib.las_dev_row_stamps.argtypes = [ctypes.c_void_p]
assert lib.las_dev_row_stamps(out) == 0
v = list(out)
labels = ["entry", "shared operands requested (Ws, keys, states)", "states in LDS", "query partials", "queries reduced (encoder rows requested)",
          "energies", "softmax", "contexts (partials)", "input rows written"]
print("dec_beam_rows4_kernel, %d utterances, workgroup 0; us since entry:" % NUTT)
for i, l in enumerate(labels):
    print("  %-48s %6.2f" % (l, (v[i] - v[0]) / 100.0))
