"""Phase stamps of the per-row attention kernel (dec_step_fwd_pf_kernel / pf_fwd_row, row 0) in a replayed decode step at 16 utterances (256 rows):
`make -C automatic-speech-recognition_amd/csrc ablf F=speller D=-DLAS_ROW_STAMPS S=rowst`, LAS_LIB_PATH=.../liblas_hip_rowst.so."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(ROOT, "tools", "probe_decode_stream.py")).read()
exec(src[:src.index("for NUTT in")])
NUTT = int(os.environ.get("NUTT", "16"))
utts = [synthetic_batch(1, 1274, 8, 30, seed=100 + k)[0] for k in range(NUTT)]
bs.decode_batch(None, utts[:2]); bs.decode_batch(None, utts)
torch.cuda.synchronize()
from las import _hip
out = (ctypes.c_ulonglong * 32)()
lib = ctypes.CDLL(_hip.LIB_PATH)
lib.las_dev_row_stamps.argtypes = [ctypes.c_void_p]
assert lib.las_dev_row_stamps(out) == 0
v = list(out)
labels = {0: "entry", 1: "bulk operands requested", 9: "(gates)", 2: "cell of the step before finished, state in LDS", 3: "query partials",
          4: "queries reduced", 5: "energies", 6: "softmax", 7: "context", 8: "input row written"}
print("pf_fwd_row (one row per workgroup), %d utterances, row 0; us since entry:" % NUTT)
for i in (1, 9, 2, 3, 4, 5, 6, 7, 8):
    print("  %-48s %6.2f" % (labels[i], (v[i] - v[0]) / 100.0))
