"""Phase stamps of las_beam_loop_step's kernel (utterance 0's workgroup) in a replayed decode step: build with
`make -C automatic-speech-recognition_amd/csrc ablf F=beam D=-DLAS_BEAM_STAMPS`, run with LAS_LIB_PATH=.../liblas_hip_ablf.so."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(ROOT, "tools", "probe_decode_stream.py")).read()
exec(src[:src.index("for NUTT in")])
NUTT = int(os.environ.get("NUTT", "16"))
utts = [synthetic_batch(1, 1274, 8, 30, seed=100 + k)[0] for k in range(NUTT)]
bs.decode_batch(None, utts[:2]); bs.decode_batch(None, utts)
torch.cuda.synchronize()
from las import _hip
out = (ctypes.c_ulonglong * 16)()
lib = ctypes.CDLL(_hip.LIB_PATH)
lib.las_dev_beam_stamps.argtypes = [ctypes.c_void_p]
assert lib.las_dev_beam_stamps(out) == 0
v = list(out)
order = [(1, "head loads issued"), (2, "projection multiplied"), (3, "partials exchanged"), (4, "logits assembled"), (7, "rank: keys made"),
         (8, "rank: wave selection (level 1)"), (9, "rank: barrier"), (10, "rank: selection among the survivors (level 2)"),
         (5, "ranked (winners ordered)"), (6, "bookkeeping done")]
print("beam_loop_kernel, %d utterances, the last step inside the bound; us since the workgroup's entry:" % NUTT)
for i, l in order:
    print("  %-48s %6.2f" % (l, (v[i] - v[0]) / 100.0))
