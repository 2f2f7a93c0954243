"""scratch: where does tests/test_gpu_configs.py::test_config3_* spend its time on the GPU box?"""
import os, sys, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R + '/automatic-speech-recognition_amd', R]
import numpy as np, torch
import helpers
from helpers import make_args, synthetic_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
V = 5000
args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
                 mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=V, unit="subword", lr=1e-3, grad_clip=5.0, label_smoothing=True)
xs, ys = synthetic_batch(8, 1274, 24, V, seed=12, min_frac=0.9)
ys = (ys[0][:, :12], np.minimum(ys[1], 12)); ys[0][np.arange(8), ys[1] - 1] = 2
U = int(ys[1].max())
rng = np.random.RandomState(3); coins = rng.rand(U) < 0.6; sampled = rng.randint(3, V, size=(8, U)).astype(np.int32)
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
r = helpers.train_step_pair(args, "lstm", prec, xs, ys, seed=8, coins=coins, sampled=sampled)
pr.disable()
print(prec, "train_step_pair", time.time() - t0)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
