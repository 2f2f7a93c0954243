import sys, time, cProfile, pstats
import os; R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[R+'/tests',R+'/automatic-speech-recognition_amd',R]
import numpy as np, torch
from helpers import make_args, synthetic_batch, oracle_mode_for
from oracle import las_oracle as O
V=5000
args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
                 mode="loc", loc_kernel_size=201, loc_num_channels=10, vocab_size=V, unit="subword", lr=1e-3, grad_clip=5.0, label_smoothing=True)
prec=sys.argv[1]; nthr=int(sys.argv[2])
if nthr > 0:
    torch.set_num_threads(nthr)
if len(sys.argv) > 3:
    x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
print('threads', torch.get_num_threads(), 'affinity', len(os.sched_getaffinity(0)), 'cuda_init', len(sys.argv) > 3)
xs, ys = synthetic_batch(8, 1274, 24, V, seed=12, min_frac=0.9)
ys = (ys[0][:, :12], np.minimum(ys[1], 12)); ys[0][np.arange(8), ys[1]-1]=2
U=12
rng=np.random.RandomState(3); coins=rng.rand(U)<0.6; sampled=rng.randint(3,V,size=(8,U)).astype(np.int32)
p0 = O.init_params(args, seed=8, cell="lstm")
O.set_precision(*oracle_mode_for(args, prec))
po = O.to_torch(p0, requires_grad=True)
z = {k: torch.zeros_like(v) for k,v in po.items()}
z2 = {k: torch.zeros_like(v) for k,v in po.items()}
t0=time.time()
pr=cProfile.Profile(); pr.enable()
O.train_step(po, z, z2, 0, (torch.tensor(xs[0]), xs[1]), (torch.tensor(ys[0]), ys[1]), args, "lstm", coins=coins, sampled=torch.tensor(sampled))
pr.disable()
print(prec, nthr, "total", time.time()-t0)
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
