"""How far apart are the oracle's own f32 and bf16-operand modes on the bench architecture at full T (B=4, T=1274)?
Sets the floor for any bf16 parity tolerance: python tests/oracle_sensitivity.py rnn lstm  (CPU only, ~2 min)."""
import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, 'tests'), R, os.path.join(R, 'automatic-speech-recognition_amd')]
import torch, numpy as np
from helpers import make_args, synthetic_batch
from oracle import las_oracle as O
torch.set_num_threads(8)
args = make_args(enc_units=256, num_enc_layers=3, dec_units=512, num_dec_layers=1, embedding_size=128, attention_size=128,
              mode="add", lr=1e-3, grad_clip=5.0, label_smoothing=True, vocab_size=30)
xs, ys = synthetic_batch(4, 1274, 256, 30, seed=7, min_frac=0.834)
for cell in sys.argv[1:]:
    res={}
    for mode in ("f32","bf16"):
        O.set_precision(mode)
        p0 = O.init_params(args, seed=3, cell=cell)
        po = O.to_torch(p0, requires_grad=True)
        z = {k: torch.zeros_like(v) for k, v in po.items()}
        t0=time.time()
        out = O.train_step(po, z, {k: torch.zeros_like(v) for k, v in po.items()}, 0, (torch.tensor(xs[0]), xs[1]), (torch.tensor(ys[0]), ys[1]), args, cell)
        print(cell, mode, "time %.1fs loss %.5f"%(time.time()-t0, float(out[0])), flush=True)
        res[mode]=out
    O.set_precision("f32")
    a,b=res["f32"],res["bf16"]
    print(" logits diff %.3e alphas diff %.3e"%((a[1]-b[1]).abs().max(), (a[2]-b[2]).abs().max()))
    errs={n:((a[3][n]-b[3][n]).abs().max()/max(a[3][n].abs().max(),1e-3)).item() for n in a[3]}
    for n in sorted(errs, key=errs.get, reverse=True)[:6]: print("  %-70s %.3e  (|g|max %.3e)"%(n, errs[n], a[3][n].abs().max()))
