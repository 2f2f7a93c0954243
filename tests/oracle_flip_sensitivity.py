"""What does a handful of flipped bf16 roundings do to the outputs?  The oracle (bf16-operand mode) is run on inputs
perturbed by 1e-7 .. 1e-5 relative: below ~1e-6 the perturbation is rounded away, at 1e-5 a few inputs land on the other
side of a bf16 boundary.  lstm cells damp such a flip (logits move by ~4e-4), the reference tanh BasicRNNCell does not
(~8e-3): this is the floor of any bf16 parity tolerance for the rnn cell.  python tests/oracle_flip_sensitivity.py"""
import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(R, "tests"), R, os.path.join(R, "automatic-speech-recognition_amd")]
import torch, numpy as np
from helpers import make_args, synthetic_batch
from oracle import las_oracle as O
for cfg in [("rnn", "add", 1, 128, 64), ("lstm","add",1,64,64)]:
    cell, mode, NL, H, D = cfg
    args = make_args(enc_units=H, num_enc_layers=2, dec_units=D, num_dec_layers=NL, embedding_size=32,
                     attention_size=32, mode=mode, loc_kernel_size=11, loc_num_channels=3, lr=1e-3, grad_clip=5.0)
    xs, ys = synthetic_batch(5, 37, 9, 30, seed=H + D)
    res=[]
    for eps in (0.0, 1e-7, 1e-6, 1e-5):
        O.set_precision("bf16", "bf")
        p0 = O.init_params(args, seed=11, cell=cell)
        po = O.to_torch(p0, requires_grad=True)
        z = {k: torch.zeros_like(v) for k, v in po.items()}
        x = torch.tensor(xs[0]); x = x*(1+eps*torch.randn(x.shape, generator=torch.Generator().manual_seed(1)))
        res.append(O.train_step(po, z, {k: torch.zeros_like(v) for k, v in po.items()}, 0, (x, xs[1]), (torch.tensor(ys[0]), ys[1]), args, cell))
    O.set_precision("f32")
    for i,eps in enumerate((1e-7,1e-6,1e-5)):
        print(cfg, "input perturbation %g -> logits change %.3e alphas %.3e"%(eps, (res[0][1]-res[i+1][1]).abs().max(), (res[0][2]-res[i+1][2]).abs().max()))
