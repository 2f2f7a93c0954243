import os, sys
sys.path.insert(0, "automatic-speech-recognition_amd")
import torch
from las import _hip
def run(prec, T, B=48, H=256):
    G=4; GH=G*H
    g = torch.Generator().manual_seed(0)
    xp = (torch.randn(B, T, 2, GH, generator=g) * 0.8).cuda()
    lim = (6.0 / (H + GH)) ** 0.5 * 1.5
    w0 = ((torch.rand(H, GH, generator=g) * 2 - 1) * lim).cuda(); w1 = ((torch.rand(H, GH, generator=g) * 2 - 1) * lim).cuda()
    dout = torch.randn(B, T, 2 * H, generator=g).cuda()
    out = torch.zeros(B, T, 2 * H, device="cuda"); cst = torch.zeros(B, T, 2, H, device="cuda")
    gates = xp.clone()
    _hip.rnn_seq_fwd(1, prec, B, T, H, gates, w0, w1, GH, out, 2 * H, T * 2 * H, cst)
    _hip.rnn_seq_bwd(1, prec, B, T, H, gates, w0, w1, GH, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H)
    return gates.double().cpu()
for T in (40, 300, 1274):
    a = run(1, T); b = run(0, T)
    print("T=%d  |dZ_bf16 - dZ_f32| / |dZ_f32| = %.4f  (max abs %.4f, ref max %.3f)" % (T, ((a-b).norm()/b.norm()).item(), (a-b).abs().max().item(), b.abs().max().item()))
