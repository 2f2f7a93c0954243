import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
import torch
from las import _hip
B, T, H = int(sys.argv[1]), int(sys.argv[2]), 256
G, cell, prec = 4, 1, 1
dev = "cuda"
g = torch.Generator().manual_seed(0)
io = torch.bfloat16
xp = (torch.randn(B, T, 2, G * H, generator=g) * 0.5).to(dev).to(io)
w0 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
w1 = (torch.randn(H, G * H, generator=g) * 0.05).to(dev)
out = torch.zeros(B, T, 2 * H, device=dev, dtype=io)
cst = torch.zeros(B, T, 2, H, device=dev, dtype=io)
dout = (torch.randn(B, T, 2 * H, generator=g) * 1e-3).to(dev).to(io)
gates = xp.clone()
_hip.rnn_seq_fwd(cell, prec, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst)
torch.cuda.synchronize()
print("fwd finite:", bool(torch.isfinite(gates.float()).all()), bool(torch.isfinite(out.float()).all()), bool(torch.isfinite(cst.float()).all()),
      "max|c|", float(cst.float().abs().max()))
db0 = torch.zeros(G * H, device=dev); db1 = torch.zeros(G * H, device=dev)
_hip.rnn_seq_bwd(cell, prec, B, T, H, gates, w0, w1, G * H, out, 2 * H, T * 2 * H, cst, dout, 2 * H, T * 2 * H, db_fw=db0, db_bw=db1)
torch.cuda.synchronize()
_hip.check_status()
gz = gates.float()
print("bwd finite:", bool(torch.isfinite(gz).all()), "max|dz|", float(gz[torch.isfinite(gz)].abs().max()), "db finite", bool(torch.isfinite(db0).all() and torch.isfinite(db1).all()))
bad = (~torch.isfinite(gz)).nonzero()
if len(bad):
    print("first bad (b,t,dir,col):", bad[:5].tolist(), "count", len(bad))
    print("bad t range dir0:", bad[bad[:, 2] == 0][:, 1].min().item() if (bad[:, 2] == 0).any() else None, "dir1:", bad[bad[:, 2] == 1][:, 1].max().item() if (bad[:, 2] == 1).any() else None)
