"""Developer aid: is the forward bitwise reproducible run to run (ADMMNET_SPECTRAL=1)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_net_amd as A
from admm_net_amd import synth

dev = torch.device("cuda:0")
Nb, Nd, K, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 16, 16, 16384)
torch.manual_seed(0)
m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
ty, tb, ts, _ = synth.make_batch_device(B, Nb, Nd, seed=20260104, device=dev)
outs = []
for r in range(3):
    outs.append(m(ty, tb, ts).clone())
    print("run", r, "status", m.last_status, flush=True)
for r in (1, 2):
    d = (outs[r] != outs[0]).any(dim=1)
    idx = torch.nonzero(d).flatten()
    print(f"run {r} vs 0: {idx.numel()} signals differ", idx[:10].tolist(),
          float((outs[r] - outs[0]).abs().max()), flush=True)
