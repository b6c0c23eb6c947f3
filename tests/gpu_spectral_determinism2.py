"""Developer aid: which elements of G differ run to run after the first matrix-function layer?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_net_amd as A
from admm_net_amd import sharded, synth

dev = torch.device("cuda:0")
Nb, Nd, K, B = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 16, 4, 8192)
n = Nb * Nd + 1
torch.manual_seed(0)
m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K).eval()
ty, tb, ts, _ = synth.make_batch_device(B, Nb, Nd, seed=20260104, device=dev)
snaps = []
for r in range(int(os.environ.get('RUNS', '6'))):
    eng = sharded.HipLayerEngine(m, ty, tb, ts)
    eng.ws.zero_()
    eng.begin()
    sc = eng.front(0)
    eng.back(0, sc[0] / sc[1])
    sc = eng.front(1)
    torch.cuda.synchronize()
    G = eng.ws[: B * n * n * 8].view(torch.float32).view(B, n, n, 2).clone()
    snaps.append(G)
    print("run", r, eng.status.tolist(), flush=True)
tril = torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev))
for r in range(1, len(snaps)):
    d = ((snaps[r] != snaps[0]).any(dim=-1)) & tril
    mats = torch.nonzero(d.flatten(1).any(dim=1)).flatten()
    print(f"run {r}: {mats.numel()} matrices differ", mats[:8].tolist())
    for b in mats[:4].tolist():
        ij = torch.nonzero(d[b])
        print("  matrix", b, "elements", ij.shape[0], "rows", sorted(set(ij[:, 0].tolist()))[:12], "cols",
              sorted(set(ij[:, 1].tolist()))[:12],
              "max abs diff", float((snaps[r][b] - snaps[0][b]).abs().max()),
              "value", snaps[0][b][ij[0, 0], ij[0, 1]].tolist(), snaps[r][b][ij[0, 0], ij[0, 1]].tolist())
