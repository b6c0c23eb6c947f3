"""Developer aid (GPU box): the forward with and without ADMMNET_SPECTRAL=1 against the f64 oracle.
Run once per setting of the environment (the switch is read once per process):
    ADMMNET_SPECTRAL=1 python tests/gpu_spectral_check.py"""
import faulthandler
import os
import sys
import time

faulthandler.dump_traceback_later(90, repeat=True)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import admm_net_amd as A
from admm_net_amd import synth
from oracle import admm_net_ref as R

torch.set_num_threads(8)   # (the box's many-core default makes the small CPU eigh calls of the oracle crawl)
dev = torch.device("cuda:0")
cases = [(10, 10, 10, 64, 0.0), (10, 10, 10, 64, 0.3), (8, 16, 8, 64, 0.3), (16, 16, 16, 16, 0.0), (16, 16, 16, 16, 0.3),
         (12, 16, 8, 16, 0.3), (4, 4, 6, 32, 0.3)]
if len(sys.argv) > 1:
    cases = [cases[int(a)] for a in sys.argv[1:]]
for Nb, Nd, K, B, pert in cases:
    sd = R.make_weights(Nb, Nd, K, seed=5, head=False, perturb=pert)
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=11)
    ty, tb, ts = torch.from_numpy(y), torch.from_numpy(b), torch.from_numpy(s)
    ref = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f64").numpy()
    ref32 = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32").numpy()
    m = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K)
    m.load_state_dict({k: v for k, v in sd.items()})
    m.eval()
    out = m(ty.to(dev), tb.to(dev), ts.to(dev)).cpu().numpy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m(ty.to(dev), tb.to(dev), ts.to(dev))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    sc = np.abs(ref).max()
    print(f"{Nb}x{Nd} K={K} B={B} perturb={pert}: err vs f64 oracle {np.abs(out - ref).max() / sc:.3e}  "
          f"(f32 oracle vs f64: {np.abs(ref32 - ref).max() / sc:.3e})  status={m.last_status}  {dt * 1e3:.1f} ms", flush=True)
