"""Developer diagnostic: where do the ~37 ms outliers of the single-signal latency loop come from?  Same model and inputs,
(a) calls back to back, (b) with a host-side pause between two calls like the scene synthesis of harness.time_net."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import admm_net_amd as A  # noqa: E402
from admm_net_amd import harness  # noqa: E402
torch.manual_seed(0)
m = A.PhiEstADMMNet(num_layers=5, M=10, N=10, L=3).eval()
rng = np.random.default_rng(0)
y, b, s = harness.demo_scene(rng)
ty = torch.from_numpy(y.flatten().reshape(1, -1)).to(torch.complex64)
tb = torch.from_numpy(b.reshape(1, -1)).to(torch.complex64)
ts = torch.from_numpy(np.asarray(s).reshape(1, -1)).to(torch.float32)
m(ty, tb, ts)
for name, pause in (("back to back", 0.0), ("1 ms pause", 0.001), ("5 ms pause", 0.005), ("20 ms pause", 0.02)):
    t = []
    for i in range(300):
        if pause:
            time.sleep(pause)
        t0 = time.perf_counter()
        m(ty, tb, ts)
        t.append(time.perf_counter() - t0)
    t = np.asarray(t)
    print(f"{name:14s} median {np.median(t) * 1e3:.3f} ms  mean {t.mean() * 1e3:.3f} ms  max {t.max() * 1e3:.2f} ms  runs > 10 ms: {(t > 0.01).sum()}")
# (c) the harness loop itself (fresh scene per run), with the host BLAS pool as it comes and limited to one thread
from threadpoolctl import threadpool_limits  # noqa: E402
for name, lim in (("harness loop", None), ("harness, 1 BLAS thread", 1)):
    if lim:
        with threadpool_limits(limits=lim):
            t = np.asarray(harness.time_net(m, 300, None, seed=1))[1:]
    else:
        t = np.asarray(harness.time_net(m, 300, None, seed=1))[1:]
    print(f"{name:24s} median {np.median(t) * 1e3:.3f} ms  mean {t.mean() * 1e3:.3f} ms  max {t.max() * 1e3:.2f} ms  runs > 10 ms: {(t > 0.01).sum()}")
