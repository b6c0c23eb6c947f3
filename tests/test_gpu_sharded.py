"""GPU (-m gpu): the sharded forward with REAL process groups -- two ranks sharing cuda:0 over gloo
(one-GPU box; RCCL needs one device per rank).  Checks that scope='global' on two shards reproduces
the single-process forward of the whole batch bit for bit except for the fp64 -> fp32 batch mean
(identical sum order is not guaranteed), and that the gathered outputs come back in rank order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import admm_net_amd as A
    from admm_net_amd import sharded, synth
    dev = torch.device("cuda:0")
    Nb, Nd, K, B = 5, 6, 4, 26
    torch.manual_seed(11)
    m = A.ADMMNet(M=Nb, N=Nd, num_layers=K).eval()
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=3)
    lo, hi = sharded.shard_bounds(B, world, rank)
    args = [torch.from_numpy(v[lo:hi]).to(dev) for v in (y, b, s)]
    phi, head = sharded.ShardedForward(m, scope="global")(*args, gather=True)
    if rank == 0:
        full = m(*[torch.from_numpy(v).to(dev) for v in (y, b, s)])
        np.savez(out, phi=phi.cpu().numpy(), head=head.cpu().numpy(), phi_full=full[3].cpu().numpy(),
                 tau_full=full[0].cpu().numpy(), conf_full=full[2].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu_global_scope(tmp_path):
    out = str(tmp_path / "r.npz")
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert r["phi"].shape == r["phi_full"].shape and r["head"].shape == (3,) + r["tau_full"].shape
    assert np.abs(r["phi"] - r["phi_full"]).max() <= 2e-6 * np.abs(r["phi_full"]).max()
    assert np.abs(r["head"][0] - r["tau_full"]).max() < 1e-6 and np.abs(r["head"][2] - r["conf_full"]).max() < 1e-6


@pytest.mark.timeout(600)
def test_bench_multi_gpu_entry_point_spawns_its_ranks():
    """`python bench.py --gpus 2` from a plain call (what the driver's scaling tier does with N > 1 when it is not
    already under torch.distributed.run): the parent starts the ranks as child processes before touching the GPU and
    relays rank 0's single JSON line.  Two ranks share cuda:0 over gloo here (RCCL wants one device per rank)."""
    import json
    import subprocess
    env = dict(os.environ, ADMMNET_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg3", "--batch", "64",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=540, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 128 and line["scaling"] == "weak"
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert line["roofline"]["profile_dropped_launches"] == 0
    assert line["cpu_baseline"] is None            # rank 0 at N = 1 only


@pytest.mark.timeout(600)
def test_bench_accounts_for_every_launch_of_the_timed_region():
    """The per-class HIP-event sums must cover the timed region at a step count that used to overflow the event pool
    (8192 pairs, ~600 scopes per cfg3 step): cfg2 with two chunks per layer records ~90 scopes per step, so 120 steps is past the old pool size."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--batch", "16384", "--steps",
                        "120", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=540,
                       cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    roof = line["roofline"]
    assert roof["profile_dropped_launches"] == 0
    assert set(roof["kernel_ms_per_step"]) == {"prep", "tridiag", "trideig", "backtransform", "rebuild", "zstep", "head",
                                               "spectrum", "gfunction"}
    # every launch is in the sums: they cover the wall time of the timed region up to launch gaps
    assert 0.85 <= roof["kernel_ms_sum_over_ms_per_step"] <= 1.03, roof
