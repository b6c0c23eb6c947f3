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
