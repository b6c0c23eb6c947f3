"""CPU, world_size 2 over gloo: the batch-sharding protocol of admm_net_amd.sharded.

The HIP engine cannot run without a GPU, so the ranks drive ShardedForward with a test engine built
from the oracle's layer functions (tests may use the oracle).  What is verified is the product's host
logic: contiguous shards, ONE (sum, count) all-reduce per layer for scope='global', none for 'shard',
and the final gather in rank order -- 'global' must reproduce the reference evaluated on the whole
batch, 'shard' the reference evaluated on each sub-batch.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class OracleEngine:
    """Layer-at-a-time engine with the HipLayerEngine interface, computed by the oracle."""

    def __init__(self, sd, M, N, K, y, b, sigma):
        from oracle import admm_net_ref as R
        self.R, self.sd, self.M, self.N, self.K = R, R.cast_weights(sd, "f32"), M, N, K
        self.y, self.b, self.sigma = y, b, sigma.reshape(-1)
        self.calls = []

    def begin(self):
        B, n = self.y.shape[0], self.M * self.N + 1
        self.G = torch.zeros(B, n, n, dtype=torch.complex64)
        self.Z = torch.zeros(B, n, n, dtype=torch.complex64)

    def front(self, k):
        R = self.R
        self.phi = R.phi_layer(self.sd, k, self.y, self.b, self.G, self.Z)
        if k == self.K - 1:
            return None
        self.h = R.h_layer(self.sd, k, self.G, self.Z, self.sigma, self.M, self.N)
        self.G = R.g_layer(self.sd, k, self.phi, self.h, self.Z)
        corner = float(1.0 / (torch.nn.functional.softplus(self.sd[f"zLayers.{k}.lambda_param"]) ** 2 + R.EPS))
        self.Rm = self.G - R.block_matrix(self.phi, self.h, corner)
        self.rn = torch.linalg.norm(self.Rm, dim=(1, 2))
        return torch.tensor([float(self.rn.double().sum()), float(self.rn.numel())], dtype=torch.float64)

    def back(self, k, mean):
        self.calls.append(float(mean))
        arho = self.R.z_step(self.sd, k, self.rn, mean_norm=mean.to(torch.float32))
        self.Z = self.Z + arho.reshape(-1, 1, 1) * self.Rm

    def finish(self):
        return self.phi, None


def _worker(rank, world, port, path, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from admm_net_amd import sharded
    import admm_net_amd as A
    z = np.load(path)
    Nb, Nd, K = [int(v) for v in z["meta"][:3]]
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    y, b, s = torch.from_numpy(z["y"]), torch.from_numpy(z["b"]), torch.from_numpy(z["sigma"])
    lo, hi = sharded.shard_bounds(y.shape[0], world, rank)
    model = A.PhiEstADMMNet(M=Nb, N=Nd, num_layers=K)
    res = {}
    for scope in ("global", "shard"):
        engines = []

        def factory(yl, bl, sl):
            e = OracleEngine(sd, Nb, Nd, K, yl, bl, sl)
            engines.append(e)
            return e
        sf = sharded.ShardedForward(model, scope=scope, engine_factory=factory)
        phi, head = sf(y[lo:hi], b[lo:hi], s[lo:hi], gather=True)
        assert head is None and len(engines[0].calls) == K - 1
        res[scope] = phi.numpy()
        res[scope + "_means"] = np.array(engines[0].calls)
    if rank == 0:
        np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_global_equals_full_batch_and_shard_equals_split(tmp_path):
    path = os.path.join(ROOT, "tests", "golden", "split_batch_4x4_K3.npz")
    out = str(tmp_path / "res.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, path, out), nprocs=2, join=True)
    r = np.load(out)
    z = np.load(path)
    full, split = z["phi_full"], z["phi_split"]        # produced by the imported reference
    assert r["global"].shape == full.shape
    assert np.abs(r["global"] - full).max() < 5e-6 * np.abs(full).max() + 2e-6
    assert np.abs(r["shard"] - split).max() < 5e-6 * np.abs(split).max() + 2e-6
    assert np.abs(r["global"] - r["shard"]).max() > 1e-5      # the two scopes really differ
    assert not np.allclose(r["global_means"], r["shard_means"])
