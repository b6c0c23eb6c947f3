"""Batch-sharded forward: one process per GPU, signals split across ranks.

Every signal is independent EXCEPT for one scalar per layer: the Z layer divides
each residual norm by the batch mean (admm_net.py:459).  ``scope='global'``
all-reduces (sum, count) of the residual norms once per layer -- two float64
over RCCL/xGMI -- so the sharded result equals the reference evaluated on the
whole batch; ``scope='shard'`` uses each rank's own mean (zero communication,
equals the reference evaluated on each sub-batch).  Nothing else crosses
ranks until the optional final all-gather of the outputs.

The layer engine is injected so the protocol can be exercised on CPU ranks
(gloo) in tests; the product engine is ``HipLayerEngine`` (C ABI layer-at-a-time
entry points of include/admmnet.h).
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class HipLayerEngine:
    """begin / front(k) / back(k, mean) / finish over admmnet_begin ... admmnet_finish."""

    def __init__(self, model, y, b, sigma):
        self.lib = _lib.load()
        self.m = model
        dev = model._compute_device(y)
        self.dev = dev
        D = model.M * model.N
        self.B = y.shape[0]
        self.y = y.detach().to(dev, torch.complex64).contiguous()
        self.b = b.detach().to(dev, torch.complex64).contiguous()
        self.sigma = sigma.detach().to(dev, torch.float32).reshape(-1).contiguous()
        with torch.cuda.device(dev):
            self.W = model.packed_weights(dev)
            self.ws = model.workspace(self.B, dev)
            self.status = torch.zeros(4, dtype=torch.int32, device=dev)
            self.sumcnt = torch.zeros(2, dtype=torch.float64, device=dev)
            self.mean = torch.zeros(4, dtype=torch.float32, device=dev)
            self.phi = torch.empty(self.B, D, dtype=torch.complex64, device=dev)
            self.head = (torch.empty(3, self.B, model.L, dtype=torch.float32, device=dev)
                         if model._HAS_HEAD else None)
        self.cfg = model.cfg()

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def begin(self):
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.admmnet_begin(ctypes.byref(self.cfg), self.B, _ptr(self.ws), self.ws.numel(),
                                              _ptr(self.status), self._stream()), "admmnet_begin")

    def front(self, k: int) -> torch.Tensor:
        """Runs layer k up to G; returns device float64 [2] = (local sum of r_b, local count)."""
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.admmnet_layer_front(ctypes.byref(self.cfg), _ptr(self.W), k, _ptr(self.y),
                                                    _ptr(self.b), _ptr(self.sigma), self.B, _ptr(self.ws),
                                                    _ptr(self.sumcnt), _ptr(self.status), self._stream()),
                       "admmnet_layer_front")   # (writes the pair: local sum, local count)
        return self.sumcnt

    def back(self, k: int, mean: torch.Tensor):
        with torch.cuda.device(self.dev):
            self.mean[0] = mean.to(torch.float32)
            _lib.check(self.lib.admmnet_layer_back(ctypes.byref(self.cfg), _ptr(self.W), k, self.B, _ptr(self.ws),
                                                   _ptr(self.mean), self._stream()), "admmnet_layer_back")

    def back_pair(self, k: int, sum_count: torch.Tensor):
        """ZLayer step from the (all-reduced) device float64 pair (sum of r_b, number of signals): the batch mean is formed
        on the device, no host arithmetic between the calls."""
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.admmnet_layer_back_pair(ctypes.byref(self.cfg), _ptr(self.W), k, self.B, _ptr(self.ws),
                                                        _ptr(sum_count), self._stream()), "admmnet_layer_back_pair")

    def finish(self):
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.admmnet_finish(ctypes.byref(self.cfg), _ptr(self.W), self.B, _ptr(self.ws),
                                               _ptr(self.phi), _ptr(self.head), self._stream()), "admmnet_finish")
            if self.m.check_status:
                bad = int(self.status[0].item())
                if bad:
                    raise _lib.AdmmNetError(f"eigensolver failed to converge on {bad} matrices")
        return self.phi, self.head


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous, as-even-as-possible split of ``total`` signals."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedForward:
    def __init__(self, model, scope: str = "global", group=None,
                 engine_factory: Optional[Callable] = None):
        if scope not in ("global", "shard"):
            raise ValueError("scope must be 'global' or 'shard'")
        self.model = model
        self.scope = scope
        self.group = group
        self.engine_factory = engine_factory or (lambda y, b, s: HipLayerEngine(model, y, b, s))

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    @torch.no_grad()
    def __call__(self, y_local, b_local, sigma_local, gather: bool = False):
        """Forward of this rank's shard.  Returns (phi, head) for the shard, or for the whole
        batch (rank order) when ``gather`` is set.  ``head`` is None for PhiEstADMMNet."""
        eng = self.engine_factory(y_local, b_local, sigma_local)
        K = self.model.num_layers
        world = self._world()
        eng.begin()
        for k in range(K):
            sc = eng.front(k)
            if k == K - 1:
                break
            if self.scope == "global" and world > 1:
                dist.all_reduce(sc, op=dist.ReduceOp.SUM, group=self.group)
            if hasattr(eng, "back_pair") and sc.is_cuda and sc.dtype == torch.float64:
                eng.back_pair(k, sc)            # (the mean from the pair on the device: csrc/zstep.hip)
            else:
                eng.back(k, sc[0] / sc[1])
        phi, head = eng.finish()
        # status words of the C ABI after the forward: [0] eigensolver failures, [1] matrix-layers the matrix-function route
        # handed to the eigensolver, [2] matrix-layers it evaluated itself, [3] of [1] those rejected by the model of f
        self.last_status = [int(v) for v in eng.status.tolist()] if hasattr(eng, "status") else [0, 0, 0, 0]
        if gather and world > 1:
            phi = self._gather(phi, dim=0)
            if head is not None:
                head = self._gather(head, dim=1)
        return phi, head

    def _gather(self, t, dim):
        if t.is_complex():   # collectives move real tensors
            return torch.view_as_complex(self._gather(torch.view_as_real(t).contiguous(), dim))
        world = self._world()
        sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([t.shape[dim]], dtype=torch.int64, device=t.device), group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad_shape = list(t.shape)
        pad_shape[dim] = mx
        buf = torch.zeros(pad_shape, dtype=t.dtype, device=t.device)
        buf.narrow(dim, 0, t.shape[dim]).copy_(t)
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf, group=self.group)
        return torch.cat([o.narrow(dim, 0, s) for o, s in zip(outs, sizes)], dim=dim)
