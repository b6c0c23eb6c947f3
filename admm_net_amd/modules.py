"""Drop-in ``ADMMNet`` / ``PhiEstADMMNet`` for /root/reference/admm_net.py.

Same constructor (``M, N, L=3, num_layers=10``), same ``forward(y, b, sigma)``
return values and the same ``state_dict`` key set / parameter names
(admm_net.py:724-816), so ``main_for_net.py``, ``test/test_time_net.py`` and
``load_state_dict`` of a reference checkpoint work unchanged.  The sub-modules
below are parameter containers registered in the reference's order (so
``torch.manual_seed(s)`` + construction yields the reference's initial
weights); all arithmetic happens in the fused HIP path behind the C ABI of
``include/admmnet.h``.  There is no CPU or eager fallback.

Inference (``.eval()`` or ``torch.no_grad()``) runs the fused forward.  In train mode with gradients
enabled the call goes through ``training.unrolled_forward`` instead: differentiable tensor operations
around the HIP eigensolver, with the reference's eigenvalue-only gradient.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import Cfg

HIDDEN = 128   # PeakSearchLayer hidden_dim (admm_net.py:496)
HEADS = 4


def _no_forward(self, *a, **k):
    raise NotImplementedError(
        f"{type(self).__name__} is a parameter container: the layer is executed inside the fused "
        "HIP forward of ADMMNet / PhiEstADMMNet (include/admmnet.h)")


class PhiLayer(nn.Module):
    """Parameters of admm_net.py:71-105."""
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.rho = nn.Parameter(torch.tensor(1.0))
        self.epsilon = epsilon
    forward = _no_forward


class HLayer(nn.Module):
    """Parameters of admm_net.py:108-205."""
    def __init__(self, M, N, epsilon=1e-8):
        super().__init__()
        self.M, self.N = M, N
        self.dim = M * N
        self.epsilon = epsilon
        self.rho = nn.Parameter(torch.tensor(1.0))
        self.projection_weight = nn.Parameter(torch.tensor(1.0))
        self.correction_net = nn.Sequential(nn.Linear(self.dim, 64), nn.ReLU(),
                                            nn.Linear(64, self.dim), nn.Tanh())
    forward = _no_forward


class GLayer(nn.Module):
    """Parameters of admm_net.py:208-354."""
    def __init__(self, M, N, epsilon=1e-8):
        super().__init__()
        self.M, self.N = M, N
        self.dim = M * N + 1
        self.epsilon = epsilon
        self.lambda_param = nn.Parameter(torch.tensor(0.1))
        self.rho = nn.Parameter(torch.tensor(1.0))
        self.threshold = nn.Parameter(torch.tensor(0.0))
        self.value_net = nn.Sequential(nn.Linear(1, 16), nn.ReLU(), nn.Linear(16, 1), nn.Sigmoid())
    forward = _no_forward


class ZLayer(nn.Module):
    """Parameters of admm_net.py:357-490 (step_adjust_net is unused by forward but is in the state_dict)."""
    def __init__(self, M, N, epsilon=1e-8):
        super().__init__()
        self.M, self.N = M, N
        self.dim_h = M * N
        self.dim_z = M * N + 1
        self.epsilon = epsilon
        self.rho = nn.Parameter(torch.tensor(1.0))
        self.lambda_param = nn.Parameter(torch.tensor(1.0))
        self.residual_scale_net = nn.Sequential(nn.Linear(3, 32), nn.ReLU(), nn.Linear(32, 1), nn.Sigmoid())
        self.step_adjust_net = nn.Sequential(nn.Linear(3, 8), nn.ReLU(), nn.Linear(8, 1), nn.Sigmoid())
    forward = _no_forward


class PeakSearchLayer(nn.Module):
    """Parameters of admm_net.py:494-630."""
    def __init__(self, M, N, L=3, hidden_dim=HIDDEN, num_heads=HEADS):
        super().__init__()
        if hidden_dim != HIDDEN or num_heads != HEADS:
            raise ValueError("the HIP head is built for hidden_dim=128, num_heads=4 (reference defaults)")
        self.M, self.N, self.L_max = M, N, L
        self.dim = M * N
        self.feature_extractor = nn.Sequential(nn.Linear(2 * self.dim, hidden_dim), nn.ReLU(),
                                               nn.Linear(hidden_dim, hidden_dim), nn.ReLU())
        tau_grid = torch.linspace(0, 1, M)
        f_grid = torch.linspace(-0.5, 0.5, N)
        tg, fg = torch.meshgrid(tau_grid, f_grid, indexing="ij")
        self.position_encoder = nn.Parameter(torch.stack([tg.flatten(), fg.flatten()], dim=1), requires_grad=True)
        self.position_projection = nn.Linear(2, hidden_dim)
        self.attention = nn.MultiheadAttention(embed_dim=hidden_dim, num_heads=num_heads, batch_first=True,
                                               dropout=0.1)
        self.peak_extractor = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                            nn.Linear(hidden_dim // 2, hidden_dim // 4), nn.ReLU(),
                                            nn.Linear(hidden_dim // 4, hidden_dim // 8), nn.ReLU())
        self.tau_regressor = nn.ModuleList([
            nn.Sequential(nn.Linear(hidden_dim // 8, 32), nn.ReLU(), nn.Linear(32, 1), nn.Sigmoid())
            for _ in range(L)])
        self.f_regressor = nn.ModuleList([
            nn.Sequential(nn.Linear(hidden_dim // 8, 32), nn.ReLU(), nn.Linear(32, 1), nn.Tanh())
            for _ in range(L)])
        self.confidence_net = nn.Sequential(nn.Linear(hidden_dim // 8, 16), nn.ReLU(), nn.Linear(16, 1),
                                            nn.Sigmoid())
    forward = _no_forward


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class _FusedBase(nn.Module):
    _HAS_HEAD = False

    def __init__(self, M, N, L=3, num_layers=10):
        super().__init__()
        self.num_layers = num_layers
        self.M, self.N, self.L = M, N, L
        self.phiLayers = nn.ModuleList([PhiLayer() for _ in range(num_layers)])
        self.hLayers = nn.ModuleList([HLayer(M, N) for _ in range(num_layers)])
        self.gLayers = nn.ModuleList([GLayer(M, N) for _ in range(num_layers)])
        self.zLayers = nn.ModuleList([ZLayer(M, N) for _ in range(num_layers)])
        if self._HAS_HEAD:
            self.peakSearchLayer = PeakSearchLayer(M, N, L)
        # execution knobs (not part of the reference API)
        self._chunk = 0                # signals per eigensolver chunk (0 = library default); see the `chunk` property
        self.check_status = True       # one D2H read per forward: raise if the eigensolver failed
        self._wcache = None
        self._ws = None

    @property
    def chunk(self) -> int:
        """Signals per eigensolver work chunk (0 = library default, 8192).  Setting it drops the cached workspace: its
        carve depends on the chunk size."""
        return self._chunk

    @chunk.setter
    def chunk(self, value: int):
        value = int(value)
        if value < 0:
            raise ValueError("chunk must be >= 0")
        if value != self._chunk:
            self._chunk = value
            self._ws = None

    # ---- weights -----------------------------------------------------------
    def cfg(self) -> Cfg:
        return Cfg(self.M, self.N, self.L, self.num_layers, int(self._HAS_HEAD), int(self.chunk), (ctypes.c_int32 * 2)(0, 0))

    def _raw_params(self):
        """Parameters in the raw order documented in include/admmnet.h."""
        out = []
        for k in range(self.num_layers):
            hl, gl, zl = self.hLayers[k], self.gLayers[k], self.zLayers[k]
            out += [self.phiLayers[k].rho, hl.rho, hl.projection_weight,
                    hl.correction_net[0].weight, hl.correction_net[0].bias,
                    hl.correction_net[2].weight, hl.correction_net[2].bias,
                    gl.lambda_param, gl.rho, gl.threshold,
                    gl.value_net[0].weight, gl.value_net[0].bias, gl.value_net[2].weight, gl.value_net[2].bias,
                    zl.rho, zl.lambda_param,
                    zl.residual_scale_net[0].weight, zl.residual_scale_net[0].bias,
                    zl.residual_scale_net[2].weight, zl.residual_scale_net[2].bias]
        if self._HAS_HEAD:
            p = self.peakSearchLayer
            out += [p.position_encoder, p.feature_extractor[0].weight, p.feature_extractor[0].bias,
                    p.feature_extractor[2].weight, p.feature_extractor[2].bias,
                    p.position_projection.weight, p.position_projection.bias,
                    p.attention.in_proj_weight, p.attention.in_proj_bias,
                    p.attention.out_proj.weight, p.attention.out_proj.bias]
            for i in (0, 2, 4):
                out += [p.peak_extractor[i].weight, p.peak_extractor[i].bias]
            for t in range(self.L):
                out += [p.tau_regressor[t][0].weight, p.tau_regressor[t][0].bias,
                        p.tau_regressor[t][2].weight, p.tau_regressor[t][2].bias,
                        p.f_regressor[t][0].weight, p.f_regressor[t][0].bias,
                        p.f_regressor[t][2].weight, p.f_regressor[t][2].bias]
            out += [p.confidence_net[0].weight, p.confidence_net[0].bias,
                    p.confidence_net[2].weight, p.confidence_net[2].bias]
        return out

    def packed_weights(self, device) -> torch.Tensor:
        """Host-pack (admmnet_pack_weights) and upload; cached until a parameter changes."""
        lib = _lib.load()
        params = self._raw_params()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if self._wcache is not None and self._wcache[0] == key:
            return self._wcache[1]
        cfg = self.cfg()
        raw = torch.cat([p.detach().to("cpu", torch.float32).reshape(-1) for p in params]).contiguous()
        nraw = lib.admmnet_raw_weight_count(ctypes.byref(cfg))
        if nraw != raw.numel():
            raise _lib.AdmmNetError(f"raw weight count mismatch: library {nraw}, module {raw.numel()}")
        packed = torch.empty(lib.admmnet_packed_weight_count(ctypes.byref(cfg)), dtype=torch.float32)
        _lib.check(lib.admmnet_pack_weights(ctypes.byref(cfg), _ptr(raw), _ptr(packed)), "admmnet_pack_weights")
        dev = packed.to(device)
        self._wcache = (key, dev)
        return dev

    def workspace(self, B: int, device) -> torch.Tensor:
        lib = _lib.load()
        cfg = self.cfg()
        need = lib.admmnet_workspace_bytes(ctypes.byref(cfg), B)
        if need < 0:
            _lib.check(-1, "admmnet_workspace_bytes")
        if self._ws is None or self._ws.numel() < need or self._ws.device != torch.device(device):
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    # ---- forward -------------------------------------------------------------
    @staticmethod
    def _compute_device(y: torch.Tensor) -> torch.device:
        if y.is_cuda:
            return y.device
        if not torch.cuda.is_available():
            raise _lib.AdmmNetError("no HIP device: the ADMM-Net forward runs only on the GPU (no CPU fallback)")
        return torch.device("cuda", torch.cuda.current_device())

    def _run(self, y, b, sigma):
        lib = _lib.load()
        dev = self._compute_device(y)
        D = self.M * self.N
        if y.dim() != 2 or y.shape[1] != D or b.shape != y.shape:
            raise ValueError(f"y, b must be [B, {D}] complex; got {tuple(y.shape)}, {tuple(b.shape)}")
        B = y.shape[0]
        yd = y.detach().to(dev, torch.complex64).contiguous()
        bd = b.detach().to(dev, torch.complex64).contiguous()
        sd = sigma.detach().to(dev, torch.float32).reshape(-1).contiguous()
        if sd.numel() != B:
            raise ValueError("sigma must have one entry per signal")
        with torch.cuda.device(dev):
            W = self.packed_weights(dev)
            ws = self.workspace(B, dev)
            phi = torch.empty(B, D, dtype=torch.complex64, device=dev)
            head = torch.empty(3, B, self.L, dtype=torch.float32, device=dev) if self._HAS_HEAD else None
            status = torch.zeros(4, dtype=torch.int32, device=dev)
            cfg = self.cfg()
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.admmnet_forward_f32(ctypes.byref(cfg), _ptr(W), _ptr(yd), _ptr(bd), _ptr(sd), B, _ptr(phi),
                                         _ptr(head), _ptr(ws), ws.numel(), _ptr(status), stream)
            _lib.check(rc, "admmnet_forward_f32")
            if self.check_status:
                # [0] eigensolver failures; with ADMMNET_SPECTRAL=1: [1] matrix-layers that took the eigensolver after all,
                # [2] matrix-layers evaluated as a matrix function (csrc/spectral.hip)
                self.last_status = status.tolist()
                bad = int(self.last_status[0])
                if bad:
                    raise _lib.AdmmNetError(f"eigensolver failed to converge on {bad} matrices")
        out_dev = y.device
        phi = phi.to(out_dev)
        if head is not None:
            head = head.to(out_dev)
        return phi, head


    # ---- training ----------------------------------------------------------
    def forward_autograd(self, y, b, sigma):
        """Differentiable forward (what ``forward`` does in train mode); usable in eval mode too, e.g. to
        take gradients without the attention dropout.  The parameters must live on the GPU."""
        from . import training
        _lib.load()                                   # fail loudly without the HIP library
        dev = self._compute_device(y)
        pdev = next(self.parameters()).device
        if pdev != dev:
            raise _lib.AdmmNetError(f"training runs on the GPU: parameters are on {pdev}, expected {dev} "
                                    "(move the model with .to(device) as train.py / trainPhi.py do)")
        out = training.unrolled_forward(self, y.to(dev), b.to(dev), sigma.to(dev))
        if isinstance(out, tuple):
            return tuple(o.to(y.device) for o in out)
        return out.to(y.device)

    def _wants_grad(self) -> bool:
        return self.training and torch.is_grad_enabled()


class PhiEstADMMNet(_FusedBase):
    """admm_net.py:724-764: K unrolled ADMM layers, returns phi [B, M*N] complex64."""
    _HAS_HEAD = False

    def forward(self, y, b, sigma):
        if self._wants_grad():
            return self.forward_autograd(y, b, sigma)
        phi, _ = self._run(y, b, sigma)
        return phi


class ADMMNet(_FusedBase):
    """admm_net.py:767-816: unrolled layers + PeakSearchLayer; returns (tau, f, confidences, phi)."""
    _HAS_HEAD = True

    def forward(self, y, b, sigma):
        if self._wants_grad():
            return self.forward_autograd(y, b, sigma)
        phi, head = self._run(y, b, sigma)
        return head[0], head[1], head[2], phi
