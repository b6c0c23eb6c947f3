"""MI355X-native ADMM-Net forward (gfx950 HIP behind a C ABI).

Public surface mirrors the reference's admm_net.py for the forward hot path:
``ADMMNet`` and ``PhiEstADMMNet`` (same ctor / forward / state_dict keys).
Importing this package never imports the CPU oracle.
"""
from .modules import ADMMNet, PhiEstADMMNet, PeakSearchLayer, PhiLayer, HLayer, GLayer, ZLayer  # noqa: F401

__all__ = ["ADMMNet", "PhiEstADMMNet"]
