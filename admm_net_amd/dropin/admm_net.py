"""Shim: lets the reference scripts' ``from admm_net import PhiEstADMMNet`` pick up the MI355X path.

Put this directory first on PYTHONPATH.
"""
from admm_net_amd.modules import (ADMMNet, GLayer, HLayer, PeakSearchLayer, PhiEstADMMNet, PhiLayer,  # noqa: F401
                                  ZLayer)
