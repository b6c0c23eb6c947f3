"""Timing harnesses and checkpoint I/O in the reference's own formats.

* ``time_net`` / ``time_admm`` are the MI355X-path equivalents of /root/reference/test/test_time_net.py:10-137 and
  /root/reference/test/test_time_admm.py:7-110: the same 10 x 10 demo scene with fresh noise per run, one
  ``time.perf_counter`` bracket around one call, ``np.savetxt`` of the list -- so the files drop into
  /root/reference/results/plot_compute_time.py next to ``time.txt`` / ``time_net.txt`` / ``time_net_5.txt``.
* ``save_checkpoint`` / ``load_checkpoint`` write and read the dict of /root/reference/train.py:306-314
  (``epoch, model_state_dict, optimizer_state_dict, scheduler_state_dict, best_val_loss, config, history``) that
  ``main_for_net.py:100-101`` and ``train.py:137-145`` load.

CLI:  python -m admm_net_amd.harness time-net  --layers 5 --runs 1000 --out time_net_5.txt [--checkpoint best_model.pth]
      python -m admm_net_amd.harness time-admm --runs 1000 --out time.txt
      python -m admm_net_amd.harness train-step --layers 10 --batch 256 --steps 20     (one JSON line)
"""
from __future__ import annotations

import argparse
import contextlib
import io
import time

import numpy as np
import torch

from . import classical
from .synth import pskdemod, pskmod, steering

NB = ND = 10
F = np.array([-0.25, 0, 0.14])                       # test_time_net.py:16-18
TAU = np.array([0.45, 0.25, 0.63])
C = np.array([-0.5 + 1j, 0.6 - 0.2j, 0.3 + 0.7j])


def demo_scene(rng, sig=None, e=None, data_type=0, snr_e=7.0, snr_w=20.0):
    """One draw of the scene both timing scripts build per run (test_time_net.py:13-92, test_time_admm.py:11-82):
    ``data_type`` 0 = fresh QPSK symbols, 1 = fixed ``sig`` + fresh demodulation noise, 2 = fixed ``sig`` and ``e``
    (the two arrays of data/data.npz).  Returns (y [100, 1], b [100], sigma) in float64 / complex128."""
    S, Dm = steering(F, NB), steering(TAU, ND)
    Psi = np.einsum("l,li,lj->ij", C, S, np.conj(Dm)).reshape(NB * ND, 1)
    if data_type == 0:
        sig = pskmod(rng.integers(0, 4, NB * ND), 4, np.pi / 4)
    if data_type in (0, 1):
        p = np.mean(np.abs(sig) ** 2) / (10 ** (snr_e / 10))
        sig_n = sig + np.sqrt(p / 2) * (rng.standard_normal(len(sig)) + 1j * rng.standard_normal(len(sig)))
        b = pskmod(pskdemod(sig_n, 4, np.pi / 4), 4, np.pi / 4)
        e = sig - b
    else:
        b = sig - e
    real_y = np.diag(b + e) @ Psi
    w = np.sqrt(1 / 2) * (rng.standard_normal((NB * ND, 1)) + 1j * rng.standard_normal((NB * ND, 1)))
    w_var = np.linalg.norm(real_y) ** 2 / (10 ** (snr_w / 10) * NB * ND)
    y = real_y + np.sqrt(w_var) * w
    sigma = np.linalg.norm(e / b) + 1
    return y, b, sigma


def time_net(model, runs=1000, out_path=None, seed=0, sig=None, e=None, data_type=0, verbose=False):
    """test_time_net.py:94-102,131-137: per run a fresh scene, CPU tensors in (as the script passes them), one bracket
    around ``model(y, b, sigma)`` -- which includes the host-to-device copies and the result's return to the CPU, as
    it does for the reference module.  Returns the list of seconds; writes it with ``np.savetxt`` when ``out_path``."""
    rng = np.random.default_rng(seed)
    model.eval()
    times = []
    for i in range(runs):
        y, b, sigma = demo_scene(rng, sig, e, data_type)
        ty = torch.from_numpy(y.flatten().reshape(1, -1)).to(torch.complex64)
        tb = torch.from_numpy(b.reshape(1, -1)).to(torch.complex64)
        ts = torch.from_numpy(np.asarray(sigma).reshape(1, -1)).to(torch.float32)
        start = time.perf_counter()
        phi = model(ty, tb, ts)
        if phi.is_cuda:                         # (GPU tensors in: wait for the result, as .numpy() would)
            torch.cuda.synchronize(phi.device)
        end = time.perf_counter()
        times.append(end - start)
        if verbose:
            print(f"Time: {end - start:.6f} s ")
    if out_path:
        np.savetxt(out_path, times)
    return times


def time_admm(runs=1000, out_path=None, seed=0, sig=None, e=None, data_type=0, verbose=False):
    """test_time_admm.py:85-94,104-110 with the classical solver of admm_net_amd.classical (host, complex128)."""
    rng = np.random.default_rng(seed)
    opts = {"eta_abs": 1e-7, "eta_rel": 1e-7, "max_iter": 100}
    times = []
    for i in range(runs):
        y, b, sigma = demo_scene(rng, sig, e, data_type)
        sink = contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())
        with sink:
            start = time.perf_counter()
            classical.admm_for_us(y, b, ND, NB, 1, sigma, opts)
            end = time.perf_counter()
        times.append(end - start)
    if out_path:
        np.savetxt(out_path, times)
    return times


def save_checkpoint(path, model, optimizer=None, scheduler=None, epoch=0, best_val_loss=float("inf"), config=None,
                    history=None):
    """The dict train.py:306-314 / trainPhi.py:238-246 save on every validation improvement."""
    torch.save({
        "epoch": epoch,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
        "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else {},
        "best_val_loss": best_val_loss,
        "config": config if config is not None else {},
        "history": history if history is not None else {},
    }, path)


def _numpy_scalar_globals():
    """What pickle needs to rebuild a numpy scalar (``np.float64`` ...): train.py:279-282 appends ``np.mean(...)`` results
    to ``history['tau_rmse']`` / ``['f_rmse']`` and saves ``history`` in the checkpoint (train.py:306-314).  These three
    reconstruct plain numbers and run nothing from the file, so the loader stays ``weights_only=True``."""
    try:
        from numpy._core import multiarray as ma          # numpy >= 2
    except ImportError:                                    # numpy 1.x
        from numpy.core import multiarray as ma
    kinds = {type(np.dtype(t)) for t in ("float64", "float32", "float16", "int64", "int32", "bool", "complex64",
                                         "complex128")}
    return [ma.scalar, np.dtype] + sorted(kinds, key=lambda k: k.__name__)


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location="cpu"):
    """train.py:137-145 (resume) / main_for_net.py:100-101 (inference).  Checkpoints written by train.py / trainPhi.py
    are dicts of tensors, plain Python values and numpy scalars (the RMSE history): the no-code loader with numpy
    scalar reconstruction allowed reads them; anything else in the file is refused (never ``weights_only=False``).
    Returns the dict."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict"):
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and ckpt.get("scheduler_state_dict"):
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt


def time_train_step(layers=10, batch=256, steps=20, warmup=3, seed=0, device="cuda:0"):
    """Wall time of one optimisation step at the reference's own training configuration (trainPhi.py:16-38: 10 x 10 grid,
    num_layers = 10, batch_size = 256, AdamW lr 1e-3 / weight decay 1e-3, gradient clipping at 1.0, :179-190): forward in
    train mode (the differentiable route of admm_net_amd.training: HIP eigensolver + HIP contractions), a phi-alignment
    loss against the classical solver's phi labels (generated on the device, csrc/synth.hip), backward, clip, step.
    The loss is a plain normalised squared error: the reference's loss.py is user code outside this path.
    Returns a dict (seconds per step, signals per second)."""
    from . import PhiEstADMMNet, synth
    dev = torch.device(device)
    torch.manual_seed(seed)
    model = PhiEstADMMNet(num_layers=layers, M=NB, N=ND, L=3).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
    y, b, sigma, extra = synth.make_batch_device(batch, NB, ND, seed=20260104 + seed, device=dev, labels=True)
    label = extra["phi"].to(torch.complex64)
    scale = label.abs().pow(2).mean()

    def step():
        opt.zero_grad(set_to_none=True)
        phi = model(y, b, sigma)
        loss = (phi - label).abs().pow(2).mean() / scale
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss

    losses = []
    for _ in range(warmup):
        losses.append(float(step().item()))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    losses.append(float(last.item()))
    return {"config": f"PhiEstADMMNet 10x10 K={layers}, batch {batch}, AdamW + clip 1.0 (trainPhi.py:16-38, 179-190)",
            "seconds_per_step": round(dt, 6), "signals_per_second": round(batch / dt, 1), "steps": steps,
            "loss_first": round(losses[0], 6), "loss_last": round(losses[-1], 6)}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("time-net")
    a.add_argument("--layers", type=int, default=5)
    a.add_argument("--runs", type=int, default=1000)
    a.add_argument("--out", default=None)
    a.add_argument("--checkpoint", default=None)
    a.add_argument("--seed", type=int, default=0)
    b = sub.add_parser("time-admm")
    b.add_argument("--runs", type=int, default=1000)
    b.add_argument("--out", default=None)
    b.add_argument("--seed", type=int, default=0)
    c = sub.add_parser("train-step")
    c.add_argument("--layers", type=int, default=10)
    c.add_argument("--batch", type=int, default=256)
    c.add_argument("--steps", type=int, default=20)
    args = ap.parse_args(argv)
    if args.cmd == "train-step":
        import json
        print(json.dumps(time_train_step(args.layers, args.batch, args.steps)))
        return
    if args.cmd == "time-net":
        from . import PhiEstADMMNet
        torch.manual_seed(0)
        model = PhiEstADMMNet(num_layers=args.layers, M=NB, N=ND, L=3)
        if args.checkpoint:
            load_checkpoint(args.checkpoint, model)
        out = args.out or (f"time_net_{args.layers}.txt" if args.layers != 10 else "time_net.txt")
        t = time_net(model, args.runs, out, args.seed)
    else:
        out = args.out or "time.txt"
        t = time_admm(args.runs, out, args.seed)
    t = np.asarray(t)
    print(f"{out}: {len(t)} runs, mean {t.mean():.6f} s, median {np.median(t):.6f} s, first {t[0]:.6f} s")


if __name__ == "__main__":
    main()
