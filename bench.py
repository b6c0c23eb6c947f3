#!/usr/bin/env python3
"""Headline benchmark: signals/sec of the K-layer ADMM-Net forward on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|ref]
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

One "step" = the whole hot path on one resident synthetic batch: ADMMNet forward (K unrolled layers
+ learned peak head) followed by the batch x steering-dictionary spectrum on Natoms (tau, f) atoms.
Weak scaling: every rank processes its own batch; the only data-path exchanges are one (sum, count)
float64 all-reduce per layer (the batch mean of the Z layer, admm_net.py:459) and the final gather of
the peak-head outputs.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel: algorithmic flops per launch / average launch time (HIP events
                  recorded on the launch stream inside the timed region) vs the fp32 MFMA peak
  cpu_baseline -- the oracle (CPU restatement, parity-pinned to the reference) on a bounded sample
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
WORKLOADS = {
    # name: (Nb, Nd, K, batch per GPU, tau atoms, f atoms)   -- BASELINE.json configs[1] / [2], SURVEY section 0
    "cfg2": (8, 16, 8, 4096, 32, 16),
    "cfg3": (16, 16, 16, 65536, 32, 32),
    "ref": (10, 10, 10, 4096, 32, 16),
}
# kernel classes of admmnet_profile_read (include/admmnet.h): "trideig" = tridiagonal eigensolver
# (divide & conquer, or QL with ADMMNET_EIG=ql), "backtransform" = V = Q W (MFMA GEMM, or rotation replay)
KERNELS = ["prep", "tridiag", "trideig", "backtransform", "rebuild", "zstep", "head", "spectrum"]


def flops_per_signal(K, n, D, natoms):
    """F = (K-1) 24 n^3 + K 256 D + 8 D Natoms."""
    return (K - 1) * 24.0 * n ** 3 + K * 256.0 * D + 8.0 * D * natoms


def measured_traffic(workload, kernel, B):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this script, FETCH_SIZE doubled as the gfx950 guide
    prescribes).  None when there is no measurement for this workload / kernel / batch."""
    try:
        t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        if t.get("workload") != workload or B != t.get("batch", 4096):
            return None
        return t["kernels"][kernel]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def kernel_flops_per_matrix(n):
    """Split of the canonical 24 n^3 per matrix over our kernels ('Kernels')."""
    return {"tridiag": 16.0 / 3.0 * n ** 3, "trideig": 8.0 / 3.0 * n ** 3, "backtransform": 8.0 * n ** 3,
            "rebuild": 8.0 * n ** 3}


T0 = time.time()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.time() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the cores this process may actually use (cgroup / affinity)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:   # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("ADMMNET_CPU_THREADS", "64"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    local = local % torch.cuda.device_count()     # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ADMMNET_DIST_BACKEND", "nccl")   # nccl = RCCL on ROCm; gloo for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import admm_net_amd as A
    from admm_net_amd import _lib, ops, sharded, synth

    Nb, Nd, K, B, ntau, nf = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    D, n, natoms = Nb * Nd, Nb * Nd + 1, ntau * nf
    torch.manual_seed(0)
    model = A.ADMMNet(M=Nb, N=Nd, L=3, num_layers=K).eval()      # random init of the reference architecture
    y, b, s, _ = synth.make_batch(B, Nb, Nd, seed=20260104 + rank)
    ty, tb, ts = (torch.from_numpy(v).to(dev) for v in (y, b, s))
    taus = torch.linspace(0.0, 1.0, ntau + 1, dtype=torch.float64)[:-1].to(dev)
    fs = torch.linspace(-0.5, 0.5, nf + 1, dtype=torch.float64)[:-1].to(dev)
    sf = sharded.ShardedForward(model, scope="global")
    lib = _lib.load()

    def step():
        if world > 1:
            phi, head = sf(ty, tb, ts)
            spec = ops.spectrum(phi, Nd, Nb, taus, fs)      # xbase = Nd (delay), ybase = Nb (Doppler)
            gathered = sf._gather(head, dim=1)                      # final peak output over RCCL / xGMI
            return phi, gathered, spec
        tau, f, conf, phi = model(ty, tb, ts)
        spec = ops.spectrum(phi, Nd, Nb, taus, fs)      # xbase = Nd (delay), ybase = Nb (Doppler)
        return phi, (tau, f, conf), spec

    log(f"workload {args.workload}: grid {Nb}x{Nd} K={K} B={B}/GPU world={world}; warmup x{args.warmup}")
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)
        log("warmup step done")

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    lib.admmnet_profile_enable(0 if os.environ.get("ADMMNET_BENCH_NOPROF") else 1)   # (developer: cost of the event pairs)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps: {dt:.3f} s")
    ms = (ctypes.c_double * 8)()
    cnt = (ctypes.c_int64 * 8)()
    _lib.check(lib.admmnet_profile_read(ms, cnt, 8), "admmnet_profile_read")
    lib.admmnet_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(torch.view_as_real(out[0])).all()

    if rank == 0:
        signals = B * world * args.steps
        value = signals / dt
        F = flops_per_signal(K, n, D, natoms)
        per = {KERNELS[i]: (ms[i], cnt[i]) for i in range(8)}
        dom = max(("tridiag", "trideig", "backtransform", "rebuild"), key=lambda k_: per[k_][0])
        kf = kernel_flops_per_matrix(n)[dom]
        chunk = min(B, 8192)
        launches = max(per[dom][1], 1)
        avg_ms = per[dom][0] / launches
        mats_per_launch = B / math.ceil(B / chunk)          # every launch of an eigen-kernel works on one chunk
        ach = kf * mats_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 5),
                "traffic": measured_traffic(args.workload, dom, B),
                "avg_launch_ms": round(avg_ms, 4), "matrices_per_launch": mats_per_launch,
                "flops_per_matrix": kf,
                "end_to_end_tflops": round(F * value / world / 1e12, 3),
                "end_to_end_frac": round(F * value / world / 1e12 / PEAK_FP32_MFMA_TFLOPS, 5),
                "kernel_ms_per_step": {k_: round(v[0] / args.steps, 3) for k_, v in per.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(model, Nb, Nd, K, y, b, s, out[0], args.workload)
        line = {"metric": "signals/sec (K-layer ADMM-Net forward)", "value": round(value, 2), "unit": "signals/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.workload}: ADMMNet K={K} grid {Nb}x{Nd} (D={D}, n={n}) "
                                       f"batch {B}/GPU + spectrum on {natoms} atoms",
                           "batch_per_gpu": B, "global_batch": B * world, "K": K, "D": D, "n": n,
                           "natoms": natoms, "batch_mean_scope": "global", "chunk": chunk,
                           "weights": "torch.manual_seed(0) default init"},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(model, Nb, Nd, K, y, b, s, phi_gpu, workload):
    """Oracle (torch CPU restatement of admm_net.py, kind = 'port') on a bounded sample of the same
    workload; also re-checks parity of the benchmarked batch on that sample."""
    from oracle import admm_net_ref as R
    cores = host_threads()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads (os.cpu_count() = {os.cpu_count()})")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def run(nsig):
        ty, tb, ts = torch.from_numpy(y[:nsig]), torch.from_numpy(b[:nsig]), torch.from_numpy(s[:nsig])
        t0 = time.perf_counter()
        out = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32", head=True)
        return time.perf_counter() - t0, out
    t_probe, _ = run(16)
    log(f"cpu probe: 16 signals in {t_probe:.2f} s")
    nsig = int(max(16, min(len(y), 12.0 / max(t_probe / 16, 1e-6))))
    nsig = min(nsig, 2048)
    t, out = run(nsig)
    log(f"cpu sample: {nsig} signals in {t:.2f} s")
    return {"value": round(nsig / t, 2), "unit": "signals/s", "cores": cores, "kind": "port",
            "sample": f"{nsig} signals of {workload} (oracle fp32, torch {torch.__version__}, {cores} threads, "
                      f"{t:.1f} s); batch mean over the sample, so not comparable signal-by-signal",
            "seconds": round(t, 2)}


if __name__ == "__main__":
    main()
