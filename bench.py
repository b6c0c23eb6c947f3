#!/usr/bin/env python3
"""Headline benchmark: signals/sec of the K-layer ADMM-Net forward on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg5|ref]
    (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks are already there; a plain
     `python bench.py --gpus N` starts them itself as child processes and relays rank 0's line.)

Default workload = cfg3, the configuration BASELINE.json's metric is quoted on (K=16, 16x16 grid -> D=256,
n=257, batch 65536 per GPU, spectrum on 1024 atoms); cfg2 = configs[1]; cfg5 = K=32 PhiEstADMMNet followed by
the classical grid peak search on a 2048-atom coarse grid inside the timed step.

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
PEAK_HBM_GBS = 8000.0                  # the same guide: HBM3E, ~8 TB/s
WORKLOADS = {
    # name: (Nb, Nd, K, batch per GPU, tau atoms, f atoms)   -- BASELINE.json configs[1] / [2] / [4], SURVEY section 0
    "cfg2": (8, 16, 8, 4096, 32, 16),
    "cfg3": (16, 16, 16, 65536, 32, 32),
    "cfg5": (16, 16, 32, 65536, 64, 32),     # PhiEstADMMNet + alt_peak_search post-processing (main_for_net.py:99-126)
    "ref": (10, 10, 10, 4096, 32, 16),        # the geometry every script of the reference uses (main_for_net.py:99, trainPhi.py:23-25)
    "mid": (12, 16, 16, 4096, 32, 32),        # a geometry between the two tuned sizes (D = 192): the padded route of csrc/api.hip
}
# kernel classes of admmnet_profile_read (include/admmnet.h): "trideig" = tridiagonal eigensolver
# (divide & conquer, or QL with ADMMNET_EIG=ql), "backtransform" = V = Q W (MFMA GEMM, or rotation replay)
# "gfunction" = the G-layer evaluated as a matrix function (csrc/spectral_fused.hip: the default route; the four eigen-classes
# then only see the matrices its per-matrix checks send back)
KERNELS = ["prep", "tridiag", "trideig", "backtransform", "rebuild", "zstep", "head", "spectrum", "gfunction"]


def flops_per_signal(K, n, D, natoms):
    """F = (K-1) 24 n^3 + K 256 D + 8 D Natoms."""
    return (K - 1) * 24.0 * n ** 3 + K * 256.0 * D + 8.0 * D * natoms


def measured_counters(workload, kernel, mats_per_launch):
    """(HBM bytes per launch, MFMA-busy fraction) of `kernel` from the committed PMC passes (profiles/traffic.json:
    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ_* in separate runs of this script, FETCH_SIZE doubled as the
    gfx950 guide prescribes).  Entries are keyed by workload and recorded per matrix, so a launch over
    `mats_per_launch` matrices scales them.  (None, None) when there is no measurement for this workload / kernel."""
    try:
        t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        e = t["workloads"][workload]["kernels"][kernel]
        tr = e.get("traffic_bytes_per_matrix")
        return (None if tr is None else tr * mats_per_launch), e.get("mfma_busy")
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


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


def spawn_ranks(n):
    """A plain `python bench.py --gpus N`: start the N ranks as CHILD processes (torch.distributed.run, one per GPU) before
    anything here touches the GPU, relay their output and return their status.  `--standalone` lets the launcher pick its
    own rendezvous port (no bind-close-reuse race between several launches on one host); 127.0.0.1 because the container
    hostname may not resolve."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as CHILD processes (nothing here has touched the GPU
        # yet, and nothing is exec'ed over this process), relay their output and exit with their status
        sys.exit(spawn_ranks(args.gpus))
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
    post_search = args.workload == "cfg5"     # PhiEst inference + classical peak search instead of head + spectrum
    torch.manual_seed(0)
    cls = A.PhiEstADMMNet if post_search else A.ADMMNet
    model = cls(M=Nb, N=Nd, L=3, num_layers=K).eval()            # random init of the reference architecture
    # inputs are generated ON the device (csrc/synth.hip: the generate_data.py:133-221 recipe, one workgroup per sample)
    ty, tb, ts, _ = synth.make_batch_device(B, Nb, Nd, seed=20260104 + rank, device=dev)
    nhost = min(B, 2048)                       # the CPU-baseline leg works on a copy of the first samples
    y, b, s = (t[:nhost].cpu().numpy() for t in (ty, tb, ts))
    taus = torch.linspace(0.0, 1.0, ntau + 1, dtype=torch.float64)[:-1].to(dev)
    fs = torch.linspace(-0.5, 0.5, nf + 1, dtype=torch.float64)[:-1].to(dev)
    # cfg5: alt_peak_search's own coarse grid (np.arange(0, 1 - xstep, xstep) x np.arange(-.5, .5 - xstep, ystep),
    # peakSearchUtils.py:105-106) sized to ntau x nf atoms, two refinement rounds
    ps_opts = {"xstep": 1.0 / (ntau + 1), "ystep": 1.0 / nf, "iter": 2}
    if post_search:
        from admm_net_amd import peak_search as ps
        ax, ay = ps.coarse_axes(ps_opts)
        assert len(ax) * len(ay) == natoms, (len(ax), len(ay), natoms)
    sf = sharded.ShardedForward(model, scope="global")
    lib = _lib.load()

    def step():
        # the same host path at every N (layer-at-a-time C ABI under ShardedForward; at N = 1 its all-reduce is skipped),
        # so the N = 1 point of a scaling run and the single-GPU line are the same measurement by construction
        phi, head = sf(ty, tb, ts)
        if post_search:
            pk, cnt = ops.peak_search(phi, Nd, Nb, ps_opts, max_peaks=256)    # xbase = Nd (delay), ybase = Nb (Doppler)
            tail = pk[:, :16].contiguous()                                     # the peak list is the final output
        else:
            spec = ops.spectrum(phi, Nd, Nb, taus, fs)
            tail = head
        if world > 1:
            tail = sf._gather(tail, dim=0 if post_search else 1)               # final peak output over RCCL / xGMI
        return phi, tail

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
    NK = len(KERNELS)
    ms = (ctypes.c_double * NK)()
    cnt = (ctypes.c_int64 * NK)()
    _lib.check(lib.admmnet_profile_read(ms, cnt, NK), "admmnet_profile_read")
    dropped = int(lib.admmnet_profile_dropped())      # launches the event pool could not record (0 in a healthy run)
    lib.admmnet_profile_enable(0)
    if dropped:
        log(f"WARNING: {dropped} launches were not recorded by the event profiler: per-step kernel sums withheld")
    ranks = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)          # ranks that actually took part in the collective
        ranks = int(one.item())
        assert ranks == dist.get_world_size() == world
    assert torch.isfinite(torch.view_as_real(out[0])).all()

    if rank == 0:
        signals = B * world * args.steps
        value = signals / dt
        F = flops_per_signal(K, n, D, natoms)
        per = {KERNELS[i]: (ms[i], cnt[i]) for i in range(NK)}
        dom = max(("tridiag", "trideig", "backtransform", "rebuild", "gfunction"), key=lambda k_: per[k_][0])
        fast = dom == "gfunction"
        kf = 0.0 if fast else kernel_flops_per_matrix(n)[dom]
        chunk = min(B, 8192)
        launches = max(per[dom][1], 1)
        avg_ms = per[dom][0] / launches
        mats_per_launch = B / math.ceil(B / chunk)          # every launch of a G-layer kernel works on one chunk
        ach = kf * mats_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, mfma_busy = measured_counters(args.workload, dom, mats_per_launch)
        # how the G-layers of the last timed step were evaluated (status words of the C ABI: [1] matrix-layers the per-matrix
        # checks sent to the eigensolver, [2] matrix-layers evaluated as a matrix function)
        st_words = [int(v) for v in getattr(sf, "last_status", [0, 0, 0, 0])]
        glayers = st_words[1] + st_words[2]
        # time-weighted matrix-core occupancy over ALL kernel classes: this run's HIP-event time per class x the
        # MFMA-busy fraction of that class from the committed SQ pass (classes without matrix-core work count as 0)
        tot_ms = sum(v[0] for v in per.values())
        busy = {k_: measured_counters(args.workload, k_, 1)[1] for k_ in per}
        known = [k_ for k_ in ("tridiag", "trideig", "backtransform", "rebuild") if busy[k_] is not None]
        mfma_w = (round(sum(per[k_][0] * busy[k_] for k_ in known) / tot_ms, 4)
                  if tot_ms > 0 and len(known) == 4 else None)
        kms = None if dropped else {k_: round(v[0] / args.steps, 3) for k_, v in per.items()}
        if fast:
            # the dominant kernel streams the lower triangles: ALGORITHMIC bytes per matrix-layer = Z in + G out, 8 n (n + 1) B
            # (DESIGN.md section 4); what it really moves is `traffic` (PMC), several sweeps of Z through the L2 / MALL
            # (with the lazy Z update of the previous layer folded into its first sweep -- the default -- also G in + Z out: 16 n (n + 1) B)
            folded = os.environ.get("ADMMNET_SF_FOLD") != "0" and os.environ.get("ADMMNET_SPECTRAL_FUSED") != "0"
            gbytes = (16.0 if folded else 8.0) * n * (n + 1)
            ach_gbs = gbytes * mats_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roof = {"bound": "hbm", "kernel": "gfunction (sp_fused_kernel)", "achieved": round(ach_gbs, 2), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(ach_gbs / PEAK_HBM_GBS, 5), "traffic": traffic, "mfma_busy": mfma_busy,
                    "avg_launch_ms": round(avg_ms, 4), "matrices_per_launch": mats_per_launch,
                    "algorithmic_bytes_per_matrix": gbytes,
                    "algorithmic_bytes_note": ("Z, G of the previous layer in; new Z, new G out (lower triangles, complex64)" if folded
                                               else "Z in, G out (lower triangles, complex64)"),
                    "eigensolver_equivalent_tflops": round(F * value / world / 1e12, 3),
                    "eigensolver_equivalent_note": (
                        "value x the CANONICAL flop count of SURVEY 8(d) (24 n^3 per matrix-layer through an eigensolver): what an "
                        "eigensolver pipeline would have to sustain for this throughput -- NOT executed work and not a utilisation "
                        "(it exceeds the fp32 matrix-core peak of %.1f TFLOP/s): the G-layer is evaluated as a matrix function "
                        "(two-vector subspace iteration + one Hermitian square in bf16 on the matrix cores, checked per matrix; "
                        "csrc/spectral.hip) and only the matrices its checks reject run the eigensolver.  ADMMNET_SPECTRAL=0 "
                        "measures the eigensolver pipeline itself (profiles/, README)." % PEAK_FP32_MFMA_TFLOPS),
                    "glayer_matrix_function_fraction": (round(st_words[2] / glayers, 5) if glayers else None),
                    "glayer_eigensolver_fallback_fraction": (round(st_words[1] / glayers, 5) if glayers else None),
                    "kernel_ms_per_step": None if dropped else {k_: round(v[0] / args.steps, 3) for k_, v in per.items()},
                    "kernel_ms_sum_over_ms_per_step": (None if dropped else round(
                        sum(v[0] for v in per.values()) / args.steps / (dt / args.steps * 1e3), 4)),
                    "profile_dropped_launches": dropped}
        roof_eig = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 5),
                "traffic": traffic, "mfma_busy": mfma_busy,
                "avg_launch_ms": round(avg_ms, 4), "matrices_per_launch": mats_per_launch,
                "flops_per_matrix": kf,
                "end_to_end_tflops": round(F * value / world / 1e12, 3),
                "end_to_end_frac": round(F * value / world / 1e12 / PEAK_FP32_MFMA_TFLOPS, 5),
                "mfma_busy_weighted": mfma_w,
                "executed_flops_note": (
                    "frac / end_to_end_frac price the CANONICAL flop count of SURVEY 8(d) (24 n^3 per matrix-layer: "
                    "16/3 tridiagonalise + 8/3 tridiagonal eigensolve + 8 back-transform + 8 rebuild), not executed "
                    "instructions: the rebuild computes the lower triangle only (~4 n^3 executed of the canonical 8 n^3, "
                    "so its class fraction can exceed 1), complex products on the matrix cores use the 3-multiplication "
                    "form (3/4 of the real MFMAs), the first layer is an O(n^2) arrowhead solve and the divide & conquer "
                    "deflates most columns.  mfma_busy_weighted is the matrix-core occupancy actually measured "
                    "(time-weighted SQ_VALU_MFMA_BUSY_CYCLES over all kernel classes); end_to_end_frac is NOT occupancy."),
                "kernel_ms_per_step": kms,
                "kernel_ms_sum_over_ms_per_step": (None if dropped else
                                                   round(tot_ms / args.steps / (dt / args.steps * 1e3), 4)),
                "profile_dropped_launches": dropped}
        if not fast:
            roof = roof_eig
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(model, Nb, Nd, K, y, b, s, out[0], args.workload, head=not post_search)
            for key, leg in (("classical", classical_baseline), ("single_signal_latency", latency_leg)):
                try:                                   # secondary legs must never cost the record of the timed run
                    cpu[key] = leg()
                except Exception as exc:               # noqa: BLE001
                    cpu[key] = {"error": f"{type(exc).__name__}: {exc}"}
        line = {"metric": "signals/sec (K-layer ADMM-Net forward)", "value": round(value, 2), "unit": "signals/s",
                "n_gpus": ranks if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic (generate_data.py recipe, generated on the device)",
                "config": {"workload": (f"{args.workload}: PhiEstADMMNet K={K} grid {Nb}x{Nd} (D={D}, n={n}) batch {B}/GPU "
                                        f"+ alt_peak_search on a {natoms}-atom coarse grid" if post_search else
                                        f"{args.workload}: ADMMNet K={K} grid {Nb}x{Nd} (D={D}, n={n}) "
                                        f"batch {B}/GPU + spectrum on {natoms} atoms"),
                           "batch_per_gpu": B, "global_batch": B * world, "K": K, "D": D, "n": n,
                           "natoms": natoms, "batch_mean_scope": "global", "chunk": chunk,
                           "weights": "torch.manual_seed(0) default init",
                           "glayer": ("eigensolver pipeline (ADMMNET_SPECTRAL=0)" if os.environ.get("ADMMNET_SPECTRAL") == "0"
                                      else "matrix function, checked per matrix; eigensolver on the rejected matrices")},
                "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def classical_baseline():
    """cfg1 (BASELINE.json configs[0]): the classical solver admm_for_us on the data/data.npz scene of main.py:51-95
    (fixture tests/golden/cfg1_data.npz holds the two arrays of that file), CPU, complex128, single signal -- wall
    time beside the published mean of results/time/time.txt."""
    import contextlib
    import io
    from admm_net_amd import classical
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_data.npz"), allow_pickle=False)
    opts = {"eta_abs": 1e-7, "eta_rel": 1e-7, "max_iter": 100}                  # main.py:88-93
    ts, it = [], 0
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                # no BLAS thread control: time it as it comes and say so
        threadpool_limits = None
    one_thread = threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()
    # one BLAS thread: at n = 101 the threaded SVD of scipy only spins (26x slower on 8 threads in the build container)
    with contextlib.redirect_stdout(io.StringIO()), one_thread:   # admm_for_us prints, as the reference does
        for r in range(21):                                                     # first run = warm-up, as test_time_admm.py
            y, b, sigma = classical.cfg1_scene(z["sig"], z["e"], seed=r)
            t0 = time.perf_counter()
            phi, it = classical.admm_for_us(y, b, 10, 10, 1, sigma, opts)
            if r:
                ts.append(time.perf_counter() - t0)
    return {"seconds_per_signal": round(float(np.mean(ts)), 6), "iterations": int(it), "runs": len(ts),
            "published_reference_seconds": 0.5244,
            "blas_threads": 1 if threadpool_limits else "uncontrolled (threadpoolctl absent)",
            "note": "admm_net_amd.classical.admm_for_us on the main.py data.npz scene (D=100), 1 core; "
                    "published = mean of /root/reference/results/time/time.txt (hardware unstated)"}


def latency_leg(runs=200):
    """Single-signal latency of the MI355X path, measured the way the reference measures the only numbers it
    publishes (test/test_time_net.py:94-102,131-137 -> results/time/time_net.txt, time_net_5.txt): 10 x 10 demo scene,
    fresh noise per run, CPU tensors in, one perf_counter bracket around model(y, b, sigma) including the host<->device
    copies.  First call (warm-up in the reference's files too) reported apart."""
    import admm_net_amd as A
    from admm_net_amd import harness
    out = {"how": f"harness.time_net, {runs} runs after the first call, PhiEstADMMNet 10x10, batch 1, CPU tensors in/out",
           "published_reference_seconds": {"K5_mean": 0.0965, "K5_median": 0.0904, "K10_mean": 0.1910,
                                           "K10_median": 0.1827, "source": "results/time/time_net_5.txt, time_net.txt "
                                           "(K=10 inferred, hardware unstated)"}}
    for K in (5, 10):
        torch.manual_seed(0)
        m = A.PhiEstADMMNet(num_layers=K, M=10, N=10, L=3)
        t = np.asarray(harness.time_net(m, runs + 1, None, seed=K))
        out[f"K{K}"] = {"mean_s": round(float(t[1:].mean()), 6), "median_s": round(float(np.median(t[1:])), 6),
                        "p99_s": round(float(np.quantile(t[1:], 0.99)), 6), "first_call_s": round(float(t[0]), 6)}
    return out


def cpu_model():
    """CPU model string of the box (BASELINE.md section 3 asks for model + core count next to the baseline)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(model, Nb, Nd, K, y, b, s, phi_gpu, workload, head=True):
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
        out = R.forward(sd, ty, tb, ts, Nb, Nd, K, dtype="f32", head=head)
        return time.perf_counter() - t0, out
    t_probe, _ = run(16)
    log(f"cpu probe: 16 signals in {t_probe:.2f} s")
    nsig = int(max(16, min(len(y), 12.0 / max(t_probe / 16, 1e-6))))
    nsig = min(nsig, 2048)
    t, out = run(nsig)
    log(f"cpu sample: {nsig} signals in {t:.2f} s")
    return {"value": round(nsig / t, 2), "unit": "signals/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "logical_cpus_on_box": os.cpu_count(),
            "sample": f"{nsig} signals of {workload} (oracle fp32, torch {torch.__version__}, {cores} threads, "
                      f"{t:.1f} s); batch mean over the sample, so not comparable signal-by-signal",
            "seconds": round(t, 2)}


if __name__ == "__main__":
    main()
