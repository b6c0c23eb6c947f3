"""Developer diagnostic: `python tests/gpu_sweep_env.py VAR v1,v2,... workload[,workload] [class]` -- bench.py per value of one
environment switch (each in a fresh child: the switches are read once per process), one line per run."""
import json
import os
import subprocess
import sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
var, vals, wls = sys.argv[1], sys.argv[2].split(","), sys.argv[3].split(",")
cls = sys.argv[4] if len(sys.argv) > 4 else "trideig"
for w in wls:
    for v in vals:
        env = dict(os.environ)
        if v != "-":
            env[var] = v
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "10", "--warmup", "2",
                            "--no-cpu-baseline"] + (["--batch", "4096"] if w in ("cfg3", "cfg5") else []),
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
            print(f"{w:5s} {var}={v:4s} {d['value']:10.1f} signals/s  {cls} {d['roofline']['kernel_ms_per_step'][cls]:8.3f} ms/step", flush=True)
        except Exception as e:   # noqa: BLE001
            print(w, var, v, "FAILED", r.returncode, r.stderr[-300:], flush=True)
