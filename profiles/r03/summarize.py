"""Turn the rocprofv3 result databases (rocpd SQLite, the default output of ROCm 7.2's rocprofv3) of profiles/r03/collect.sh
into the small CSV summaries kept in this directory:  python profiles/r03/summarize.py <prof dir> <out prefix>

  <prefix>_kernel_stats.csv    per kernel: calls, total / average / min / max duration (ns), share of GPU kernel time
  <prefix>_pmc_by_kernel.csv   per kernel and counter: mean value per dispatch (FETCH_SIZE / WRITE_SIZE in KB as rocprofv3
                               reports them; the gfx950 guide's x2 correction for FETCH_SIZE is applied in traffic.json, not here)
"""
import csv
import os
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict


_DEMANGLED = {}


def demangle(name):
    """rocpd stores the mangled kernel symbol (`_ZN7admmnet...kd`): c++filt it (binutils, in the ROCm image)."""
    if name not in _DEMANGLED:
        raw = name[:-3] if name.endswith(".kd") else name
        out = raw
        if raw.startswith("_Z"):
            try:
                out = subprocess.run(["c++filt", raw], capture_output=True, text=True, check=True).stdout.strip() or raw
            except (OSError, subprocess.CalledProcessError):
                out = raw
        _DEMANGLED[name] = out
    return _DEMANGLED[name]


def short(name):
    name = demangle(name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("admmnet::", "").replace("void ", "")
    return name.strip()


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id").fetchall()
    agg = defaultdict(list)
    for name, st, en in rows:
        agg[short(name)].append(en - st)
    tot = sum(sum(v) for v in agg.values())
    out = [(k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot) for k, v in agg.items()]
    return sorted(out, key=lambda r: -r[2])


def pmc_by_kernel(db):
    c = sqlite3.connect(db)
    # one row per (dispatch, counter, hardware instance): sum the instances of a dispatch, then average over dispatches
    q = ("select s.kernel_name, p.name, d.event_id, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on "
         "e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on "
         "d.kernel_id = s.id group by s.kernel_name, p.name, d.event_id")
    agg = defaultdict(lambda: defaultdict(list))
    for name, ctr, _ev, val in c.execute(q):
        agg[short(name)][ctr].append(val)
    return {k: {ctr: (sum(v) / len(v), len(v)) for ctr, v in d.items()} for k, d in agg.items()}


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    kt = os.path.join(src, "kt")
    db = [os.path.join(kt, f) for f in os.listdir(kt) if f.endswith(".db")][0]
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in kernel_stats(db):
            w.writerow([r[0], r[1], r[2], round(r[3], 1), r[4], r[5], round(r[6], 3)])
    merged = defaultdict(dict)
    for sub in sorted(os.listdir(src)):
        if not sub.startswith("pmc_") or not os.path.isdir(os.path.join(src, sub)):
            continue
        d = os.path.join(src, sub)
        for f in os.listdir(d):
            if f.endswith(".db"):
                for k, ctrs in pmc_by_kernel(os.path.join(d, f)).items():
                    merged[k].update(ctrs)
    ctr_names = sorted({c for d in merged.values() for c in d})
    with open(prefix + "_pmc_by_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Dispatches"] + ctr_names)
        for k in sorted(merged):
            n = max(v[1] for v in merged[k].values())
            w.writerow([k, n] + [round(merged[k][c][0], 2) if c in merged[k] else "" for c in ctr_names])


if __name__ == "__main__":
    main()
