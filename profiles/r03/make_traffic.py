"""Rebuild one workload's section of profiles/traffic.json from a `<prefix>_pmc_by_kernel.csv` of summarize.py:
    python profiles/r03/make_traffic.py profiles/r03/cfg3_b4096_pmc_by_kernel.csv 4096 cfg3
    python profiles/r03/make_traffic.py profiles/r03/cfg2_b4096_pmc_by_kernel.csv 4096 cfg2      (also: ref)

Per kernel CLASS of bench.py's roofline block (a class may be several kernels: the three stages of the panel
tridiagonalisation, the back-transform's slab kernel + last-column kernel), summed over the class's kernels:
  traffic_bytes_per_matrix = (2 x FETCH_SIZE + WRITE_SIZE) KB per dispatch x 1024 / matrices per dispatch
                             (FETCH_SIZE doubled: gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md, HBM section)
  mfma_busy                = sum SQ_VALU_MFMA_BUSY_CYCLES / (32 x sum SQ_BUSY_CYCLES)
                             (SQ_BUSY_CYCLES sums the 32 shader engines, MFMA-busy the 1024 SIMDs)
  wave_cycles_*            = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
"""
import csv
import json
import os
import sys

CLASSES = {   # kernel classes of bench.py -> kernel names (D = 256 route; D <= 128 route: the back-transform is fused into
    # the rebuild kernel, so that class is empty there)
    "tridiag": ["tridiag_panel_kernel", "tridiag_reg_kernel"],
    "trideig": ["dc_kernel"],
    "backtransform": ["wy_apply_kernel", "wy_lastcol_kernel"],
    "rebuild": ["rebuild_big_kernel", "back_rebuild_kernel"],
    "prep": ["prep_kernel"],
    "gfunction": ["sp_fused_kernel"],   # the G-layer as a matrix function (default route): profile it with `only=gfunction,prep`
}


def main():
    src, mats = sys.argv[1], float(sys.argv[2])
    wl = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
    rows = list(csv.DictReader(open(src)))

    def val(r, k):
        return float(r[k]) if r.get(k) not in (None, "") else 0.0

    only = None   # optional 4th argument "only=cls1,cls2": merge just these classes into the workload's existing entry
    for a in sys.argv[4:]:
        if a.startswith("only="):
            only = a[5:].split(",")
    out = {}
    for cls, names in CLASSES.items():
        if only and cls not in only:
            continue
        rs = [r for r in rows if any(r["Kernel"].startswith(nm) for nm in names)]
        if not rs:
            continue
        fetch = sum(val(r, "FETCH_SIZE") for r in rs)
        write = sum(val(r, "WRITE_SIZE") for r in rs)
        busy = sum(val(r, "SQ_BUSY_CYCLES") for r in rs)
        wc = sum(val(r, "SQ_WAVE_CYCLES") for r in rs)
        e = {
            "kernels": sorted(r["Kernel"] for r in rs),
            "traffic_bytes_per_matrix": round((2.0 * fetch + write) * 1024.0 / mats, 1),
            "fetch_kb_per_dispatch": round(fetch, 2),
            "write_kb_per_dispatch": round(write, 2),
            "mfma_busy": round(sum(val(r, "SQ_VALU_MFMA_BUSY_CYCLES") for r in rs) / (32.0 * busy), 4) if busy else None,
            "valu_insts_per_matrix": round(sum(val(r, "SQ_INSTS_VALU") for r in rs) / mats, 1),
        }
        if wc:
            e["wave_cycles_parked"] = round(sum(val(r, "SQ_WAIT_ANY") for r in rs) / wc, 3)
            e["wave_cycles_issue_stalled"] = round(sum(val(r, "SQ_WAIT_INST_ANY") for r in rs) / wc, 3)
            e["wave_cycles_issuing"] = round(sum(val(r, "SQ_ACTIVE_INST_ANY") for r in rs) / wc, 3)
        out[cls] = e
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "traffic.json")
    t = json.load(open(path))
    if only and wl in t["workloads"]:
        for cls, e in out.items():
            e["source"] = os.path.basename(src)
            t["workloads"][wl]["kernels"][cls + ("_matrix_function_route" if cls == "prep" else "")] = e
    else:
        t["workloads"][wl] = {"batch_profiled": int(mats), "round": 3, "source": os.path.basename(src), "kernels": out}
    json.dump(t, open(path, "w"), indent=1)
    tot = sum(e["traffic_bytes_per_matrix"] for e in out.values())
    print("total HBM bytes per matrix and layer: %.2f MB" % (tot / 1e6))
    for k, e in out.items():
        print("  %-14s %8.2f MB  mfma_busy %s" % (k, e["traffic_bytes_per_matrix"] / 1e6, e["mfma_busy"]))


if __name__ == "__main__":
    main()
