#!/usr/bin/env python
"""Turn two rocprofv3 counter passes of bench.py into profiles/traffic_latest.json.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py --steps 10 --warmup 2 --no-vcycle --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_WRITE_SIZE -- python3 bench.py --steps 10 --warmup 2 --no-vcycle --no-cpu-baseline
    python tools/pmc_to_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE

Counters are in KiB per dispatch; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section; factor confirmed for
this access shape with tools/bw_probe.hip `cal`), WRITE_SIZE is taken as it is. Only full-size launches
(grid of the 512^3 sweep) enter the mean."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 511 ** 3
KERNELS = {
    "single_step": ("k_stencil7_zmarch<2", 24.0 * N),
    "two_step": ("k_two_stage7", 48.0 * N),
}


def per_kernel(directory, counter):
    rows = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                rows.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]) * 1024.0)
    return rows


def pick(rows, prefix):
    """mean over the dispatches of the kernel whose name contains `prefix` (largest launches only)."""
    best_name, vals = None, []
    for name, v in rows.items():
        if prefix in name and len(v) > len(vals):
            best_name, vals = name, v
    if not vals:
        return None, None
    top = max(vals)
    big = [x for x in vals if x > 0.5 * top]
    return best_name, sum(big) / len(big)


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 10 --warmup 2 "
                   "--no-vcycle --no-cpu-baseline` on MI355X, per-dispatch means (tools/pmc_to_traffic.py); counters are in KiB; "
                   "FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950 -- "
                   "factor confirmed on this access shape by tools/bw_probe.hip cal (aligned 16-B loads: raw/true = 0.500, "
                   "8-byte-aligned 16-B loads: 0.516; WRITE_SIZE raw/true = 1.000 aligned, 1.052 for 8-byte-aligned "
                   "non-temporal stores). The counters sit on the L2's fabric side: Infinity-Cache hits are included."}
    for key, (prefix, alg) in KERNELS.items():
        kn, fb = pick(fetch, prefix)
        _, wb = pick(write, prefix)
        if fb is None or wb is None:
            continue
        out[key] = {"kernel": kn.split("(")[0][-120:], "fetch_bytes": 2.0 * fb, "write_bytes": wb,
                    "bytes_per_launch": 2.0 * fb + wb, "algorithmic_bytes": alg}
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
