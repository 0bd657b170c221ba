#!/usr/bin/env python
"""Reduce the two rocprofv3 counter passes of tools/pmc_kernels.py to per-kernel fabric traffic:

    python3 tools/pmc_reduce.py [--level 9] [--out profiles/r02_pmc_kernels.json]

Reads gpurun_out/pmcK_FETCH, gpurun_out/pmcK_WRITE (counter_collection.csv; counters in KiB per dispatch) and the timing
pass gpurun_out/pmcK_times_L<level>.json.  Cases are separated in the dispatch stream by the k_fill_random launch that
follows each of them; within a case's segment the dispatches whose kernel name contains the case's pattern are averaged.
FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section;
confirmed for these access shapes by `tools/lab/bw_probe.hip cal`, profiles/README.md); WRITE_SIZE is taken as it is.  The
counters sit at the L2's fabric side: Infinity-Cache hits are included, so `traffic` bounds HBM bytes from above."""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8000.0e9

# the sources a case's kernel is compiled from: their digest travels with the counters, and bench.py reports `traffic` only for a
# kernel whose sources are still the profiled ones (kernel_source_digest)
COMMON_SOURCES = ["examg_common.h"]
CASE_SOURCES = {"jacobi_2step": ["kernels_twostage.hip"], "jacobi_3step": ["kernels_twostage.hip"], "rbgs_3colours": ["kernels_twostage.hip"], "rbgs_fused_sweep": ["kernels_twostage.hip"], "rbgs_fused_sweep_prolong": ["kernels_twostage.hip"],
                "rbgs_fused_sweep_zero": ["kernels_twostage.hip"], "residual_restrict": ["kernels_transfer.hip"], "restrict": ["kernels_transfer.hip"],
                "prolong_add": ["kernels_transfer.hip"], "dot_norm": ["kernels_blas.hip"], "jacobi_27entry_two_steps": ["kernels_sf27pair.hip"],
                "jacobi_27entry_step_residual": ["kernels_sf27pair.hip"]}


def kernel_source_digest(case):
    """sha256 (16 hex digits) over the HIP sources the kernel of `case` is compiled from."""
    import hashlib

    h = hashlib.sha256()
    for name in COMMON_SOURCES + CASE_SOURCES.get(case, ["kernels_stencil.hip"]):
        with open(os.path.join(ROOT, "exastencils_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def dispatches(directory, counter):
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows


def segments(rows):
    segs, cur = [], []
    for _, name, val in rows:
        if "k_fill_random" in name:
            if cur:
                segs.append(cur)
            cur = []
        else:
            cur.append((name, val))
    if cur:
        segs.append(cur)
    return segs


def per_case(rows, cases):
    segs = segments(rows)[-len(cases):]
    out = []
    for seg, case in zip(segs, cases):
        vals = [v for name, v in seg if case["kernel"] in name]
        names = sorted({name.split("(")[0] for name, v in seg if case["kernel"] in name})
        out.append((sum(vals) / len(vals) if vals else None, names[0][-140:] if names else None, len(vals)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_pmc_kernels.json"))
    args = ap.parse_args()
    go = os.path.join(ROOT, "gpurun_out")
    cases = json.load(open(os.path.join(go, "pmcK_times_L%d%s.json" % (args.level, args.tag))))
    fetch = per_case(dispatches(os.path.join(go, "pmcK_FETCH" + args.tag), "FETCH_SIZE"), cases)
    write = per_case(dispatches(os.path.join(go, "pmcK_WRITE" + args.tag), "WRITE_SIZE"), cases)
    import subprocess

    try:      # the tree the profiled library was built from (gpurun ships no .git: build_commit.txt)
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:      # noqa: BLE001
        stamp = os.path.join(ROOT, "build_commit.txt")      # written by tools/grun.sh before the snapshot leaves for the GPU box
        commit = open(stamp).read().strip() if os.path.exists(stamp) else None
    out = {"note": __doc__.split("\n\n")[-1].replace("\n", " "), "commit": commit, "level": args.level, "cells": "%d^3" % (1 << args.level),
           "align": cases[0].get("align", 0) if cases else 0, "kernels": []}
    for c, (fb, kn, nf), (wb, _, nw) in zip(cases, fetch, write):
        rec = dict(c)
        rec["kernel_name"] = kn
        rec["source_digest"] = kernel_source_digest(c["case"])
        if fb is not None and wb is not None:
            rec.update(fetch_bytes=2.0 * fb, write_bytes=wb, traffic=2.0 * fb + wb, traffic_over_compulsory=(2.0 * fb + wb) / c["compulsory_bytes"],
                       dispatches_averaged=[nf, nw])
        rec["frac"] = c["compulsory_bytes"] / (c["ms"] * 1e-3) / HBM_PEAK
        out["kernels"].append(rec)
        print("%-22s %7.4f ms  compulsory %6.3f GB  traffic %s GB (%s x)  frac %.3f" % (
            c["case"], c["ms"], c["compulsory_bytes"] / 1e9, "%6.3f" % (rec["traffic"] / 1e9) if "traffic" in rec else "   n/a",
            "%.2f" % rec["traffic_over_compulsory"] if "traffic" in rec else "n/a", rec["frac"]))
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
