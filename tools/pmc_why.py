#!/usr/bin/env python
"""Memory-side and wave-state counters of the hot kernels: what the time of the streaming kernels goes to.

    for S in A B C D; do rocprofv3 --pmc <set S> --output-format csv -d gpurun_out/pmcY_${S}${TAG} -- python3 tools/pmc_kernels.py [--align 16]; done
    python3 tools/pmc_why.py [--tag _a16] -> gpurun_out/<round>_pmc_why[_align16].json (copy into profiles/)

Counter sets (TCC has 4 slots per pass, SQ 8; MI355X_MICROARCH.md, rocprofv3 PMC slots):
  A  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
  B  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
  C  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
  D  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
Per case the mean over the launches of the case's kernel (cases are separated by k_fill_random launches, as in pmc_reduce.py)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_reduce  # noqa: E402

SETS = {
    "A": "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
    "B": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum",
    "C": "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum",
    "D": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE",
}


def rows_of(directory):
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sets":
        for k, v in SETS.items():
            print(k, v)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--round", default="r03")
    args = ap.parse_args()
    go = os.path.join(ROOT, "gpurun_out")
    cases = json.load(open(os.path.join(go, "pmcK_times_L%d%s.json" % (args.level, args.tag))))
    res = {c["case"]: {"ms": c["ms"], "compulsory_bytes": c["compulsory_bytes"]} for c in cases}
    for sname in SETS:
        per_counter = rows_of(os.path.join(go, "pmcY_%s%s" % (sname, args.tag)))
        for cname, rows in per_counter.items():
            rows.sort()
            segs = pmc_reduce.segments([(d, n, v) for d, n, v in rows])[-len(cases):]
            for seg, c in zip(segs, cases):
                vals = [v for name, v in seg if c["kernel"] in name]
                if vals:
                    res[c["case"]][cname] = sum(vals) / len(vals)
    for case, r in res.items():
        w, w64 = r.get("TCC_EA0_WRREQ_sum"), r.get("TCC_EA0_WRREQ_64B_sum")
        if w:
            r["write_requests_narrower_than_64B_share"] = 1.0 - (w64 or 0.0) / w
        h, m = r.get("TCC_HIT_sum"), r.get("TCC_MISS_sum")
        if h is not None and m:
            r["l2_hit_rate"] = h / (h + m)
        wc, wa = r.get("SQ_WAVE_CYCLES"), r.get("SQ_WAIT_ANY")
        if wc and wa is not None:
            r["wave_cycles_waiting_share"] = wa / wc
    out = {"note": __doc__.split("\n\n")[0], "sets": SETS, "level": args.level, "align": 16 if "a16" in args.tag else 0, "kernels": res}
    path = os.path.join(go, "%s_pmc_why%s.json" % (args.round, "_align16" if "a16" in args.tag else ""))
    json.dump(out, open(path, "w"), indent=1)
    for case, r in res.items():
        print(case, {k: (round(v, 4) if isinstance(v, float) and v < 10 else v) for k, v in r.items() if "share" in k or "rate" in k or k == "ms"})


if __name__ == "__main__":
    main()
