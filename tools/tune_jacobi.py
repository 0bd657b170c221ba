#!/usr/bin/env python
"""Sweep the tuning knobs of the 7-point z-march kernel on the GPU box (needs the -DEXAMG_TUNE build):

    python tools/tune_jacobi.py [--level 9] [--quick]

Builds exastencils_amd/libexamg_tune.so if missing, times every (ry, wy, nt, my, remap, blocks) combination
with events on the launch stream, checks every variant bit for bit against the first, prints a table sorted by
time and writes gpurun_out/tune_jacobi.json."""
import argparse
import ctypes as C
import glob
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_tune():
    import __graft_entry__ as ge

    out = os.path.join(ROOT, "exastencils_amd", "libexamg_tune.so")
    srcs = sorted(glob.glob(os.path.join(ge.CSRC, "*.hip")))
    if ge._newer(out, srcs + glob.glob(os.path.join(ge.CSRC, "*.h"))):
        subprocess.check_call(["hipcc"] + ge.HIPCC_FLAGS + ["-DEXAMG_TUNE", "-o", out] + srcs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch

    from exastencils_amd import lib

    lib.LIB_PATH = build_tune()
    from exastencils_amd.field import laplace_fd
    from exastencils_amd.layout import FieldLayout
    from exastencils_amd.ops import HipOps

    ops = HipOps(0)
    L = ops.L
    L.examg_debug_tune.argtypes = [C.c_char_p, C.c_int]
    n = 1 << args.level
    lu, lf = FieldLayout.node(3, (n, n, n), 1), FieldLayout.node(3, (n, n, n), 0)
    u, un, f = ops.new_array(lu.size), ops.new_array(lu.size), ops.new_array(lf.size)
    ops.fill_random(u, 12345)
    ops.fill_random(f, 777)
    A = laplace_fd(3, (1.0 / n,) * 3)
    w = 0.8 / A.diag
    b, e = [1, 1, 1], [n, n, n]
    lus, lfs = lu.c_struct(), lf.c_struct()
    updates = (n - 1) ** 3

    state = [u, un]

    def run():
        # ping-pong like the smoother's slots, so that what one sweep wrote is what the next one reads
        ops.stencil_op(2, lus, state[0], lfs, f, lus, state[1], A, w, -1, b, e)
        state.reverse()

    def run_check():
        ops.stencil_op(2, lus, u, lfs, f, lus, un, A, w, -1, b, e)

    # bandwidth reference points: same three arrays, flat 16-byte-per-lane streams
    L.examg_debug_triad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
    nflat = min(u.numel(), f.numel()) // 2 * 2
    for mode, name, nbytes in ((0, "triad 2R+1W", 24), (1, "copy 1R+1W", 16), (2, "read 2R", 16)):
        for nt in ((0, 1) if mode < 2 else (0,)):
            for blocks in (2048, 4096, 8192, 16384):
                def tr():
                    L.examg_debug_triad(un.data_ptr(), u.data_ptr(), f.data_ptr(), nflat, 0.5, nt, mode, blocks, None)
                tr()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    tr()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                print("REF %-12s nt=%d blocks=%5d  %.4f ms  %.0f GB/s" % (name, nt, blocks, ms, nbytes * nflat / ms / 1e6), flush=True)

    if args.quick:
        grid = dict(ry=[1, 2], wy=[4], nt=[0, 1], my=[0], pf=[0, 1], remap=[0, 1], blocks=[1, 1024, 2048], dir=[0, 1, -1])
    else:
        grid = dict(ry=[1, 2, 4], wy=[1, 2, 4, 8], nt=[0, 1], my=[0, 1], pf=[0, 1], remap=[0, 1], blocks=[1024, 2048, 4096, 8192])
    keys = list(grid)
    ref = None
    results = []
    for combo in itertools.product(*[grid[k] for k in keys]):
        cfg = dict(zip(keys, combo))
        for k, v in cfg.items():
            assert L.examg_debug_tune(k.encode(), v) == 0
        ops.fill_random(u, 12345)
        un.zero_()
        run_check()
        torch.cuda.synchronize()
        chk = un.clone()
        if ref is None:
            ref = chk
        same = bool(torch.equal(chk, ref))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        cfg.update(ms=ms, gbs=24.0 * updates / ms / 1e6, same=same)
        results.append(cfg)
        print(json.dumps(cfg), flush=True)
    results.sort(key=lambda r: r["ms"])
    print("\nbest 15:")
    for r in results[:15]:
        print(r)
    bad = [r for r in results if not r["same"]]
    print("variants differing from the first: %d" % len(bad))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "tune_jacobi.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
