#!/usr/bin/env python3
"""Randomised alignment parity probe (run ON the GPU box): local maps and scans of the synthetic world at random places, random initial
perturbations (up to 1 m / 8°), all three ICP methods and direct NDT — final pose and iteration count, GPU vs the oracle.

    python tools/fuzz_align.py [--cases 40]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api, synth  # noqa: E402
from oracle import locref  # noqa: E402


def pose_delta(a, b):
    dt = float(np.linalg.norm(a[4:] - b[4:]))
    qa, qb = a[:4] / np.linalg.norm(a[:4]), b[:4] / np.linalg.norm(b[:4])
    return dt, 2.0 * float(np.arccos(min(1.0, abs(float(np.dot(qa, qb))))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    worst_t = worst_r = 0.0
    it_bad = 0
    for case in range(a.cases):
        sid = int(rng.integers(0, 256))
        m = synth.make_local_map(int(10 ** rng.uniform(4.5, 5.6)), sid, half=40.0)
        scan = synth.make_scan(sid, subsample=int(10 ** rng.uniform(3.0, 4.3)), crop_half=36.0)
        _, init = synth.make_pose(sid, trans_amp=float(10 ** rng.uniform(-2, 0)), rot_amp_deg=float(10 ** rng.uniform(-1, 0.9)), seed=int(rng.integers(1, 1 << 30)))
        ctx = api.Context(0)
        ctx.icp_set_target(m)
        ctx.ndt_set_target(m)
        for method in (api.P2PLANE, api.P2LINE, api.P2P, -1):
            if method >= 0:
                ref = locref.Icp(method=method)
                ref.set_target(m)
                want = ref.align(scan, init)
                got, st = ctx.icp_align(scan, init, api.icp_opts(method=method))
                if case % 2 == 1:  # the same scan as a batch large enough for the batch search kernel (> 2048 waves)
                    copies = (2049 * 64 + len(scan) - 1) // len(scan)
                    b = ctx.batch([scan] * copies)
                    bp, bst = ctx.icp_align_batch(b, np.stack([init] * copies), api.icp_opts(method=method))
                    b.close()
                    for j in (0, copies - 1):
                        dtb, drb = pose_delta(bp[j], want["pose"])
                        worst_t, worst_r = max(worst_t, dtb), max(worst_r, drb)
                        if bst[j]["iterations"] != want["iters"]:
                            it_bad += 1
                            print("ITERATIONS differ (batch): case %d method %d gpu %d oracle %d" % (case, method, bst[j]["iterations"], want["iters"]), flush=True)
            else:
                ref = locref.Ndt()
                ref.set_target(m)
                want = ref.align(scan, init)
                got, st = ctx.ndt_align(scan, init)
            dt, dr = pose_delta(np.asarray(got), want["pose"])
            worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
            if st["iterations"] != want["iters"]:
                it_bad += 1
                print("ITERATIONS differ: case %d method %d gpu %d oracle %d (pose delta %.2e m)" % (case, method, st["iterations"], want["iters"], dt), flush=True)
                if method >= 0:  # what is wrong: the tree on the device, the call, or the context?
                    q = np.ascontiguousarray(scan[:, :3], dtype=np.float32)
                    kd = locref.KdTree(m)
                    n_bad = int((ctx.knn(q, k=5, approximate=True) != kd.knn(q, k=5, approximate=True)).any(axis=1).sum())
                    again, st2 = ctx.icp_align(scan, init, api.icp_opts(method=method))
                    ok_o, Ho, Bo, eff_o = ref.hb(scan, init)
                    hb = ctx.icp_hb(scan, init, api.icp_opts(method=method))
                    print("   device tree vs oracle tree: %d of %d 5-NN lists of the raw scan differ; the same call again: %d iterations; H/B at the initial pose: effective_num %d vs %d, |dH| %.2e of |H|"
                          % (n_bad, len(q), st2["iterations"], hb[3], eff_o, float(np.abs(hb[1] - Ho).max() / max(np.abs(Ho).max(), 1e-300))), flush=True)
        del ctx
    print("cases %d x 4 methods: worst pose delta %.2e m / %.2e rad, iteration-count mismatches %d" % (a.cases, worst_t, worst_r, it_bad))
    return 0


if __name__ == "__main__":
    sys.exit(main())
