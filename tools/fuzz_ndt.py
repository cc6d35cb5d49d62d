#!/usr/bin/env python3
"""Randomised parity probe for the NDT half of the path (run ON the GPU box): direct and incremental NDT (ndt_registration.cpp), random
NdtOptions (voxel size 0.3 … 3 m, CENTER / NEARBY6, min_pts_in_voxel, res_outlier_th, capacity small enough for the incremental map's
LRU to evict), local maps and scans of the synthetic world at random places and perturbations — voxel count, final pose, status and
iteration count, GPU vs the oracle.

Direct NDT with min_pts_in_voxel < 3 (the default is 3) keeps voxels of two or three points, whose covariance has rank 1 or 2. The
reference inverts it as V·diag(1/λ')·Uᵀ from Eigen's JacobiSVD (ndt_registration.cpp:118-130): for a singular value that is rounding
noise, U's column is the negated V column about half the time, and the "information" matrix gets a large NEGATIVE eigenvalue. Which
voxels that happens to is decided by the last bit of the covariance sums — it cannot be reproduced by anything but the same
instruction sequence on the same sums (the oracle restates it; since round 5 the device forms the same sums in the same order — the
cases that then agree completely are counted). Those cases are counted separately ("rank-deficient voxels"), required to agree in
their voxel counts only, and stated as a deviation in INTEGRATION.md.

    python tools/fuzz_ndt.py [--cases 60]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api, synth  # noqa: E402
from oracle import locref  # noqa: E402


def pose_delta(a, b):
    return float(np.linalg.norm(a[4:] - b[4:])), float(np.abs(np.abs(a[:4]) - np.abs(b[:4])).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--only", type=int, default=-1, help="run only this case and compare the voxel tables as well")
    ap.add_argument("--seed", type=int, default=9)
    ap.add_argument("--inc-only", action="store_true", help="execute the incremental-NDT cases only")
    ap.add_argument("--first", type=int, default=0, help="skip the cases before this one (their random draws are still consumed)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    worst_t = worst_q = 0.0
    bad = degenerate = degenerate_equal = unstable = 0
    for case in range(a.cases):
        sid = int(rng.integers(0, 256))
        m = synth.make_local_map(int(10 ** rng.uniform(4.3, 5.5)), sid, half=40.0)
        scan = synth.make_scan(sid, subsample=int(10 ** rng.uniform(3.0, 4.3)), crop_half=36.0)
        _, init = synth.make_pose(sid, trans_amp=float(10 ** rng.uniform(-2, 0.2)), rot_amp_deg=float(10 ** rng.uniform(-1, 0.9)), seed=int(rng.integers(1, 1 << 30)))
        method = api.DIRECT_NDT if case % 3 else api.INCREMENTAL_NDT
        kw = dict(voxel_size=float(10 ** rng.uniform(-0.5, 0.5)), nearby_type=int(rng.integers(0, 2)), min_pts_in_voxel=int(rng.integers(1, 8)),
                  res_outlier_th=float(rng.choice([5.0, 20.0, 100.0])), min_effective_pts=int(rng.choice([10, 200])), max_iteration=int(rng.choice([4, 20, 30])))
        cap = int(rng.choice([300, 3000, 100000]))
        if (a.only >= 0 and case != a.only) or case < a.first or (a.inc_only and method != api.INCREMENTAL_NDT):
            continue
        ctx = api.Context(0)
        ref = locref.Ndt(method=method, capacity=cap, **kw)
        try:
            ctx.ndt_set_target(m, api.ndt_opts(method=method, capacity=cap, **kw))
            ref.set_target(m)
            if method == api.INCREMENTAL_NDT:  # a second, shifted cloud: updates of existing voxels + evictions at small capacity
                m2 = synth.make_local_map(len(m) // 2, (sid + 1) % 256, half=40.0)
                ctx.ndt_set_target(m2, api.ndt_opts(method=method, capacity=cap, **kw))
                ref.set_target(m2)
            nv_g, nv_o = ctx.ndt_target_info()["num_voxels"], ref.num_voxels()
            want = ref.align(scan, init)
            got, st = ctx.ndt_align(scan, init)
        except api.LocGpuError as e:
            bad += 1
            print("ERROR case %d %s: %s" % (case, kw, str(e)[:160]), flush=True)
            del ctx
            continue
        if a.only >= 0:
            kg, mug, ig = ctx.ndt_dump()
            ko, muo, io = ref.dump()
            og, oo = np.lexsort(kg.T[::-1]), np.lexsort(ko.T[::-1])
            same_keys = kg.shape == ko.shape and np.array_equal(kg[og], ko[oo])
            print("keys equal:", same_keys, "voxels", len(kg), len(ko))
            if same_keys:
                scale = np.abs(io[oo]).max(axis=(1, 2), keepdims=True) + 1e-300
                rel = (np.abs(ig[og] - io[oo]) / scale).max(axis=(1, 2))
                print("mu max abs diff %.3e; info rel diff: max %.3e, voxels over 1e-6: %d" % (np.abs(mug[og] - muo[oo]).max(), rel.max(), int((rel > 1e-6).sum())))
                w = np.argsort(rel)[::-1][:3]
                for j in w:
                    print("  voxel", kg[og][j], "rel", rel[j], "\n   gpu info", ig[og][j].ravel(), "\n   ref info", io[oo][j].ravel())
            else:
                sg, so = set(map(tuple, kg)), set(map(tuple, ko))
                print("  only on GPU:", len(sg - so), "only in oracle:", len(so - sg))
            print("trace (oracle):", want["trace"][:8, :3] if "trace" in want else None)
            print("gpu stats", st, "oracle iters", want["iters"], "status", want["status"])
        dt, dq = pose_delta(np.asarray(got), want["pose"])
        if method == api.DIRECT_NDT and kw["min_pts_in_voxel"] < 3:
            degenerate += 1
            if nv_g != nv_o:
                bad += 1
                print("MISMATCH case %d (rank-deficient voxels): voxels %d/%d" % (case, nv_g, nv_o), flush=True)
            # round 5: the device's covariance sums are the oracle's bits, so these cases may agree as well — counted, not required
            if nv_g == nv_o and st["iterations"] == want["iters"] and st["status"] == want["status"] and dt <= 1e-7 and dq <= 1e-7:
                degenerate_equal += 1
            del ctx
            continue
        if nv_g != nv_o or st["iterations"] != want["iters"] or st["status"] != want["status"] or dt > 1e-7 or dq > 1e-7:
            # Is the reference itself stable here? An alignment against a handful of voxels (most dropped by min_pts_in_voxel) has
            # near-singular normal equations: a 1e-9 change of the initial pose sends the ORACLE somewhere else, and the last-bit
            # differences between the device's per-point sums (tree-shaped block reductions) and the oracle's sequential ones do the
            # same to the device. Such a case is counted as "reference unstable", not as a mismatch.
            init2 = np.array(init, dtype=np.float64)
            init2[4:] += 1e-9
            want2 = ref.align(scan, init2)
            dt2, dq2 = pose_delta(want2["pose"], want["pose"])
            if nv_g == nv_o and (want2["iters"] != want["iters"] or want2["status"] != want["status"] or dt2 > 1e-6 or dq2 > 1e-6):
                unstable += 1
                del ctx
                continue
            bad += 1
            print("MISMATCH case %d method %d cap %d %s: voxels %d/%d iterations %d/%d status %d/%d pose delta %.2e m %.2e" % (
                case, method, cap, kw, nv_g, nv_o, st["iterations"], want["iters"], st["status"], want["status"], dt, dq), flush=True)
        worst_t, worst_q = max(worst_t, dt), max(worst_q, dq)
        del ctx
    print("cases %d: mismatches %d, worst pose delta %.2e m / %.2e (quaternion components); %d direct cases with rank-deficient voxels compared by voxel count only (%d of them agree in pose, iterations and status all the same); %d cases where the oracle itself is unstable (1e-9 on the initial pose changes its result)" % (
        a.cases, bad, worst_t, worst_q, degenerate, degenerate_equal, unstable))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
