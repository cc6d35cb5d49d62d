#!/usr/bin/env python3
"""Randomised H, B, effective_num, ok parity probe (run ON the GPU box): the map kinds of fuzz_search.py, random scans and poses, all three ICP
methods, GPU (locgpu_icp_hb_batch) vs the oracle. Prints the worst relative error per map kind and any effective_num / ok mismatch.

    python tools/fuzz_hb.py [cases=120] [seed=11]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from loc_lib_amd import api
from oracle import locref
import fuzz_search as fz
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
kinds = ["uniform", "clusters", "sheets", "lines", "dups", "lattice"]
worst = {k: 0.0 for k in kinds}
effbad = 0
singular = 0
for case in range(n_cases):
    kind = kinds[case % 6]
    n = int(10 ** rng.uniform(2.0, 5.5))
    cloud = fz.make_map(rng, kind, n).astype(np.float32)
    ctx = api.Context(0); ctx.icp_set_target(cloud)
    nq = int(10 ** rng.uniform(2.0, 4.3))
    scan = (cloud[rng.integers(0, len(cloud), nq)].astype(np.float64) + rng.normal(0, 10 ** rng.uniform(-3, -0.5), size=(nq, 3))).astype(np.float32)
    q = rng.normal(size=4) * np.array([0.02, 0.02, 0.02, 1.0]); q /= np.linalg.norm(q)
    pose = np.concatenate([q, rng.normal(0, 0.05, size=3)])
    for method in (api.P2PLANE, api.P2LINE, api.P2P):
        icp = locref.Icp(method=method); icp.set_target(cloud)
        opts = api.icp_opts(method=method)
        b = ctx.batch([scan])
        hb = ctx.icp_hb_batch(b, pose[None], opts)[0]
        ok, H, B, eff = icp.hb(scan, pose)
        Hg, Bg, effg, okg = hb[:36].reshape(6, 6), hb[36:42], int(hb[42]), bool(hb[43])
        if eff != effg or ok != okg:
            # `ok` is `effective_num >= min && det(H) != 0` (icp_registration.cpp:204-211). With equal effective_num and an H that is
            # singular to working precision (a handful of correspondences that all constrain the same directions), whether the LU's last
            # pivots come out as exact zeros is an accident of the summation order — in Eigen as much as here: counted apart.
            sv = np.linalg.svd(H, compute_uv=False)
            if eff == effg and sv[-1] <= 1e-13 * sv[0]:
                singular += 1; print("ok differs on a numerically singular H (rank %d, effective_num %d):" % (int((sv > 1e-13 * sv[0]).sum()), eff), case, kind, method, ok, okg, flush=True)
            else:
                effbad += 1; print("EFF/OK differ", case, kind, method, eff, effg, ok, okg, flush=True)
        scale = max(np.abs(H).max(), 1e-300)
        err = max(np.abs(Hg - H).max() / scale, np.abs(Bg - B).max() / max(np.abs(B).max(), 1e-300) if np.abs(B).max() > 0 else 0)
        worst[kind] = max(worst[kind], err)
        b.close()
    del ctx
print("worst relative H/B error by map kind:", {k: float("%.2e" % v) for k, v in worst.items()}, "eff/ok mismatches:", effbad,
      "| ok differs on a numerically singular H (not a mismatch: det == 0 there is a rounding accident on either side):", singular)
