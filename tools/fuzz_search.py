#!/usr/bin/env python3
"""Randomised parity hunt for the hot search kernel (run ON the GPU box): random maps (uniform, clustered, planar sheets, lines, duplicated
points, lattices), random scans and poses, ANN and exact pruning, k = 5 and 1 — the neighbour lists of icp_search_walk_kernel (read back with
locgpu_debug_batch_nn) must equal the oracle's KdTree::GetClosestPoint lists index for index. Odd cases run as a batch large enough for
the batch kernel (walk + deep pass), even ones through the one-scan kernel; LOCGPU_FAST_STACK=12 makes the rare paths common.

    python tools/fuzz_search.py [--cases 200] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api  # noqa: E402
from oracle import locref  # noqa: E402


def make_map(rng, kind, n):
    if kind == "uniform":
        return rng.uniform(-50, 50, size=(n, 3))
    if kind == "clusters":
        c = rng.uniform(-80, 80, size=(max(2, n // 500), 3))
        return c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.05, 2.0), size=(n, 3))
    if kind == "sheets":
        p = rng.uniform(-60, 60, size=(n, 3))
        ax = rng.integers(0, 3, n)
        p[np.arange(n), ax] = np.round(p[np.arange(n), ax] / 20.0) * 20.0 + rng.normal(0, 0.01, n)
        return p
    if kind == "lines":
        t = rng.uniform(-100, 100, size=(n, 1))
        d = rng.normal(size=(8, 3))
        o = rng.uniform(-30, 30, size=(8, 3))
        i = rng.integers(0, 8, n)
        return o[i] + t * d[i] / np.linalg.norm(d[i], axis=1, keepdims=True)
    if kind == "dups":
        base = rng.uniform(-30, 30, size=(max(1, n // 4), 3))
        return np.repeat(base, 4, axis=0)[:n]
    if kind == "lattice":
        m = max(2, int(round(n ** (1 / 3))))
        g = np.arange(m, dtype=np.float64) * rng.choice([0.25, 0.5, 1.0])
        return np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    if kind == "geometric":  # coordinates spread over many octaves: the mean split peels off a few points per level — very deep trees
        e = rng.integers(0, int(rng.integers(8, 45)), size=(n, 3))
        return rng.choice([-1.0, 1.0], size=(n, 3)) * 100.0 * 2.0 ** (-e.astype(np.float64)) * rng.uniform(1.0, 1.3, size=(n, 3))
    if kind == "offset":  # a small box far from the origin: float32 has ≈1e-3 … 1e-2 m resolution there, distances tie all the time
        return rng.uniform(-10, 10, size=(n, 3)) + rng.choice([1e4, 3e4, 1e5]) * rng.choice([-1.0, 1.0], size=3)
    if kind == "rings":  # what a spinning LiDAR leaves on flat ground
        r = rng.choice(np.linspace(2.0, 60.0, 16), n)
        th = rng.uniform(0, 2 * np.pi, n)
        return np.stack([r * np.cos(th), r * np.sin(th), rng.normal(0, 0.005, n)], axis=1)
    raise ValueError(kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--grid", action="store_true", help="also run the exact grid search (search_mode = SEARCH_GRID_EXACT) against the exact oracle lists")
    ap.add_argument("--only", type=int, default=-1, help="run only this case of the sequence (the random stream is still consumed case by case)")
    ap.add_argument("--dump", type=str, default="", help="with --only: save the case (cloud, scan, pose) to this .npz")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    kinds = ["uniform", "clusters", "sheets", "lines", "dups", "lattice", "geometric", "offset", "rings"]  # an odd number: every kind meets both kernels
    bad = 0
    t0 = time.time()
    for case in range(a.cases):
        kind = kinds[case % len(kinds)]
        n = int(10 ** rng.uniform(1.0, 6.2))
        cloud = make_map(rng, kind, n).astype(np.float32)
        if len(cloud) < 1:
            continue
        if a.only >= 0 and case != a.only:  # consume exactly what the case would have consumed
            nq = int(10 ** rng.uniform(1.0, 4.7))
            rng.integers(0, len(cloud), nq); rng.uniform(-3, 1); rng.normal(0, 1.0, size=(nq, 3)); rng.normal(size=4); rng.uniform(-3, 0.5); rng.normal(0, 1.0, size=3)
            continue
        ctx = api.Context(0)
        nq = int(10 ** rng.uniform(1.0, 4.7))
        src = cloud[rng.integers(0, len(cloud), nq)].astype(np.float64) + rng.normal(0, 10 ** rng.uniform(-3, 1), size=(nq, 3))
        scan = src.astype(np.float32)
        try:
            ctx.icp_set_target(cloud)
        except api.LocGpuError as e:  # depth > 64: refused by design
            print("case %d kind %s n %d: target refused (%s)" % (case, kind, n, str(e)[:80]), flush=True)
            rng.normal(size=4); rng.uniform(-3, 0.5); rng.normal(0, 1.0, size=3)  # what the case would have consumed (--only stays aligned)
            continue
        tree = locref.KdTree(cloud)
        q = rng.normal(size=4) * np.array([0.05, 0.05, 0.05, 1.0])
        q /= np.linalg.norm(q)
        pose = np.concatenate([q, rng.normal(0, 10 ** rng.uniform(-3, 0.5), size=3)])
        if a.dump and case == a.only:
            np.savez(a.dump, cloud=cloud, scan=scan, pose=pose)
        # every other case (when the scan is large enough) as a batch of copies that exceeds 2048 waves: the batch kernel (un-stored
        # top levels, deep pass) instead of the one-scan kernel (every level stored)
        copies = (2049 * 64 + nq - 1) // nq if (case % 2 == 1 and nq >= 2100) else 1
        b = ctx.batch([scan] * copies)
        for method, k in ((api.P2PLANE, 5), (api.P2P, 1)):
            for approximate in (1, 0):
                if k > tree.num_leaves:
                    continue
                opts = api.icp_opts(method=method)
                opts.approximate = approximate
                try:
                    ctx.icp_hb_batch(b, np.stack([pose] * copies), opts)
                except api.LocGpuError:
                    continue
                got = ctx.debug_batch_nn(b, k)[copies - 1, :nq]
                qq = locref.transform_points(pose, np.ascontiguousarray(scan[:, :3], dtype=np.float64)).astype(np.float32)
                want = tree.knn(qq, k, approximate=bool(approximate), alpha=0.1)
                if a.grid and not approximate:
                    gopts = api.icp_opts(method=method, search_mode=api.SEARCH_GRID_EXACT)
                    try:
                        ctx.icp_hb_batch(b, np.stack([pose] * copies), gopts)
                        gg = ctx.debug_batch_nn(b, k)[copies - 1, :nq]
                        if not np.array_equal(gg, want):
                            bad += 1
                            print("MISMATCH (grid) case %d kind %s n %d nq %d k %d: %d lists differ" % (case, kind, n, nq, k, int(np.sum(np.any(gg != want, axis=1)))), flush=True)
                    except api.LocGpuError as e:
                        print("grid refused case %d kind %s: %s" % (case, kind, str(e)[:100]), flush=True)
                if not np.array_equal(got, want):
                    bad += 1
                    print("MISMATCH case %d kind %s n %d leaves %d depth %d nq %d k %d approx %d: %d lists differ" % (
                        case, kind, n, tree.num_leaves, tree.depth, nq, k, approximate, int(np.sum(np.any(got != want, axis=1)))), flush=True)
        b.close()
        del ctx
    print("cases %d, mismatching runs %d, %.1f s" % (a.cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
