#!/usr/bin/env python3
"""Randomised parity probe for the LOAM feature picker (run ON the GPU box): random ring-structured clouds — smooth walls, corners, depth
jumps, flat runs whose curvatures tie exactly, duplicated points, rings just above and below the 131-point minimum, rings missing,
points of the rings interleaved — GPU (locgpu_cloud_loam_extract) vs the oracle with ties broken by id: the same edge and surface
points in the same order, byte for byte.

    python tools/fuzz_loam.py [--cases 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api  # noqa: E402
from oracle import locref  # noqa: E402


def ring_points(rng, n, elev):
    """One ring: range as a function of azimuth made of smooth pieces, jumps and exactly flat runs."""
    az = np.sort(rng.uniform(-np.pi, np.pi, n)) if rng.integers(0, 2) else np.linspace(-np.pi, np.pi, n, endpoint=False)
    r = np.full(n, rng.uniform(3, 40))
    pos = 0
    while pos < n:
        seg = int(rng.integers(3, max(4, n // 3)))
        kind = rng.integers(0, 5)
        sl = slice(pos, min(n, pos + seg))
        m = sl.stop - sl.start
        if kind == 0:
            r[sl] = rng.uniform(3, 60)                                  # constant range: exactly tied curvatures
        elif kind == 1:
            r[sl] = rng.uniform(3, 60) + np.linspace(0, rng.uniform(-5, 5), m)  # a wall
        elif kind == 2:
            r[sl] = rng.uniform(3, 60) + rng.normal(0, rng.choice([0.005, 0.05, 0.5]), m)  # noise at several scales
        elif kind == 3:
            r[sl] = r[max(pos - 1, 0)]                                  # continue the previous value
        else:
            r[sl] = rng.uniform(3, 60) * (1 + 0.3 * np.sin(np.linspace(0, rng.uniform(1, 20), m)))
        pos += seg
    x, y, z = r * np.cos(az) * np.cos(elev), r * np.sin(az) * np.cos(elev), r * np.sin(elev)
    return np.stack([x, y, z, rng.uniform(0, 255, n)], axis=1).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = api.Context(0)
    bad = 0
    seen = dict(points=0, edge=0, surf=0, empty=0)
    t0 = time.time()
    for case in range(a.cases):
        num_scan = int(rng.choice([16, 32, 64]))
        rings, ids = [], []
        for rg in range(num_scan):
            if rng.random() < 0.15:
                continue  # a missing ring
            n = int(rng.choice([int(rng.integers(1, 131)), 131, 132, int(rng.integers(133, 400)), int(rng.integers(400, 2500))], p=[0.1, 0.05, 0.05, 0.4, 0.4]))
            p = ring_points(rng, n, np.deg2rad(-15 + 30 * rg / num_scan))
            if rng.random() < 0.2:
                k = rng.integers(0, n, max(1, n // 20))
                p[k] = p[np.maximum(k - 1, 0)]  # duplicated neighbours: zero differences
            rings.append(p)
            ids.append(np.full(n, rg, np.uint8))
        if not rings:
            continue
        cloud, ring = np.concatenate(rings), np.concatenate(ids)
        if case % 3 == 0:  # interleave the rings (a driver that emits columns, not rings): order within a ring is kept
            key = rng.random(len(cloud))
            order = np.argsort(np.concatenate([np.sort(key[ring == r]) for r in np.unique(ring)]), kind="stable")
            cloud, ring = cloud[order], ring[order]
        e_ref, s_ref = locref.loam_extract(cloud, ring, num_scan, order=locref.SORT_STABLE)
        try:
            edge, surf = api.Cloud(ctx, cloud).loam_extract(ring, num_scan)
            e, s = edge.download(), surf.download()
        except api.LocGpuError as ex:
            bad += 1
            print("ERROR case %d: %s" % (case, str(ex)[:160]), flush=True)
            continue
        seen["points"] += len(cloud); seen["edge"] += len(e_ref); seen["surf"] += len(s_ref); seen["empty"] += int(len(e_ref) + len(s_ref) == 0)
        if not (np.array_equal(e, e_ref) and np.array_equal(s, s_ref)):
            bad += 1
            print("MISMATCH case %d num_scan %d points %d: edge %d/%d surf %d/%d" % (case, num_scan, len(cloud), len(e), len(e_ref), len(s), len(s_ref)), flush=True)
    print("cases %d, mismatches %d, %.1f s; seen %s" % (a.cases, bad, time.time() - t0, seen))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
