#!/usr/bin/env python3
"""Find small 'straight lines' maps on which the hot search kernel's stack outgrows its rows (the candidate descent): prints, per (seed, n),
how many ANN lists differ from the oracle's. LOCGPU_LIB selects the library under test."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api  # noqa: E402
from oracle import locref  # noqa: E402


def lines_map(rng, n):
    t = rng.uniform(-100, 100, size=(n, 1))
    d = rng.normal(size=(8, 3))
    o = rng.uniform(-30, 30, size=(8, 3))
    i = rng.integers(0, 8, n)
    return (o[i] + t * d[i] / np.linalg.norm(d[i], axis=1, keepdims=True)).astype(np.float32)


for seed in range(1, 7):
    for n in (100_000, 300_000, 1_000_000):
        rng = np.random.default_rng(seed)
        cloud = lines_map(rng, n)
        nq = 20000
        scan = (cloud[rng.integers(0, n, nq)].astype(np.float64) + rng.normal(0, 3.0, size=(nq, 3))).astype(np.float32)
        scans = [scan] * 8  # > 2048 waves: the batch kernel, not the one-scan kernel
        pose = np.array([0, 0, 0, 1, 0.3, -0.2, 0.1])
        ctx = api.Context(0)
        ctx.icp_set_target(cloud)
        tree = locref.KdTree(cloud)
        b = ctx.batch(scans)
        opts = api.icp_opts(method=api.P2PLANE)
        ctx.icp_hb_batch(b, np.stack([pose] * 8), opts)
        got = ctx.debug_batch_nn(b, 5)[0, :nq]
        qq = locref.transform_points(pose, np.ascontiguousarray(scan[:, :3], dtype=np.float64)).astype(np.float32)
        want = tree.knn(qq, 5, approximate=True, alpha=0.1)
        print("seed", seed, "n", n, "depth", tree.depth, "differing", int(np.sum(np.any(got != want, axis=1))), flush=True)
        b.close()
        ctx.close()
