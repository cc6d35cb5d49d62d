#!/usr/bin/env python3
"""Randomised parity hunt for the cloud filters either side of the matcher (run ON the GPU box): random clouds (sizes 1 … 2·10^6 so that both
the one-read-back and the early-read-back voxel paths run, boxes from centimetres to kilometres, offsets far from the origin, NaN / ±inf
sprinkled in, dense flag right and wrong), random leaf sizes (down to PCL's "leaf size is too small" rule) and crop boxes: the resident
cloud entry points (locgpu_cloud_voxel_filter / crop_box / remove_nan / transform, chained like Lio::AddCloud chains them) must return the
oracle's clouds byte for byte.

    python tools/fuzz_filters.py [--cases 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api  # noqa: E402
from oracle import locref  # noqa: E402


def make_cloud(rng, n):
    kind = rng.integers(0, 5)
    scale = 10 ** rng.uniform(-1.5, 3.0)
    if kind == 0:
        p = rng.uniform(-scale, scale, size=(n, 3))
    elif kind == 1:
        p = rng.normal(0, scale, size=(n, 3)) * np.array([1.0, 1.0, 0.05])
    elif kind == 2:  # a grid whose points sit exactly ON voxel borders for commensurate leaves
        m = max(1, int(round(n ** (1 / 3))))
        g = np.arange(m) * rng.choice([0.125, 0.25, 0.5, 1.0])
        p = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)[:n]
        p = p - rng.choice([0.0, 1.0, 7.5])
    elif kind == 3:
        c = rng.uniform(-scale, scale, size=(max(1, n // 200), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.02 * scale, size=(n, 3))
    else:
        p = rng.uniform(-scale, scale, size=(n, 3)) + rng.choice([0.0, 1e3, -2e4]) * rng.choice([0.0, 1.0], size=3)
    out = np.zeros((len(p), 4), np.float32)
    out[:, :3] = p
    out[:, 3] = rng.uniform(0, 255, len(p))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    ctx = api.Context(0)
    bad = 0
    seen = dict(points=0, passthrough=0, non_dense=0, empty_out=0, big=0)
    t0 = time.time()
    raw, tmp, out = api.Cloud(ctx), api.Cloud(ctx), api.Cloud(ctx)
    for case in range(a.cases):
        n = int(10 ** rng.uniform(0.0, 6.3 if case % 10 == 0 else 5.3))
        cloud = make_cloud(rng, n)
        n = len(cloud)
        dirty = case % 3 != 0
        if dirty and n > 3:
            k = max(1, n // int(rng.integers(3, 50)))
            cloud[rng.integers(0, n, k), rng.integers(0, 3, k)] = rng.choice([np.nan, np.inf, -np.inf], k)
        flag_dense = (not dirty) if case % 7 else bool(rng.integers(0, 2))  # sometimes the flag lies, as PCL lets it
        if flag_dense and dirty:
            cloud = np.nan_to_num(cloud, nan=1.0, posinf=2.0, neginf=-2.0)  # a dense-flagged cloud with non-finite points is UB in PCL too
        ext = float(np.nanmax(np.abs(np.where(np.isfinite(cloud[:, :3]), cloud[:, :3], 0)))) + 1e-3
        leaf = float(ext * 10 ** rng.uniform(-3.2, 0.3))
        tag = "case %d n %d dense %d leaf %.4g" % (case, n, flag_dense, leaf)
        try:
            raw.upload(cloud, is_dense=flag_dense)
            # 1. voxel filter of the raw cloud
            got, passthrough = raw.voxel_filter(leaf, out=out, with_passthrough=True)
            want, info = locref.voxel_grid(cloud, flag_dense, leaf, order=locref.SORT_STABLE, with_info=True)
            g = got.download()
            seen["points"] += n; seen["passthrough"] += int(passthrough); seen["non_dense"] += int(not flag_dense)
            seen["empty_out"] += int(len(g) == 0); seen["big"] += int(n >= (1 << 20))
            if info["status"] == 1:
                ok = passthrough and np.array_equal(g, cloud, equal_nan=True)
            else:
                ok = (not passthrough) and np.array_equal(g, want, equal_nan=True)
            if not ok:
                bad += 1
                print("MISMATCH voxel", tag, "status", info["status"], "passthrough", passthrough, len(g), len(want), flush=True)
            # 2. removeNaN → crop box → voxel filter, the chain of the front-ends
            raw.remove_nan(out=tmp)
            w1 = locref.remove_nan(cloud, flag_dense)
            if not np.array_equal(tmp.download(), w1, equal_nan=True):
                bad += 1
                print("MISMATCH remove_nan", tag, flush=True)
            mn = rng.uniform(-ext, 0.2 * ext, 3).astype(np.float32)
            mx = (mn + rng.uniform(0, 1.5 * ext, 3)).astype(np.float32)
            tmp.crop_box(mn, mx, out=out)
            w2 = locref.crop_box(w1, True, mn, mx)  # removeNaN leaves a dense-flagged cloud either way
            g2 = out.download()
            if not np.array_equal(g2, w2, equal_nan=True):
                bad += 1
                print("MISMATCH crop", tag, len(g2), len(w2), flush=True)
            elif len(w2):
                out.voxel_filter(leaf, out=tmp)
                w3, info3 = locref.voxel_grid(w2, True, leaf, order=locref.SORT_STABLE, with_info=True)
                g3 = tmp.download()
                if not np.array_equal(g3, w2 if info3["status"] == 1 else w3, equal_nan=True):
                    bad += 1
                    print("MISMATCH chain voxel", tag, len(g3), len(w3), info3["status"], flush=True)
            # 3. transform with a double pose (lio.cpp:244,279)
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            pose = np.concatenate([q, rng.normal(0, ext, 3)])
            raw.transform(pose, out=tmp)
            w4 = locref.transform_cloud_f64(pose, cloud, is_dense=flag_dense)
            if not np.array_equal(tmp.download(), w4, equal_nan=True):
                bad += 1
                print("MISMATCH transform", tag, flush=True)
        except api.LocGpuError as e:
            bad += 1
            print("ERROR", tag, str(e)[:200], flush=True)
    print("cases %d, mismatches %d, %.1f s; seen %s" % (a.cases, bad, time.time() - t0, seen))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
