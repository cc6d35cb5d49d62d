#!/usr/bin/env python3
"""Create / use / destroy everything the C ABI hands out a few hundred times (run ON the GPU box) and report what does not come back:
free device memory (hipMemGetInfo) and the process's resident set before and after.

    python tools/leak_check.py [--reps 300]
"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loc_lib_amd import api, synth  # noqa: E402


def n_threads():
    return len(os.listdir("/proc/self/task"))


def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"):
            return int(ln.split()[1]) / 1024.0
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)

    def dev_free_mb():
        hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
        return free.value / 2 ** 20

    m = synth.make_local_map(60000, 3, half=40.0)
    full = synth.make_scan(3, crop_half=36.0)
    scan = full[::4].copy()
    xyzi = np.concatenate([full[:, :3], np.zeros((len(full), 1), np.float32)], axis=1).astype(np.float32)
    _, init = synth.make_pose(3)

    def once():
        ctx = api.Context(0)
        ctx.icp_set_target(m)
        ctx.ndt_set_target(m)
        ctx.icp_align(scan, init, api.icp_opts(method=api.P2PLANE))
        ctx.ndt_align(scan, init)
        ctx.icp_scan_match(scan, init, api.icp_opts(method=api.P2PLANE))   # ScanMatch whole: helper thread, piece events, pinned staging
        ctx.ndt_scan_match(scan, init)
        ctx.transform_cloud(init, scan)
        b = ctx.batch([scan, scan[::2].copy()])
        ctx.icp_align_batch(b, np.stack([init] * 2), api.icp_opts(method=api.P2LINE))
        ctx.graph_enable(True)
        ctx.icp_align_batch(b, np.stack([init] * 2), api.icp_opts(method=api.P2PLANE))
        ctx.graph_enable(False)
        b.close()
        c = api.Cloud(ctx, xyzi)
        f = c.voxel_filter(0.5)
        sub = api.Submap(ctx, 3, 0.5) if hasattr(api, "Submap") else None
        if sub is not None:
            sub.add_keyframe(f, init)
            ctx.icp_set_target_cloud(sub.cloud(), wait=False)
            ctx.icp_align_cloud(f, init, api.icp_opts(method=api.P2PLANE))
            sub.close()
        f.close(); c.close()
        ctx.close()

    for _ in range(5):
        once()  # warm: runtime pools, kernel code objects, the host build pool
    d0, r0, t0 = dev_free_mb(), rss_mb(), n_threads()
    for i in range(a.reps):
        once()
        if (i + 1) % 100 == 0:
            print("after %4d: device free %+.1f MB, RSS %+.1f MB" % (i + 1, dev_free_mb() - d0, rss_mb() - r0), flush=True)
    d1, r1 = dev_free_mb(), rss_mb()
    print("reps %d: device memory not returned %.1f MB (%.3f MB per repetition), RSS growth %.1f MB (%.3f MB per repetition), threads %+d"
          % (a.reps, d0 - d1, (d0 - d1) / a.reps, r1 - r0, (r1 - r0) / a.reps, n_threads() - t0))


if __name__ == "__main__":
    main()
