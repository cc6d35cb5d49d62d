"""Helper of tests/test_gpu_configs.py::test_plane_cache_gives_the_uncached_kernels_bits: a ragged batch large enough for the 64-lane
search kernel (so that the P2Plane fit kernel runs with its plane cache) aligned blocking, as two alignments in flight, through hipGraph
replay, and with some scans dropping out after the first chunk; poses, iteration counts and one H/B evaluation go to the .npz named on
the command line. The library reads LOCGPU_PLANE_CACHE once per process, so the two settings are two processes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, synth  # noqa: E402


def main(out):
    m = synth.make_local_map(400000, 3, half=40.0)
    full = synth.make_scan(3, crop_half=36.0)  # every return within the local map: a few tens of thousands of points
    truth, init = synth.make_pose(3)
    scans = [full, full[::2], full[1::3], full[: len(full) // 2], full[5::2], full[::5], full[100:], full[::-1].copy()]
    inits = np.stack([init] * len(scans))
    inits[1] = truth                                   # converges at once
    inits[2, 4:] += [0.5, -0.4, 0.1]                   # far: more than eight iterations
    inits[5, 4:] += [-0.6, 0.3, 0.0]
    assert ((len(full) + 63) // 64) * len(scans) > 2048, "batch too small for the 64-lane search kernel: the cache would not run"
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    res = {}
    b = ctx.batch(scans)
    res["pose"], st = ctx.icp_align_batch(b, inits, opts)
    res["it"] = np.array([s["iterations"] for s in st])
    res["pose_again"], _ = ctx.icp_align_batch(b, inits, opts)          # the cache of the first alignment must not leak into the second
    res["hb"] = ctx.icp_hb_batch(b, res["pose"], opts)
    b2 = ctx.batch(scans[::-1])
    ctx.icp_align_batch_begin(b, inits, opts)
    ctx.icp_align_batch_begin(b2, inits[::-1].copy(), opts)
    p1, _ = ctx.align_batch_end(b)
    p2, _ = ctx.align_batch_end(b2)
    res["flight"] = np.concatenate([p1, p2[::-1]])
    ctx.graph_enable(True)
    res["graph"], _ = ctx.icp_align_batch(b, inits, opts)
    res["graph_again"], _ = ctx.icp_align_batch(b, inits, opts)
    ctx.graph_enable(False)
    line = ctx.icp_align_batch(b, inits, api.icp_opts(method=api.P2LINE))[0]  # another method on the same batch in between
    res["line"] = line
    res["pose_after_line"], _ = ctx.icp_align_batch(b, inits, opts)
    b.close(); b2.close()
    ctx.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
