"""Helper of tests/test_gpu_configs.py::test_secular_plane_fit_agrees_with_the_four_column_fit: a ragged eight-scan P2Plane batch
aligned once, plus one H/B evaluation at the resulting poses; poses, iteration counts and H/B go to the .npz named on the command
line. The library reads LOCGPU_PLANE_FIT once per process, so the two fits are two processes."""
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
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    opts = api.icp_opts(method=api.P2PLANE)
    res = {}
    b = ctx.batch(scans)
    res["pose"], st = ctx.icp_align_batch(b, inits, opts)
    res["it"] = np.array([s["iterations"] for s in st])
    res["hb"] = ctx.icp_hb_batch(b, res["pose"], opts)
    b.close()
    ctx.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
