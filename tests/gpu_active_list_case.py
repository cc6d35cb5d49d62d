"""Helper of tests/test_gpu_configs.py::test_later_chunks_launch_only_the_open_scans: aligns a fixed ragged batch whose scans need from 3 to
more than 12 Gauss–Newton iterations and writes poses + iteration counts to the .npz named on the command line (the library reads
LOCGPU_ACTIVE_LIST once per process, so the two settings are two processes)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, synth  # noqa: E402


def main(out):
    m = synth.make_local_map(200000, 3, half=40.0)
    s10 = synth.make_scan(3, subsample=10000, crop_half=36.0)
    s2 = synth.make_scan(3, subsample=2000, crop_half=36.0)
    truth, init = synth.make_pose(3)
    scans = [s10, s2, s10[::3], s10[::2], s2[::2], s10[1::2], s10[:6000], s2[:1500], s10[500:9000]]
    inits = np.stack([init] * len(scans))
    inits[0] = truth                                  # converges at once
    inits[2, 4:] += [0.5, -0.4, 0.1]                  # far: many iterations
    inits[5, 4:] += [-0.7, 0.3, 0.0]
    inits[7, 4:] += [0.25, 0.25, -0.05]
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    res = {}
    for name, method in (("plane", api.P2PLANE), ("line", api.P2LINE), ("point", api.P2P)):
        b = ctx.batch(scans)
        poses, st = ctx.icp_align_batch(b, inits, api.icp_opts(method=method))
        res[name] = poses
        res[name + "_it"] = np.array([s["iterations"] for s in st])
        # and two of them in flight
        b2 = ctx.batch(scans[::-1])
        ctx.icp_align_batch_begin(b, inits, api.icp_opts(method=method))
        ctx.icp_align_batch_begin(b2, inits[::-1].copy(), api.icp_opts(method=method))
        p1, _ = ctx.align_batch_end(b)
        p2, _ = ctx.align_batch_end(b2)
        res[name + "_flight"] = np.concatenate([p1, p2[::-1]])
        b.close(); b2.close()
    ctx.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
