"""Helper of tests/test_gpu_configs.py::test_paced_one_scan_alignment_equals_the_chunked_one: one-scan alignments — P2Plane, P2P,
direct NDT, incremental NDT; a start that converges at once, far starts, a run cut short by max_iteration, a two-point scan, a new
upload right behind a call — whose poses, iteration counts and stats go to the .npz named on the command line. The library reads
LOCGPU_PACE_AHEAD once per process, so the paced and the chunked loop are two processes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from loc_lib_amd import api, synth  # noqa: E402


def main(out):
    m = synth.make_local_map(400000, 3, half=40.0)
    full = synth.make_scan(3, crop_half=36.0)
    truth, init = synth.make_pose(3)
    far = init.copy()
    far[4:] += [0.5, -0.4, 0.1]
    ctx = api.Context(0)
    ctx.icp_set_target(m)
    ctx.ndt_set_target(m)
    poses, its, stats = [], [], []

    def keep(p, st):
        poses.append(np.asarray(p).reshape(-1)[:7].copy())
        its.append(st[0]["iterations"])
        stats.append([st[0]["converged"], st[0]["status"], st[0]["last_effective_num"], st[0]["last_dx_norm"]])

    b = ctx.batch([full])
    for start in (init, truth, far):
        for method in (api.P2PLANE, api.P2P, api.P2LINE):
            keep(*ctx.icp_align_batch(b, start, api.icp_opts(method=method)))
        keep(*ctx.ndt_align_batch(b, start))
    keep(*ctx.icp_align_batch(b, far, api.icp_opts(method=api.P2PLANE, max_iteration=3)))   # cut short
    keep(*ctx.icp_align_batch(b, far, api.icp_opts(method=api.P2PLANE, max_iteration=1)))
    # a new upload right behind a call (the iterations still queued behind a finished scan must not read it), shorter scans
    for cut in (2, 3, 5):
        b.upload_async([np.ascontiguousarray(full[::cut])])
        keep(*ctx.icp_align_batch(b, init, api.icp_opts(method=api.P2PLANE)))
        keep(*ctx.ndt_align_batch(b, init))
    b.upload_async([np.ascontiguousarray(full[:2])])                                         # two points: never `ok`, runs to max_iteration
    keep(*ctx.icp_align_batch(b, init, api.icp_opts(method=api.P2PLANE)))
    b.close()
    # the host-pointer call the façade's ScanMatch makes, and the incremental-NDT matcher
    for start in (init, far):
        p, st = ctx.icp_align(full, start, api.icp_opts(method=api.P2PLANE))
        keep(p, [st] if isinstance(st, dict) else st)
    ctx.close()
    np.savez(out, pose=np.stack(poses), it=np.array(its), stats=np.array(stats, dtype=np.float64))


if __name__ == "__main__":
    main(sys.argv[1])
