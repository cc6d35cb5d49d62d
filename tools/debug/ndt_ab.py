#!/usr/bin/env python3
"""A/B of two builds of the library on the direct-NDT path: writes the poses of a fixed 16-scan batch (and of single scans) to the .npy named on
the command line; LOCGPU_LIB selects the build. Compare the two files with numpy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api, synth  # noqa: E402

m = synth.make_map(1_000_000)
scans = [synth.make_scan(i, subsample=30000) for i in range(16)]
inits = np.stack([synth.make_pose(i)[1] for i in range(16)])
ctx = api.Context(0)
out = []
for kw in (dict(), dict(nearby_type=api.CENTER), dict(voxel_size=2.0)):
    ctx.ndt_set_target(m, api.ndt_opts(**kw))
    b = ctx.batch(scans)
    p, st = ctx.ndt_align_batch(b, inits)
    out.append(p)
    out.append(np.array([[s["iterations"], s["last_effective_num"], 0, 0, 0, 0, 0] for s in st], dtype=np.float64))
    b.close()
    out.append(np.stack([ctx.ndt_align(scans[i], inits[i])[0] for i in range(3)]))
np.save(sys.argv[1], np.concatenate(out))
