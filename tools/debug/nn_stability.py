#!/usr/bin/env python3
"""How many 5-NN lists survive from one Gauss–Newton iteration to the next? (Would a per-point plane cache pay in the fit kernel?)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loc_lib_amd import api, synth
import ctypes

ctx = api.Context(0)
ctx.icp_set_target(synth.make_map(10_000_000))
ids = [0, 37, 90, 200]
scans = [synth.make_scan(i) for i in ids]
poses = np.stack([synth.make_pose(i)[1] for i in ids])
b = ctx.batch(scans)
opts = api.icp_opts(method=api.P2PLANE)
prev = None
lib = api.lib()
for it in range(10):
    hb = ctx.icp_hb_batch(b, poses, opts)
    nn = ctx.debug_batch_nn(b, 5)
    if prev is not None:
        same = np.all(nn == prev, axis=2)
        same_set = np.all(np.sort(nn, axis=2) == np.sort(prev, axis=2), axis=2)
        w = same.reshape(len(ids), -1, 64).all(axis=2)
        print("iter %d: lists unchanged %.3f (as sets %.3f); whole 64-point waves unchanged %.3f" % (it, same.mean(), same_set.mean(), w.mean()))
    prev = nn
    stop_all = True
    for s in range(len(ids)):
        pose = poses[s].copy(); dx = np.zeros(6); ap = ctypes.c_int(0); st = ctypes.c_int(0)
        lib.locgpu_gn_update(hb[s].ctypes.data, 2, 10, ctypes.c_double(1e-2), pose.ctypes.data, dx.ctypes.data, ctypes.byref(ap), ctypes.byref(st))
        poses[s] = pose
        print("   scan %d |dx| %.4f stop %d" % (s, np.linalg.norm(dx), st.value))
