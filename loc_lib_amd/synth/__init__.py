"""Synthetic world ``cityblock-v1``: ctypes binding of ``cityblock.cpp``.

Deterministic map / scan / pose generator shared by the tests, ``bench.py`` and the CPU baseline.
Host-only utility; not part of the registration hot path.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcityblock.so")

WORLD_SEED = 20240901
SCAN_SEED_BASE = 777
PERTURB_SEED = 4242
MAP_SIGMA = 0.01
RANGE_SIGMA = 0.02
N_BEAMS = 64
N_AZ = 1800


def build(force=False):
    """Compile libcityblock.so in-tree (g++)."""
    src = os.path.join(_HERE, "cityblock.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-o", _LIB_PATH, src])
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.cityblock_map.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t]
        lib.cityblock_map.restype = None
        lib.cityblock_scan.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t]
        lib.cityblock_scan.restype = ctypes.c_size_t
        lib.cityblock_pose.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                       ctypes.c_void_p]
        lib.cityblock_pose.restype = None
        lib.cityblock_map_local.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t]
        lib.cityblock_map_local.restype = None
        lib.cityblock_sensor.argtypes = [ctypes.c_int, ctypes.c_void_p]
        lib.cityblock_sensor.restype = None
        _lib = lib
    return _lib


def make_map(n, seed=WORLD_SEED, sigma=MAP_SIGMA, stride=3):
    """n map points, float32 [n, stride] (xyz in the first three columns, rest zero)."""
    out = np.zeros((n, stride), dtype=np.float32)
    _load().cityblock_map(seed, n, sigma, out.ctypes.data, stride)
    return out


def sensor_position(scan_id):
    """(x, y, z, yaw) of the sensor for ``scan_id``."""
    out = np.zeros(4)
    _load().cityblock_sensor(scan_id, out.ctypes.data)
    return out


def make_local_map(n, scan_id, half=30.0, seed=WORLD_SEED, sigma=MAP_SIGMA, stride=3):
    """n map points inside the ±half square around scan ``scan_id``'s sensor (a box-cropped local map)."""
    c = sensor_position(scan_id)
    out = np.zeros((n, stride), dtype=np.float32)
    _load().cityblock_map_local(seed, n, sigma, c[0], c[1], half, out.ctypes.data, stride)
    return out


def make_scan(scan_id, n_beams=N_BEAMS, n_az=N_AZ, world_seed=WORLD_SEED, range_sigma=RANGE_SIGMA, stride=3, subsample=None,
              crop_half=None):
    """Scan ``scan_id`` of the circuit in the sensor frame, float32 [m, stride], ring-major.

    ``crop_half``: keep only returns whose true world position lies within ±crop_half (x, y) of the sensor — pairs a scan
    with ``make_local_map(half > crop_half)`` the way the reference's ±150 m local map always contains its scans.
    """
    out = np.zeros((n_beams * n_az, stride), dtype=np.float32)
    m = _load().cityblock_scan(world_seed, scan_id, SCAN_SEED_BASE + scan_id, n_beams, n_az, range_sigma, out.ctypes.data, stride)
    out = out[:m]
    if crop_half is not None:
        yaw = sensor_position(scan_id)[3]
        c, s_ = np.cos(yaw), np.sin(yaw)
        wx = c * out[:, 0] - s_ * out[:, 1]
        wy = s_ * out[:, 0] + c * out[:, 1]
        out = np.ascontiguousarray(out[(np.abs(wx) <= crop_half) & (np.abs(wy) <= crop_half)])
        m = len(out)
    if subsample is not None and subsample < m:
        idx = np.floor(np.arange(subsample) * (m / subsample)).astype(np.int64)
        out = np.ascontiguousarray(out[idx])
    return out


def make_pose(scan_id, trans_amp=0.3, rot_amp_deg=2.0, seed=PERTURB_SEED):
    """(true_pose7, init_pose7): quaternion xyzw + translation (Sophus::SE3d::data() order)."""
    t = np.zeros(7)
    i = np.zeros(7)
    _load().cityblock_pose(scan_id, seed, trans_amp, np.deg2rad(rot_amp_deg), t.ctypes.data, i.ctypes.data)
    return t, i
