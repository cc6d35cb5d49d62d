#!/bin/bash
# Re-reads what oracle/PINNING.md quotes from the reference's prebuilt binary (build container only: /root/reference is not on the GPU
# box). Nothing is executed or loaded: objdump prints the instruction windows behind each row of the table, and a few constants are read
# from .rodata.      bash tools/pinning_audit.sh [row …]      rows: split leafdist needexpand knn transform dis fitplane nR p2p p2line
#                                                              fitline ndt_chi2 dx norm exp quat ndt_stats ndt_jtj outcloud consts
SO=${LOCUTILS_SO:-/root/reference/LocUtils/libs/libLocUtils.so}
[ -r "$SO" ] || { echo "no $SO here"; exit 2; }
win() { echo "== $1  [$2, $3)"; objdump -d --no-show-raw-insn --start-address=$2 --stop-address=$3 "$SO" | grep -E '^\s+[0-9a-f]+:' | cut -c1-96; }
rows=${@:-split leafdist needexpand knn transform dis fitplane nR p2p p2line fitline ndt_chi2 dx norm exp quat ndt_stats ndt_jtj outcloud consts}
for r in $rows; do case $r in
  split)      win "KdTree::FindSplitAxisAndThresh: f32 sums in index order, first arg-max, thresh > x -> left" 0x9e2d0 0x9e460;;
  leafdist)   win "KdTree::ComputeDisForLeaf: dx^2 + (dy^2 + dz^2), push / push-then-pop iff top > dist2" 0x9e8d3 0x9e951;;
  needexpand) win "KdTree::NeedExpand: d*d against top*alpha (approximate) or top" 0x9e0a0 0x9e0f0;;
  knn)        win "KdTree::Knn: thresh > q[axis] -> left first, NeedExpand, then the other side" 0x9ebe0 0x9ec80;;
  transform)  win "Eigen Quaternion::_transformVector (SE3 * q)" 0x62e90 0x62fd0;;
  dis)        win "P2Plane dis = ((qs.x n0 + qs.y n1) + qs.z n2) + n3 and its gate" 0x5869a 0x586f2;;
  fitplane)   win "FitPlane residual ((p.x n0 + p.y n1) + p.z n2) + n3, squared, > eps" 0x79e65 0x79e9a;;
  nR)         win "P2Plane -n^T R: (-n0) R0c + ((-n1) R1c + (-n2) R2c); then * hat(q): (c0 + c1) + c2" 0x58717 0x58890;;
  p2p)        win "P2P dis2 = (ex^2 + ey^2) + ez^2 and its gate" 0x5792f 0x57976;;
  p2line)     win "P2Line e = hat(d) (qs - p0), e.norm() > max_line_distance" 0x590ea 0x591b1;;
  fitline)    win "FitLine |dir x (d - origin)|^2 = (cx^2 + cy^2) + cz^2 > eps" 0x7a3dd 0x7a471;;
  ndt_chi2)   win "direct NDT res = (e^T info) e (out-of-line Eigen redux, called from AlignNdt 0x828b1 and the incremental lambda 0x85c5d)" 0x7b690 0x7b72b;;
  dx)         win "dx = H^-1 err: compute_inverse<Matrix6d> then row sums left to right" 0x5ae2a 0x5afd3;;
  norm)       win "dx.norm(): p0 + (p1 + p2) over three packets, then low + high; < eps" 0x5b113 0x5b18f;;
  exp)        win "Sophus SO3::expAndTheta: theta = sqrt((x^2 + y^2) + z^2), Taylor iff theta < 1e-10" 0x66572 0x66700;;
  quat)       win "pose.so3() * exp: SSE2 quaternion product, squared norm (z^2 + x^2) + (w^2 + y^2), * 2/(sq + 1) iff sq != 1" 0x5afd8 0x5b113;;
  ndt_stats)  win "SetDirectNdtTargetCloud: size > min_pts, mean (index order) / len, cov / (len - 1), SVD, lambda clamp 1e-3, 1/lambda" 0x7f624 0x7f9f0;;
  ndt_jtj)    win "AlignNdt J^T J entries: (J0a J0b + J1a J1b) + J2a J2b" 0x82c06 0x82ce0;;
  outcloud)   win "ScanMatch: transformPointCloud inlined, f32: ((x m0 + y m1) + z m2) + m3" 0x5b4f0 0x5b581;;
  consts)     python3 - "$SO" <<'PY'
import re, struct, subprocess, sys
so = sys.argv[1]
out = subprocess.run(["readelf", "-S", "-W", so], capture_output=True, text=True).stdout
secs = [(int(m.group(3), 16), int(m.group(4), 16), int(m.group(5), 16)) for m in re.finditer(r"\]\s+(\S+)\s+(\S+)\s+([0-9a-f]{16})\s+([0-9a-f]{6,})\s+([0-9a-f]{6,})", out)]
data = open(so, "rb").read()
def rd(a):
    for va, off, sz in secs:
        if va <= a < va + sz:
            return struct.unpack("<d", data[off + a - va:off + a - va + 8])[0]
for a, what in ((0x11ea90, "FitPlane eps at the P2Plane call site"), (0x11ea50, "SO3::exp threshold on theta"), (0x11ea58, "1/48"), (0x11ea60, "1/3840"),
                (0x11ea68, "1/8"), (0x11ea70, "1/384"), (0x11ea98, "2.0 of the renormalisation"), (0x11f1f8, "NDT lambda clamp factor")):
    print("== .rodata 0x%x = %r  (%s)" % (a, rd(a), what))
PY
  ;;
  *) echo "unknown row $r";;
esac; done
