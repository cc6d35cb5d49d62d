// oracle/locref.cpp
//
// TEST INFRASTRUCTURE ONLY — CPU oracle for the registration hot path of maotian123/loc_lib.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
// and only as the checker / reported baseline. The product (loc_lib_amd/, include/locgpu.h)
// never links, imports or calls it.
//
// PARITY UNPINNED: the reference ships no golden vectors for this path and
// cannot be built here; this file follows the reference sources line by line instead:
//   ICP   LocUtils/src/model/matching/3d/icp/icp_registration.cpp:57-103 (P2P H,B), :105-159 (P2Line),
//         :161-213 (P2Plane), :216-244 (ScanMatch), :267-381 (Align*)
//   fits  LocUtils/include/LocUtils/common/math_utils.h:112-136 (FitPlane), :138-163 (FitLine),
//         :55-72 (ComputeMeanAndCov)
//   NDT   LocUtils/src/model/matching/3d/ndt/ndt_registration.cpp:51-63 (nearby grids), :87-148 (direct build),
//         :374-464 (AlignNdt), :150-236 + :262-372 (incremental NDT)
//   out   pcl::transformPointCloud(in, out, pose.matrix().cast<float>()) (icp cpp:241, ndt cpp:258)
// Every quirk of SURVEY.md Appendix A is reproduced on purpose (A11-A28).
//
// Build: g++ -std=c++17 -O3 -ffp-contract=off (the reference's flags, LocUtils/CMakeLists.txt:6; x86-64 GCC 7.5
// without -march emits no FMA, hence contraction off).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <list>
#include <thread>
#include <atomic>
#include <set>
#include <unordered_map>
#include <vector>

#include "locref_filters.hpp"
#include "locref_flat.hpp"
#include "locref_kdtree.hpp"
#include "locref_loam.hpp"
#include "locref_math.hpp"

namespace locref {

// ---------------------------------------------------------------------------------------------
// The plane 4-vector from a one-sided Jacobi SVD: columns of `a` = U·Σ, `v` = right singular vectors.
//
// Full-rank case (at least three singular values above 1e-13·σmax — every neighbourhood of a real map): V.col(3) of JacobiSVD
// (math_utils.h:124-125) = the right singular vector of the smallest singular value.
//
// Rank-deficient case (five collinear or coincident neighbours, exactly, in FP64): the null space of [x y z 1] has two or more
// dimensions, every unit vector in it is a valid V.col(3), and which one Eigen's JacobiSVD returns is an accident of its
// rotation sequence on rounding noise — it cannot be restated without Eigen itself. What the reference DOES define there is
// the control flow: any null vector has zero residual on the five points, so the fit passes and effective_num counts the
// point (icp cpp:184). The vector is fixed here by a rule the device kernel shares (loc_lib_amd/csrc/device_math.hpp,
// plane_null_vector): with r = number of dominant singular values, the unit vector of the orthogonal complement of the r
// dominant right singular vectors that is closest to e4 = (0,0,0,1) — or to e3, e2, e1, the first whose projection keeps
// at least 0.4 of its squared length (one always keeps ≥ 0.5). For r = 3 this IS V.col(3).
template <int M>
static void PlaneVectorFromSvd(const double a[4][M], const double v[4][4], double n[4]) {
    double sn[4], mx = 0.0;
    int best = 0;
    double bn = 1e300;
    for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int i = 0; i < M; ++i) s += a[c][i] * a[c][i];
        sn[c] = s;
        if (s < bn) { bn = s; best = c; }
        if (s > mx) mx = s;
    }
    int r = 0;
    bool dom[4];
    for (int c = 0; c < 4; ++c) { dom[c] = sn[c] > 1e-26 * mx; r += dom[c] ? 1 : 0; }
    if (r >= 3) {
        for (int i = 0; i < 4; ++i) n[i] = v[best][i];
        return;
    }
    for (int k = 3; k >= 0; --k) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        t[k] = 1.0;
        for (int pass = 0; pass < 2; ++pass)
            for (int c = 0; c < 4; ++c) {
                if (!dom[c]) continue;
                double d = 0;
                for (int i = 0; i < 4; ++i) d += t[i] * v[c][i];
                for (int i = 0; i < 4; ++i) t[i] -= d * v[c][i];
            }
        double nn = 0;
        for (int i = 0; i < 4; ++i) nn += t[i] * t[i];
        if (nn >= 0.4 || k == 0) {
            const double inv = nn > 0.0 ? 1.0 / std::sqrt(nn) : 0.0;
            for (int i = 0; i < 4; ++i) n[i] = t[i] * inv;
            return;
        }
    }
}

// math_utils.h:112-136  FitPlane<double>: rows [x y z 1], smallest right singular vector, eps check.
static bool FitPlane(const std::vector<V3>& data, double n[4], double eps = 1e-2) {
    if (data.size() < 3) return false;
    if (data.size() != 5 && data.size() != 4) {
        // the reference only ever passes nn.size() in {4,5}; k=5 always yields 5.
        return false;
    }
    double v[4][4];
    if (data.size() == 5) {
        double a[4][5];
        for (int i = 0; i < 5; ++i) { a[0][i] = data[i].x; a[1][i] = data[i].y; a[2][i] = data[i].z; a[3][i] = 1.0; }
        jacobi_svd_onesided<5, 4>(a, v);
        PlaneVectorFromSvd<5>(a, v, n);
    } else {
        double a[4][4];
        for (int i = 0; i < 4; ++i) { a[0][i] = data[i].x; a[1][i] = data[i].y; a[2][i] = data[i].z; a[3][i] = 1.0; }
        jacobi_svd_onesided<4, 4>(a, v);
        PlaneVectorFromSvd<4>(a, v, n);
    }
    for (size_t i = 0; i < data.size(); ++i) {
        const double err = dot(V3{n[0], n[1], n[2]}, data[i]) + n[3];
        if (err * err > eps) return false;
    }
    return true;
}

// FitPlane for exactly five points on fixed arrays — the operations of the five-point branch above, no std::vector (R2).
static bool FitPlane5(const V3 data[5], double n[4], double eps = 1e-2) {
    double a[4][5], v[4][4];
    for (int i = 0; i < 5; ++i) { a[0][i] = data[i].x; a[1][i] = data[i].y; a[2][i] = data[i].z; a[3][i] = 1.0; }
    jacobi_svd_onesided<5, 4>(a, v);
    PlaneVectorFromSvd<5>(a, v, n);
    for (int i = 0; i < 5; ++i) {
        const double err = dot(V3{n[0], n[1], n[2]}, data[i]) + n[3];
        if (err * err > eps) return false;
    }
    return true;
}

// math_utils.h:138-163  FitLine<double>: origin = mean, dir = dominant right singular vector of the centred 5×3.
static bool FitLine(const std::vector<V3>& data, V3& origin, V3& dir, double eps) {
    if (data.size() != 5) return false;
    V3 s{0, 0, 0};
    for (const V3& p : data) s = s + p;
    origin = {s.x / 5.0, s.y / 5.0, s.z / 5.0};
    double a[3][5], v[3][3];
    for (int i = 0; i < 5; ++i) {
        const V3 d = data[i] - origin;
        a[0][i] = d.x; a[1][i] = d.y; a[2][i] = d.z;
    }
    jacobi_svd_onesided<5, 3>(a, v);
    int best = 0; double bn = -1.0;
    for (int c = 0; c < 3; ++c) {
        double sn = 0; for (int i = 0; i < 5; ++i) sn += a[c][i] * a[c][i];
        if (sn > bn) { bn = sn; best = c; }
    }
    dir = {v[best][0], v[best][1], v[best][2]};
    for (const V3& p : data) {
        const V3 c = cross(dir, p - origin);
        if (dot(c, c) > eps) return false;
    }
    return true;
}

struct IcpOptions {  // icp_registration.hpp:22-39 defaults
    int max_iteration = 20;
    double max_nn_distance = 1.0;
    double max_plane_distance = 0.1;
    double max_line_distance = 0.5;
    int min_effective_pts = 10;
    double eps = 1e-2;
    int method = 0;  // 0 P2P, 1 P2LINE, 2 P2PLANE (IcpMethod, hpp:15-20)
};

struct IterTrace {  // one Gauss-Newton iteration, for golden vectors
    double H[36];
    double B[6];
    double dx[6];
    double effective_num;
    double ok;  // 1 when CaculateMatrixHAndB* returned true
};

class Icp {
public:
    KdTree tree;
    std::vector<F3> target;  // the matcher's own deep copy (icp cpp:16); same coordinates as the tree's
    IcpOptions opt;
    KnnStats stats;
    bool count_stats = false;
    FlatKdTree flat;       // R2: the same tree as one flat array (locref_flat.hpp), built on request
    bool use_flat = false;
    bool approximate = true;
    float alpha = 0.1f;

    void SetInputTarget(const float* xyz, size_t n, size_t stride_floats) {
        target.resize(n);
        for (size_t i = 0; i < n; ++i) target[i] = {xyz[i * stride_floats], xyz[i * stride_floats + 1], xyz[i * stride_floats + 2]};
        tree.Build(xyz, n, stride_floats);
        if (use_flat) flat.FromTree(tree);
    }
    void EnableFlat(bool on) {
        use_flat = on;
        if (on && flat.size() != tree.size() && tree.size() > 0) flat.FromTree(tree);
    }

    // icp cpp:161-213 on the flat tree with fixed arrays (R2). Read-only on the object: safe to call from several threads.
    bool HB_P2Plane_flat(const std::vector<F3>& src, const SE3& pose, double H[36], double B[6], double* eff_out) const {
        size_t effective_num = 0;
        const M3 R = rotation_matrix(pose);
        for (size_t i = 0; i < src.size(); ++i) {
            const V3 q = ToVec3d(src[i]);
            const V3 qs = transform(pose, q);
            int nn[FlatKdTree::kMaxK];
            const int cnt = flat.Knn(CastF(qs), 5, approximate, alpha, nn);
            if (cnt > 3) {
                V3 nb[5];
                for (int j = 0; j < 5; ++j) nb[j] = ToVec3d(target[nn[j]]);
                double n[4];
                if (!FitPlane5(nb, n)) continue;
                effective_num++;
                const V3 n3{n[0], n[1], n[2]};
                const double dis = dot(n3, qs) + n[3];
                if (std::fabs(dis) > opt.max_plane_distance) continue;
                const M3 hq = hat(q);
                double nR[3];
                for (int c = 0; c < 3; ++c) nR[c] = -n3.x * R(0, c) + (-n3.y * R(1, c) + -n3.z * R(2, c));  // the binary: 0x58761-0x58801 (PINNING.md)
                double J[1][6];
                for (int c = 0; c < 3; ++c) J[0][c] = (nR[0] * hq(0, c) + nR[1] * hq(1, c)) + nR[2] * hq(2, c);
                J[0][3] = n3.x; J[0][4] = n3.y; J[0][5] = n3.z;
                AddJtJ(H, B, J, 1, &dis);
            }
        }
        if (eff_out) *eff_out = (double)effective_num;
        if (effective_num < (size_t)opt.min_effective_pts) return false;
        double x[6];
        if (lu6_det_solve(H, B, x) == 0) return false;
        return true;
    }

    // AlignP2Plane (icp cpp:345-381) over HB_P2Plane_flat; const ⇒ callable concurrently for different scans (R3)
    int AlignFlat(const std::vector<F3>& src, const SE3& init, SE3& result) const {
        SE3 pose = init;
        int iters = 0;
        for (int iter = 0; iter < opt.max_iteration; ++iter) {
            double H[36] = {0}, err[6] = {0}, dx[6] = {0}, eff = 0;
            const bool ok = HB_P2Plane_flat(src, pose, H, err, &eff);
            ++iters;
            if (ok) {
                lu6_det_solve(H, err, dx);
                apply_update(pose, dx);
                if (norm6(dx) < opt.eps) break;
            }
        }
        result = pose;
        return iters;
    }

    // KdtreeRegistration::FindNearstPoints (kdtree.cpp:272-283)
    std::vector<int> FindNearest(const F3& q, int k) {
        std::vector<int> result;
        tree.GetClosestPoint(q, result, k, count_stats ? &stats : nullptr);
        return result;
    }

    static inline V3 ToVec3d(const F3& p) { return {(double)p.x, (double)p.y, (double)p.z}; }
    static inline F3 CastF(const V3& p) { return {(float)p.x, (float)p.y, (float)p.z}; }

    static void AddJtJ(double H[36], double B[6], const double J[][6], int rows, const double* e) {
        for (int r = 0; r < rows; ++r)
            for (int i = 0; i < 6; ++i) {
                for (int j = 0; j < 6; ++j) H[6 * i + j] += J[r][i] * J[r][j];
                B[i] += -J[r][i] * e[r];
            }
    }

    // icp cpp:57-103
    bool HB_P2P(const std::vector<F3>& src, const SE3& pose, double H[36], double B[6], double* eff_out) {
        size_t effective_num = 0;
        const M3 R = rotation_matrix(pose);
        for (size_t i = 0; i < src.size(); ++i) {
            if (!(std::isfinite(src[i].x) && std::isfinite(src[i].y) && std::isfinite(src[i].z))) continue;
            const V3 q = ToVec3d(src[i]);
            const V3 qs = transform(pose, q);
            std::vector<int> nn = FindNearest(CastF(qs), 1);
            if (!nn.empty()) {
                const V3 p = ToVec3d(target[nn[0]]);
                const V3 e = p - qs;
                const double dis2 = dot(e, e);
                if (dis2 > opt.max_nn_distance) continue;
                effective_num++;
                const M3 Rh = mul(R, hat(q));
                double J[3][6];
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) { J[r][c] = Rh(r, c) / 16; J[r][3 + c] = (r == c) ? -1.0 : 0.0; }
                }
                const double ev[3] = {e.x, e.y, e.z};
                AddJtJ(H, B, J, 3, ev);
            }
        }
        if (eff_out) *eff_out = (double)effective_num;
        if (effective_num < (size_t)opt.min_effective_pts) return false;
        double x[6];
        if (lu6_det_solve(H, B, x) == 0) return false;
        return true;
    }

    // icp cpp:105-159
    bool HB_P2Line(const std::vector<F3>& src, const SE3& pose, double H[36], double B[6], double* eff_out) {
        size_t effective_num = 0;
        const M3 R = rotation_matrix(pose);
        for (size_t i = 0; i < src.size(); ++i) {
            const V3 q = ToVec3d(src[i]);
            const V3 qs = transform(pose, q);
            std::vector<int> nn = FindNearest(CastF(qs), 5);
            if (nn.size() == 5) {
                std::vector<V3> nn_eigen;
                for (int j = 0; j < 5; ++j) nn_eigen.emplace_back(ToVec3d(target[nn[j]]));
                V3 d, p0;
                if (!FitLine(nn_eigen, p0, d, opt.max_line_distance)) continue;
                effective_num++;
                const V3 e = cross(d, qs - p0);  // SO3::hat(d) * (qs - p0)
                if (norm(e) > opt.max_line_distance) continue;
                const M3 hd = hat(d);
                const M3 A = mul(mul(hd, R), hat(q));
                double J[3][6];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { J[r][c] = -A(r, c); J[r][3 + c] = hd(r, c); }
                const double ev[3] = {e.x, e.y, e.z};
                AddJtJ(H, B, J, 3, ev);
            }
        }
        if (eff_out) *eff_out = (double)effective_num;
        if (effective_num < (size_t)opt.min_effective_pts) return false;
        double x[6];
        if (lu6_det_solve(H, B, x) == 0) return false;
        return true;
    }

    // icp cpp:161-213
    bool HB_P2Plane(const std::vector<F3>& src, const SE3& pose, double H[36], double B[6], double* eff_out) {
        size_t effective_num = 0;
        const M3 R = rotation_matrix(pose);
        for (size_t i = 0; i < src.size(); ++i) {
            const V3 q = ToVec3d(src[i]);
            const V3 qs = transform(pose, q);
            std::vector<int> nn = FindNearest(CastF(qs), 5);
            if (nn.size() > 3) {
                std::vector<V3> nn_eigen;
                for (size_t j = 0; j < nn.size(); ++j) nn_eigen.emplace_back(ToVec3d(target[nn[j]]));
                double n[4];
                if (!FitPlane(nn_eigen, n)) continue;
                effective_num++;  // before the residual gate (A14)
                const V3 n3{n[0], n[1], n[2]};
                const double dis = dot(n3, qs) + n[3];
                if (std::fabs(dis) > opt.max_plane_distance) continue;
                // J = [ -n3^T R hat(q) | n3^T ]
                const M3 hq = hat(q);
                double nR[3];
                for (int c = 0; c < 3; ++c) nR[c] = -n3.x * R(0, c) + (-n3.y * R(1, c) + -n3.z * R(2, c));  // the binary: 0x58761-0x58801 (PINNING.md)
                double J[1][6];
                for (int c = 0; c < 3; ++c) J[0][c] = (nR[0] * hq(0, c) + nR[1] * hq(1, c)) + nR[2] * hq(2, c);
                J[0][3] = n3.x; J[0][4] = n3.y; J[0][5] = n3.z;
                AddJtJ(H, B, J, 1, &dis);
            }
        }
        if (eff_out) *eff_out = (double)effective_num;
        if (effective_num < (size_t)opt.min_effective_pts) return false;
        double x[6];
        if (lu6_det_solve(H, B, x) == 0) return false;
        return true;
    }

    bool HB(const std::vector<F3>& src, const SE3& pose, double H[36], double B[6], double* eff) {
        switch (opt.method) {
            case 0: return HB_P2P(src, pose, H, B, eff);
            case 1: return HB_P2Line(src, pose, H, B, eff);
            case 2: return HB_P2Plane(src, pose, H, B, eff);
            default: return true;
        }
    }

    // icp cpp:267-303 / 305-343 / 345-381. Returns number of GN iterations executed.
    int Align(const std::vector<F3>& src, const SE3& init, SE3& result, IterTrace* trace, int trace_cap) {
        SE3 pose = init;
        int iters = 0;
        for (int iter = 0; iter < opt.max_iteration; ++iter) {
            double H[36] = {0}, err[6] = {0}, dx[6] = {0}, eff = 0;
            const bool ok = HB(src, pose, H, err, &eff);
            ++iters;
            bool stop = false;
            if (ok) {
                lu6_det_solve(H, err, dx);
                if (opt.method == 0)
                    for (int i = 0; i < 6; ++i) dx[i] = dx[i] / 16;  // dx = H.inverse()/16 * err (icp cpp:287)
                apply_update(pose, dx);
                if (norm6(dx) < opt.eps) stop = true;
            }
            if (trace && iter < trace_cap) {
                std::memcpy(trace[iter].H, H, sizeof(H));
                std::memcpy(trace[iter].B, err, sizeof(err));
                std::memcpy(trace[iter].dx, dx, sizeof(dx));
                trace[iter].effective_num = eff;
                trace[iter].ok = ok ? 1.0 : 0.0;
            }
            if (stop) break;
        }
        result = pose;
        return iters;
    }
};

// ---------------------------------------------------------------------------------------------
// NDT
struct KeyHash {  // eigen_types.h:104-107 (results do not depend on it; only key equality matters)
    size_t operator()(const std::array<int, 3>& v) const {
        return size_t((((int64_t)v[0] * 73856093) ^ ((int64_t)v[1] * 471943) ^ ((int64_t)v[2] * 83492791)) % 10000000);
    }
};

struct NdtOptions {  // ndt_registration.hpp:27-42
    int max_iteration = 20;
    double voxel_size = 1.0;
    int min_effective_pts = 10;
    int min_pts_in_voxel = 3;
    int max_pts_in_voxel = 50;
    double eps = 1e-2;
    double res_outlier_th = 20.0;
    size_t capacity = 100000;
    int nearby_type = 1;  // 0 CENTER, 1 NEARBY6
    int method = 1;       // 1 DIRECT_NDT, 2 INCREMENTAL_NDT
};

struct NdtVoxel {
    std::vector<size_t> idx;
    std::vector<V3> pts;  // incremental
    bool ndt_estimated = false;
    int num_pts = 0;
    V3 mu{0, 0, 0};
    double sigma[9] = {0};
    double info[9] = {0};
};

// 3×3: info = V diag(1/λ) Uᵀ after clamping λ1,λ2 ≥ 1e-3 λ0 (ndt cpp:118-130).
static void ClampedInfo(const double sigma[9], double info[9]) {
    double a[3][3], v[3][3];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) a[c][r] = sigma[3 * r + c];
    jacobi_svd_onesided<3, 3>(a, v);
    double sv[3];
    int order[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) sv[c] = std::sqrt(a[c][0] * a[c][0] + a[c][1] * a[c][1] + a[c][2] * a[c][2]);
    std::sort(order, order + 3, [&](int i, int j) { return sv[i] > sv[j]; });
    double lam[3] = {sv[order[0]], sv[order[1]], sv[order[2]]};
    double U[3][3], V[3][3];  // [col][row]
    for (int k = 0; k < 3; ++k) {
        const int c = order[k];
        for (int r = 0; r < 3; ++r) {
            V[k][r] = v[c][r];
            // Σ symmetric PSD ⇒ U = V; use the computed left vector when it is well defined.
            U[k][r] = (sv[c] > 0) ? a[c][r] / sv[c] : v[c][r];
        }
    }
    if (lam[1] < lam[0] * 1e-3) lam[1] = lam[0] * 1e-3;
    if (lam[2] < lam[0] * 1e-3) lam[2] = lam[0] * 1e-3;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += V[k][r] * (1.0 / lam[k]) * U[k][c];
            info[3 * r + c] = s;
        }
}

// math_utils.h:55-72 with dim = 3
static void MeanAndCov(const std::vector<V3>& pts, V3& mean, double cov[9]) {
    const size_t len = pts.size();
    V3 s{0, 0, 0};
    for (const V3& p : pts) s = s + p;
    mean = {s.x / (double)len, s.y / (double)len, s.z / (double)len};
    double c[9] = {0};
    for (const V3& p : pts) {
        const V3 d = p - mean;
        const double dv[3] = {d.x, d.y, d.z};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) c[3 * i + j] += dv[i] * dv[j];
    }
    for (int i = 0; i < 9; ++i) cov[i] = c[i] / (double)(len - 1);
}

static void Inverse3(const double m[9], double inv[9]) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    inv[0] = (e * i - f * h) * id; inv[1] = (c * h - b * i) * id; inv[2] = (b * f - c * e) * id;
    inv[3] = (f * g - d * i) * id; inv[4] = (a * i - c * g) * id; inv[5] = (c * d - a * f) * id;
    inv[6] = (d * h - e * g) * id; inv[7] = (b * g - a * h) * id; inv[8] = (a * e - b * d) * id;
}

class Ndt {
public:
    NdtOptions opt;
    double inv_voxel = 1.0;
    std::vector<F3> target;
    using Key = std::array<int, 3>;
    std::unordered_map<Key, NdtVoxel, KeyHash> grids;
    // incremental
    using KeyAndData = std::pair<Key, NdtVoxel>;
    std::list<KeyAndData> data;
    std::unordered_map<Key, std::list<KeyAndData>::iterator, KeyHash> inc_grids;
    bool flag_first_scan = true;
    std::vector<Key> nearby;

    void Configure(const NdtOptions& o) {
        opt = o;
        inv_voxel = 1.0 / opt.voxel_size;  // ctor recomputes it (ndt cpp:15,25)
        nearby.clear();
        if (opt.nearby_type == 0) nearby.push_back({0, 0, 0});
        else nearby = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, 1}};  // ndt cpp:57-58
    }

    Key KeyOf(const V3& p) const {  // (pt * inv).cast<int>(): truncation toward zero (A22)
        return {(int)(p.x * inv_voxel), (int)(p.y * inv_voxel), (int)(p.z * inv_voxel)};
    }

    // ndt cpp:87-148
    void SetDirectTarget(const float* xyz, size_t n, size_t stride) {
        target.resize(n);
        for (size_t i = 0; i < n; ++i) target[i] = {xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2]};
        grids.clear();
        for (size_t idx = 0; idx < n; ++idx) {
            const V3 pt = Icp::ToVec3d(target[idx]);
            grids[KeyOf(pt)].idx.emplace_back(idx);
        }
        for (auto& kv : grids) {
            NdtVoxel& v = kv.second;
            if (v.idx.size() > (size_t)opt.min_pts_in_voxel) {
                std::vector<V3> pts;
                pts.reserve(v.idx.size());
                for (size_t id : v.idx) pts.push_back(Icp::ToVec3d(target[id]));
                MeanAndCov(pts, v.mu, v.sigma);
                ClampedInfo(v.sigma, v.info);
            }
        }
        for (auto it = grids.begin(); it != grids.end();) {
            if (it->second.idx.size() > (size_t)opt.min_pts_in_voxel) ++it;
            else it = grids.erase(it);
        }
    }

    // ndt cpp:150-183
    void SetIncTarget(const float* xyz, size_t n, size_t stride) {
        std::set<Key> active;
        for (size_t i = 0; i < n; ++i) {
            const V3 pt{(double)xyz[i * stride], (double)xyz[i * stride + 1], (double)xyz[i * stride + 2]};
            const Key key = KeyOf(pt);
            auto iter = inc_grids.find(key);
            if (iter == inc_grids.end()) {
                NdtVoxel nv;
                nv.pts.emplace_back(pt);
                nv.num_pts = 1;
                data.push_front({key, nv});
                inc_grids.insert({key, data.begin()});
                if (data.size() >= opt.capacity) {
                    inc_grids.erase(data.back().first);
                    data.pop_back();
                }
            } else {
                NdtVoxel& v = iter->second->second;
                v.pts.emplace_back(pt);
                if (!v.ndt_estimated) v.num_pts++;
                data.splice(data.begin(), data, iter->second);
                iter->second = data.begin();
            }
            active.emplace(key);
        }
        for (const Key& key : active) {
            auto it = inc_grids.find(key);
            if (it == inc_grids.end()) continue;  // evicted by the LRU within this very call (reference would UB via operator[])
            UpdateVoxel(it->second->second);
        }
        flag_first_scan = true;  // ndt cpp:181 (A28)
    }

    // ndt cpp:185-236; with flag_first_scan forced true only the first branch is reachable.
    void UpdateVoxel(NdtVoxel& v) {
        if (flag_first_scan) {
            if (v.pts.size() > 1) {
                MeanAndCov(v.pts, v.mu, v.sigma);
                double m[9];
                for (int i = 0; i < 9; ++i) m[i] = v.sigma[i] + ((i % 4 == 0) ? 1e-3 : 0.0);
                Inverse3(m, v.info);
            } else {
                v.mu = v.pts[0];
                for (int i = 0; i < 9; ++i) v.info[i] = (i % 4 == 0) ? 1e2 : 0.0;
            }
            v.ndt_estimated = true;
            v.pts.clear();
            return;
        }
    }

    // ndt cpp:374-464. status: 0 ok (result written), 1 = det(H)==0 ⇒ `return false` before result_pose is assigned (A26).
    int AlignDirect(const std::vector<F3>& src, const SE3& init, SE3& result, IterTrace* trace, int trace_cap, int* iters_out) {
        SE3 pose = init;
        int iters = 0;
        for (int iter = 0; iter < opt.max_iteration; ++iter) {
            size_t effective_num = 0;
            double H[36] = {0}, err[6] = {0}, dx[6] = {0};
            const M3 R = rotation_matrix(pose);
            for (size_t i = 0; i < src.size(); ++i) {
                const V3 q = Icp::ToVec3d(src[i]);
                const V3 qs = transform(pose, q);
                const Key key = KeyOf(qs);
                for (const Key& off : nearby) {
                    const Key ko{key[0] + off[0], key[1] + off[1], key[2] + off[2]};
                    auto it = grids.find(ko);
                    if (it != grids.end()) {
                        const NdtVoxel& v = it->second;
                        const V3 e = qs - v.mu;
                        const double ev[3] = {e.x, e.y, e.z};
                        // `e.transpose() * v.info_ * e` (ndt cpp:416) is (eᵀ·info)·e by C++ precedence: the 1×3 row vector first — Eigen
                        // evaluates a nested product into a temporary — then its dot product with e
                        double t[3];
                        for (int c = 0; c < 3; ++c) t[c] = (ev[0] * v.info[c] + ev[1] * v.info[3 + c]) + ev[2] * v.info[6 + c];
                        const double res = (t[0] * ev[0] + t[1] * ev[1]) + t[2] * ev[2];
                        if (std::isnan(res) || res > opt.res_outlier_th) continue;
                        const M3 Rh = mul(R, hat(q));
                        double J[3][6];
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) { J[r][c] = -Rh(r, c); J[r][3 + c] = (r == c) ? 1.0 : 0.0; }
                        Icp::AddJtJ(H, err, J, 3, ev);  // NOT info-weighted (A25)
                    }
                }
                effective_num++;  // once per source point (ndt cpp:432)
            }
            ++iters;
            double x[6] = {0};
            const double det = lu6_det_solve(H, err, x);
            if (trace && iter < trace_cap) {
                std::memcpy(trace[iter].H, H, sizeof(H));
                std::memcpy(trace[iter].B, err, sizeof(err));
                std::memset(trace[iter].dx, 0, sizeof(dx));
                trace[iter].effective_num = (double)effective_num;
                trace[iter].ok = (det != 0 && effective_num >= (size_t)opt.min_effective_pts) ? 1.0 : 0.0;
            }
            if (det == 0) { if (iters_out) *iters_out = iters; return 1; }
            if (effective_num < (size_t)opt.min_effective_pts) continue;
            std::memcpy(dx, x, sizeof(dx));
            if (trace && iter < trace_cap) std::memcpy(trace[iter].dx, dx, sizeof(dx));
            apply_update(pose, dx);
            if (norm6(dx) < opt.eps) break;
        }
        result = pose;
        if (iters_out) *iters_out = iters;
        return 0;
    }

    // ndt cpp:262-372. status 0 ok, 2 = effective_num too small (result = last pose, returns false).
    int AlignInc(const std::vector<F3>& src, const SE3& init, SE3& result, IterTrace* trace, int trace_cap, int* iters_out) {
        SE3 pose = init;
        int iters = 0;
        for (int iter = 0; iter < opt.max_iteration; ++iter) {
            double H[36] = {0}, err[6] = {0}, dx[6] = {0};
            int effective_num = 0;
            const M3 R = rotation_matrix(pose);
            for (size_t i = 0; i < src.size(); ++i) {
                const V3 q = Icp::ToVec3d(src[i]);
                const V3 qs = transform(pose, q);
                const Key key = KeyOf(qs);
                for (const Key& off : nearby) {
                    const Key ko{key[0] + off[0], key[1] + off[1], key[2] + off[2]};
                    auto it = inc_grids.find(ko);
                    if (it == inc_grids.end() || !it->second->second.ndt_estimated) continue;
                    const NdtVoxel& v = it->second->second;
                    const V3 e = qs - v.mu;
                    const double ev[3] = {e.x, e.y, e.z};
                    // here the chi² test is formed as e·(info·e): info·e is what the err term below needs anyway (ndt cpp:308 writes the same
                    // `e.transpose() * v.info_ * e` as the direct variant: the two groupings differ by rounding only, and only in this test)
                    double ie[3];
                    for (int r = 0; r < 3; ++r) ie[r] = (v.info[3 * r] * ev[0] + v.info[3 * r + 1] * ev[1]) + v.info[3 * r + 2] * ev[2];
                    const double res = (ev[0] * ie[0] + ev[1] * ie[1]) + ev[2] * ie[2];
                    if (std::isnan(res) || res > opt.res_outlier_th) continue;
                    const M3 Rh = mul(R, hat(q));
                    double J[3][6];
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 3; ++c) { J[r][c] = -Rh(r, c); J[r][3 + c] = (r == c) ? 1.0 : 0.0; }
                    // H += Jᵀ info J ; err += -Jᵀ info e (ndt cpp:345-346)
                    double IJ[3][6];
                    for (int r = 0; r < 3; ++r)
                        for (int c = 0; c < 6; ++c) IJ[r][c] = (v.info[3 * r] * J[0][c] + v.info[3 * r + 1] * J[1][c]) + v.info[3 * r + 2] * J[2][c];
                    for (int a = 0; a < 6; ++a) {
                        for (int b = 0; b < 6; ++b) H[6 * a + b] += (J[0][a] * IJ[0][b] + J[1][a] * IJ[1][b]) + J[2][a] * IJ[2][b];
                        err[a] += -((J[0][a] * ie[0] + J[1][a] * ie[1]) + J[2][a] * ie[2]);
                    }
                    effective_num++;
                }
            }
            ++iters;
            if (trace && iter < trace_cap) {
                std::memcpy(trace[iter].H, H, sizeof(H));
                std::memcpy(trace[iter].B, err, sizeof(err));
                std::memset(trace[iter].dx, 0, sizeof(dx));
                trace[iter].effective_num = effective_num;
                trace[iter].ok = effective_num >= opt.min_effective_pts ? 1.0 : 0.0;
            }
            if (effective_num < opt.min_effective_pts) { result = pose; if (iters_out) *iters_out = iters; return 2; }
            lu6_det_solve(H, err, dx);
            if (trace && iter < trace_cap) std::memcpy(trace[iter].dx, dx, sizeof(dx));
            apply_update(pose, dx);
            if (norm6(dx) < opt.eps) break;
        }
        result = pose;
        if (iters_out) *iters_out = iters;
        return 0;
    }
};

static std::vector<F3> LoadCloud(const float* xyz, size_t n, size_t stride) {
    std::vector<F3> c(n);
    for (size_t i = 0; i < n; ++i) c[i] = {xyz[i * stride], xyz[i * stride + 1], xyz[i * stride + 2]};
    return c;
}

}  // namespace locref

// ---------------------------------------------------------------------------------------------
// C ABI for ctypes (tests / bench cpu_baseline only).
using namespace locref;

extern "C" {

// ---- KD-tree ----
void* locref_kdtree_create(const float* xyz, size_t n, size_t stride_floats) {
    auto* t = new KdTree();
    if (!t->Build(xyz, n, stride_floats)) { delete t; return nullptr; }
    return t;
}
void locref_kdtree_destroy(void* t) { delete (KdTree*)t; }
void locref_kdtree_info(void* tp, int64_t out[3]) {
    auto* t = (KdTree*)tp;
    out[0] = (int64_t)t->size(); out[1] = (int64_t)t->num_nodes(); out[2] = t->depth();
}
// out_idx: nq*k ints (-1 padded). stats_out: [nodes_visited, leaves_visited] totals.
void locref_kdtree_knn(void* tp, const float* q, size_t nq, int k, int approximate, float alpha, int* out_idx,
                       uint64_t* stats_out) {
    auto* t = (KdTree*)tp;
    t->SetEnableANN(approximate != 0, alpha);
    KnnStats st;
    std::vector<int> r;
    for (size_t i = 0; i < nq; ++i) {
        t->GetClosestPoint({q[3 * i], q[3 * i + 1], q[3 * i + 2]}, r, k, stats_out ? &st : nullptr);
        for (int j = 0; j < k; ++j) out_idx[i * k + j] = j < (int)r.size() ? r[j] : -1;
    }
    if (stats_out) { stats_out[0] = st.nodes_visited; stats_out[1] = st.leaves_visited; }
}
// Preorder dump of the tree for structural parity: per node {axis (-1 leaf), thresh bits, point_idx}.
size_t locref_kdtree_dump(void* tp, int32_t* axis, float* thresh, int32_t* point_idx, size_t cap) {
    auto* t = (KdTree*)tp;
    std::vector<const KdNode*> stack{t->root()};
    size_t n = 0;
    while (!stack.empty()) {
        const KdNode* nd = stack.back();
        stack.pop_back();
        if (n < cap) {
            axis[n] = nd->IsLeaf() ? -1 : nd->axis;
            thresh[n] = nd->IsLeaf() ? 0.0f : nd->thresh;
            point_idx[n] = nd->IsLeaf() ? nd->point_idx : -1;
        }
        ++n;
        if (!nd->IsLeaf()) { stack.push_back(nd->right); stack.push_back(nd->left); }
    }
    return n;
}

// ---- fits ----
int locref_fit_plane(const double* pts, int n, double out4[4]) {
    std::vector<V3> d;
    for (int i = 0; i < n; ++i) d.push_back({pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
    return FitPlane(d, out4) ? 1 : 0;
}
int locref_fit_line(const double* pts, int n, double eps, double origin[3], double dir[3]) {
    std::vector<V3> d;
    for (int i = 0; i < n; ++i) d.push_back({pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
    V3 o, di;
    const bool ok = FitLine(d, o, di, eps);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z; dir[0] = di.x; dir[1] = di.y; dir[2] = di.z;
    return ok ? 1 : 0;
}
void locref_clamped_info(const double sigma[9], double info[9]) { ClampedInfo(sigma, info); }
double locref_lu6(const double H[36], const double b[6], double x[6]) { return lu6_det_solve(H, b, x); }
void locref_apply_update(double pose[7], const double dx[6]) {
    SE3 T = se3_from_array(pose);
    apply_update(T, dx);
    se3_to_array(T, pose);
}
void locref_transform_points_f64(const double pose[7], const double* in, size_t n, double* out) {
    const SE3 T = se3_from_array(pose);
    for (size_t i = 0; i < n; ++i) {
        const V3 r = transform(T, {in[3 * i], in[3 * i + 1], in[3 * i + 2]});
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}

// ---- ICP ----
// opts: [max_iteration, max_nn_distance, max_plane_distance, max_line_distance, min_effective_pts, eps]
void* locref_icp_create(int method, const double* opts, int use_ann, float alpha) {
    auto* m = new Icp();
    m->opt.method = method;
    if (opts) {
        m->opt.max_iteration = (int)opts[0]; m->opt.max_nn_distance = opts[1]; m->opt.max_plane_distance = opts[2];
        m->opt.max_line_distance = opts[3]; m->opt.min_effective_pts = (int)opts[4]; m->opt.eps = opts[5];
    }
    m->tree.SetEnableANN(use_ann != 0, alpha);
    m->approximate = use_ann != 0;
    m->alpha = alpha;
    return m;
}
// R2 / R3: point-to-plane alignment of n_scans scans through the flat-array port, `threads` native threads taking
// whole scans (1 = R2). Bit-identical poses to locref_icp_align. iters (optional): GN iterations per scan.
void locref_icp_align_flat(void* mp, const float* const* srcs, const size_t* counts, size_t stride_floats, int n_scans, const double* inits,
                           double* out_poses, int* iters, int threads) {
    auto* m = (Icp*)mp;
    m->EnableFlat(true);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int s = next.fetch_add(1); s < n_scans; s = next.fetch_add(1)) {
            std::vector<F3> cloud = LoadCloud(srcs[s], counts[s], stride_floats);
            SE3 res;
            const int it = m->AlignFlat(cloud, se3_from_array(inits + 7 * s), res);
            se3_to_array(res, out_poses + 7 * s);
            if (iters) iters[s] = it;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}
void locref_icp_destroy(void* m) { delete (Icp*)m; }
void locref_icp_set_target(void* mp, const float* xyz, size_t n, size_t stride_floats) { ((Icp*)mp)->SetInputTarget(xyz, n, stride_floats); }
void locref_icp_tree_info(void* mp, int64_t out[3]) {
    auto* m = (Icp*)mp;
    out[0] = (int64_t)m->tree.size(); out[1] = (int64_t)m->tree.num_nodes(); out[2] = m->tree.depth();
}
// CaculateMatrixHAndB (icp cpp:31-55). Returns the reference's bool; H/B are accumulated from zero.
int locref_icp_hb(void* mp, const float* src, size_t n, size_t stride_floats, const double pose[7], double H[36], double B[6],
                  double* effective_num) {
    auto* m = (Icp*)mp;
    std::vector<F3> s = LoadCloud(src, n, stride_floats);
    std::memset(H, 0, 36 * sizeof(double));
    std::memset(B, 0, 6 * sizeof(double));
    return m->HB(s, se3_from_array(pose), H, B, effective_num) ? 1 : 0;
}
// ScanMatch's align part. trace: trace_cap × 50 doubles (H36,B6,dx6,eff,ok) or null. Returns GN iterations run.
int locref_icp_align(void* mp, const float* src, size_t n, size_t stride_floats, const double init[7], double out_pose[7],
                     double* trace, int trace_cap, uint64_t* visit_stats) {
    auto* m = (Icp*)mp;
    std::vector<F3> s = LoadCloud(src, n, stride_floats);
    SE3 res;
    m->count_stats = visit_stats != nullptr;
    m->stats = KnnStats();
    static_assert(sizeof(IterTrace) == 50 * sizeof(double), "trace layout");
    const int it = m->Align(s, se3_from_array(init), res, (IterTrace*)trace, trace_cap);
    se3_to_array(res, out_pose);
    if (visit_stats) { visit_stats[0] = m->stats.nodes_visited; visit_stats[1] = m->stats.leaves_visited; }
    m->count_stats = false;
    return it;
}

// ---- NDT ----
// opts: [max_iteration, voxel_size, min_effective_pts, min_pts_in_voxel, eps, res_outlier_th, capacity, nearby_type, method]
void* locref_ndt_create(const double* opts) {
    auto* m = new Ndt();
    NdtOptions o;
    if (opts) {
        o.max_iteration = (int)opts[0]; o.voxel_size = opts[1]; o.min_effective_pts = (int)opts[2]; o.min_pts_in_voxel = (int)opts[3];
        o.eps = opts[4]; o.res_outlier_th = opts[5]; o.capacity = (size_t)opts[6]; o.nearby_type = (int)opts[7]; o.method = (int)opts[8];
    }
    m->Configure(o);
    return m;
}
void locref_ndt_destroy(void* m) { delete (Ndt*)m; }
void locref_ndt_set_target(void* mp, const float* xyz, size_t n, size_t stride_floats) {
    auto* m = (Ndt*)mp;
    if (m->opt.method == 2) m->SetIncTarget(xyz, n, stride_floats);
    else m->SetDirectTarget(xyz, n, stride_floats);
}
size_t locref_ndt_num_voxels(void* mp) {
    auto* m = (Ndt*)mp;
    return m->opt.method == 2 ? m->inc_grids.size() : m->grids.size();
}
// Dump voxels: keys[3*i..], mu[3*i..], info[9*i..]; returns count (≤ cap written).
size_t locref_ndt_dump(void* mp, int32_t* keys, double* mu, double* info, size_t cap) {
    auto* m = (Ndt*)mp;
    size_t n = 0;
    auto put = [&](const Ndt::Key& k, const NdtVoxel& v) {
        if (n < cap) {
            keys[3 * n] = k[0]; keys[3 * n + 1] = k[1]; keys[3 * n + 2] = k[2];
            mu[3 * n] = v.mu.x; mu[3 * n + 1] = v.mu.y; mu[3 * n + 2] = v.mu.z;
            std::memcpy(info + 9 * n, v.info, 9 * sizeof(double));
        }
        ++n;
    };
    if (m->opt.method == 2) { for (auto& kv : m->inc_grids) put(kv.first, kv.second->second); }
    else { for (auto& kv : m->grids) put(kv.first, kv.second); }
    return n;
}
// Returns status (0 ok; 1 det(H)==0: out_pose left untouched like the reference; 2 inc-NDT too few points).
int locref_ndt_align(void* mp, const float* src, size_t n, size_t stride_floats, const double init[7], double out_pose[7],
                     double* trace, int trace_cap, int* iters_out) {
    auto* m = (Ndt*)mp;
    std::vector<F3> s = LoadCloud(src, n, stride_floats);
    SE3 res;
    int st;
    if (m->opt.method == 2) st = m->AlignInc(s, se3_from_array(init), res, (IterTrace*)trace, trace_cap, iters_out);
    else st = m->AlignDirect(s, se3_from_array(init), res, (IterTrace*)trace, trace_cap, iters_out);
    if (st != 1) se3_to_array(res, out_pose);
    return st;
}

// ---- output cloud: pcl::transformPointCloud with pose.matrix().cast<float>() (A19) ----
void locref_transform_cloud_f32(const double pose[7], const float* in, size_t n, size_t in_stride, float* out, size_t out_stride) {
    const SE3 T = se3_from_array(pose);
    const M3 R = rotation_matrix(T);
    float m[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[4 * r + c] = (float)R(r, c);
    }
    m[3] = (float)T.t.x; m[7] = (float)T.t.y; m[11] = (float)T.t.z;
    for (size_t i = 0; i < n; ++i) {
        const float x = in[i * in_stride], y = in[i * in_stride + 1], z = in[i * in_stride + 2];
        for (int r = 0; r < 3; ++r) out[i * out_stride + r] = ((m[4 * r] * x + m[4 * r + 1] * y) + m[4 * r + 2] * z) + m[4 * r + 3];
    }
}

// ---- cloud filters either side of the matcher (locref_filters.hpp); points are float32 [n][4] = x, y, z, intensity ----
size_t locref_remove_nan(const float* in, size_t n, int is_dense, float* out) {
    return RemoveNaN((const PointXYZI*)in, n, is_dense != 0, (PointXYZI*)out);
}
size_t locref_crop_box(const float* in, size_t n, int is_dense, const float mn[3], const float mx[3], float* out) {
    return CropBox((const PointXYZI*)in, n, is_dense != 0, mn, mx, (PointXYZI*)out);
}
void locref_box_edges(const float step[3], const float origin[3], float mn[3], float mx[3]) { BoxEdges(step, origin, mn, mx); }
// info[0] = status, info[1..3] = min_b, info[4..6] = div_b
size_t locref_voxel_grid(const float* in, size_t n, int is_dense, float leaf, int order, float* out, int info[7]) {
    VoxelGridInfo I;
    const size_t m = VoxelGrid((const PointXYZI*)in, n, is_dense != 0, leaf, order, (PointXYZI*)out, &I);
    if (info) {
        info[0] = I.status;
        for (int a = 0; a < 3; ++a) { info[1 + a] = I.min_b[a]; info[4 + a] = I.div_b[a]; }
    }
    return m;
}
void locref_transform_cloud_f64(const double pose[7], const float* in, size_t n, int is_dense, float* out) {
    const SE3 T = se3_from_array(pose);
    const M3 R = rotation_matrix(T);
    double m[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[4 * r + c] = R(r, c);
    }
    m[3] = T.t.x; m[7] = T.t.y; m[11] = T.t.z;
    TransformCloudF64((const PointXYZI*)in, n, is_dense != 0, m, (PointXYZI*)out);
}
void* locref_localmap_create(size_t num_kfs, float leaf, int order) {
    LocalMap* m = new LocalMap();
    m->num_kfs = num_kfs; m->leaf = leaf; m->order = order;
    return m;
}
void locref_localmap_destroy(void* h) { delete (LocalMap*)h; }
void locref_localmap_add_keyframe(void* h, const float* pts, size_t n, int is_dense) { ((LocalMap*)h)->AddKeyframe((const PointXYZI*)pts, n, is_dense != 0); }
size_t locref_localmap_size(void* h) { return ((LocalMap*)h)->map.size(); }
int locref_localmap_dense(void* h) { return ((LocalMap*)h)->map_dense ? 1 : 0; }
void locref_localmap_copy(void* h, float* out) { LocalMap* m = (LocalMap*)h; std::memcpy(out, m->map.data(), m->map.size() * sizeof(PointXYZI)); }

// ---- LOAM feature extraction (locref_loam.hpp): pts float32 [n][4] = x, y, z, intensity; ring uint8 [n]. Outputs need room for n points.
void locref_loam_extract(const float* pts, const uint8_t* ring, size_t n, int num_scan, int order, float* edge, size_t* n_edge, float* surf, size_t* n_surf) {
    std::vector<PointXYZI> e, s;
    LoamExtract((const PointXYZI*)pts, ring, n, num_scan, order, e, s);
    std::memcpy(edge, e.data(), e.size() * sizeof(PointXYZI));
    std::memcpy(surf, s.data(), s.size() * sizeof(PointXYZI));
    *n_edge = e.size();
    *n_surf = s.size();
}

}  // extern "C"


