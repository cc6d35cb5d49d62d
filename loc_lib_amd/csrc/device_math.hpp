// loc_lib_amd/csrc/device_math.hpp
//
// Device-side fixed-size FP64 algebra for the registration kernels (gfx950). All device code in this
// library is compiled with -ffp-contract=off: the float32 search arithmetic must round exactly like the
// reference's FMA-free x86-64 build (kdtree.cpp:197-236), and keeping FP64 un-fused as well makes the
// per-point math reproducible against the CPU oracle to the last bit (only summation order differs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace locgpu {

struct D3 { double x, y, z; };

__device__ __forceinline__ D3 operator+(const D3& a, const D3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(const D3& a, const D3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, const D3& a) { return {s * a.x, s * a.y, s * a.z}; }
// Eigen's fixed-size-3 reduction order: x0 + (x1 + x2).
// Vector3d reductions as the reference's binary evaluates them: (x + y) + z (one SSE2 packet, then the scalar tail: read off LocUtils/libs/libLocUtils.so, e.g. 0x5869a)
__device__ __forceinline__ double dot3(const D3& a, const D3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ D3 cross3(const D3& a, const D3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Per-scan pose state kept in HBM between Gauss–Newton iterations.
struct PoseState {
    double q[4];   // x y z w  (Sophus::SE3d::data() order)
    double t[3];
    double R[9];   // row-major rotation matrix of q (refreshed by the solve kernel)
    double last_dx_norm;
    long long last_eff;
    int iterations;  // H,B evaluations so far
    int converged;   // left through |dx| < eps
    int done;        // 1: no further iterations for this scan
    int status;      // 0 ok, 1 NDT det(H)==0 (reference returns before writing result_pose)
};

// SE3 * p the way Sophus 1.0 does it: Eigen Quaternion::_transformVector, then + t
// (call sites: icp_registration.cpp:68,113,169; ndt_registration.cpp:403).
__device__ __forceinline__ D3 se3_apply(const double* q, const double* t, const D3& v) {
    const D3 qv{q[0], q[1], q[2]};
    D3 uv = cross3(qv, v);
    uv = uv + uv;
    const D3 r = (v + q[3] * uv) + cross3(qv, uv);
    return {r.x + t[0], r.y + t[1], r.z + t[2]};
}

// Eigen::Quaternion::toRotationMatrix.
__device__ __host__ inline void quat_to_R(const double* q, double* R) {
    const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// 1/√x for x > 0 to double precision: v_rsq_f64 (≈2^-23 relative) refined by ONE third-order step,
// y·(1 + e/2 + 3e²/8) with e = 1 − x·y² (error ∝ e³ ≈ 2^-67): five instructions instead of the seven of two Newton steps.
__device__ __forceinline__ double rsqrt_refined(double x) {
#pragma clang fp contract(fast)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), y, 1.0);
    const double q = __builtin_fma(0.375, e, 0.5) * e;
    return __builtin_fma(y, q, y);
}

// One-sided (Hestenes) Jacobi SVD, M×N, columns in a[N][M], right vectors accumulated in v[N][N].
// Fully unrolled over (p,q) so both arrays live in VGPRs; the sweep loop exits per lane when a sweep made no rotation.
// This is FP64-issue bound, so FMA contraction is allowed HERE (results differ from the un-fused CPU oracle by
// rounding only, ~1e-16 relative; the float32 search arithmetic stays un-fused).
// `norms` (optional): the squared column norms of the result (the converged sweep's fresh sums — no rotation touched them).
template <int M, int N, bool ACC_V = true>
__device__ __forceinline__ void jacobi_svd_onesided(double (&a)[N][M], double (&v)[N][N], double* norms = nullptr) {
#pragma clang fp contract(fast)
    if (ACC_V) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        // Squared column norms: summed afresh once per sweep, then carried through the sweep's rotations by the exact identities
        // α' = α − tγ, β' = β + tγ (they follow from γt² + (β−α)t − γ = 0). A pair then costs one inner product instead of three.
        // The carried values only steer the rotation angle and scale the convergence test; orthogonality itself is always
        // tested on a freshly summed γ, so the accuracy of the result is that of the plain scheme.
        double nrm[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
            double sn = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) sn += a[c][i] * a[c][i];
            nrm[c] = sn;
        }
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double alpha = nrm[p], beta = nrm[q];
                double gamma = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) gamma += a[p][i] * a[q][i];
                // converged pair: |γ| ≤ 1e-15·√(αβ)  ⇔  γ² ≤ 1e-30·αβ (no square root on the common path)
                if (!(gamma == 0.0 || gamma * gamma <= 1e-30 * (alpha * beta))) {
                    rotated = true;
                    // Jacobi rotation that zeroes γ (the inner one, |θ| ≤ π/4). With d = β−α and r = √(d²+4γ²):
                    //   c² = ½ + ½|d|/r,  s = sign(d)·γ/(r·c),  t = s/c  (t solves γt² + dt − γ = 0)
                    // — two refined v_rsq_f64 (1/r and 1/c), no sqrt, no divide; c ≥ 1/√2, so nothing cancels.
                    const double d = beta - alpha;
                    const double x2 = d * d + 4.0 * (gamma * gamma);   // > 0 here (γ ≠ 0)
                    const double ir = rsqrt_refined(x2);
                    const double c2 = 0.5 + (0.5 * fabs(d)) * ir;
                    const double ic = rsqrt_refined(c2);
                    const double c = c2 * ic;
                    const double gs = d >= 0.0 ? gamma : -gamma;
                    const double s = (gs * ir) * ic;
                    const double t = s * ic;
                    nrm[p] = alpha - t * gamma;
                    nrm[q] = beta + t * gamma;
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const double ap = a[p][i], aq = a[q][i];
                        a[p][i] = c * ap - s * aq;
                        a[q][i] = s * ap + c * aq;
                    }
                    if (ACC_V) {
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            const double vp = v[p][i], vq = v[q][i];
                            v[p][i] = c * vp - s * vq;
                            v[q][i] = s * vp + c * vq;
                        }
                    }
                }
            }
        }
        if (!rotated) {
            if (norms) {
#pragma unroll
                for (int c = 0; c < N; ++c) norms[c] = nrm[c];
            }
            return;
        }
    }
    if (norms) {  // sweep limit reached (never seen): sum them
#pragma unroll
        for (int c = 0; c < N; ++c) {
            double sn = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) sn += a[c][i] * a[c][i];
            norms[c] = sn;
        }
    }
}

// Right singular vector of the smallest singular value of the 5×4 matrix A = [x y z 1] (rows = the five neighbours) — what
// math::FitPlane takes from JacobiSVD (math_utils.h:124-127: V.col(3)), up to its sign, which cancels in H and B.
//
// A = QR (modified Gram–Schmidt; only R is kept), then one-sided Jacobi on the COLUMNS of Rᵀ: Rᵀ·J = W·Σ, so A's right singular
// vectors are the normalised columns of the rotated matrix itself — no accumulation of rotations — and the preconditioning
// cuts the sweeps a wave pays (its maximum over 64 lanes) from ≈4.1 to ≈3.2 on real neighbourhoods, on 4-vectors instead of
// 5-vectors. The wanted vector is then the unit vector orthogonal to the three columns of LARGEST norm (a generalised cross
// product): as accurate as those three are, whatever the size of the smallest singular value (an exactly planar neighbourhood
// leaves a column that is rounding noise). Measured against an extended-precision reference on 4 000 real neighbourhoods: max
// relative error 2.8e-13, median 6.5e-16 (direct one-sided Jacobi on A: 2.0e-13 / 9.9e-16; LAPACK gesdd: 7.5e-13 / 1.7e-15).
__device__ __forceinline__ void plane_null_vector(const D3 (&nb)[5], double (&n4)[4]) {
#pragma clang fp contract(fast)
    double a[4][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { a[0][j] = nb[j].x; a[1][j] = nb[j].y; a[2][j] = nb[j].z; a[3][j] = 1.0; }
    double l[4][4];  // l[k][i] = R(k, i): column k of Rᵀ (zero above the diagonal)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double nrm2 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) nrm2 += a[k][i] * a[k][i];
        const double rn = nrm2 > 0.0 ? rsqrt_refined(nrm2) : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) l[k][i] = 0.0;
        l[k][k] = nrm2 * rn;
        double q[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) q[i] = a[k][i] * rn;
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) r += q[i] * a[j][i];
            l[k][j] = r;
#pragma unroll
            for (int i = 0; i < 5; ++i) a[j][i] -= r * q[i];
        }
    }
    double unused[4][4];
    double sn4[4];
    jacobi_svd_onesided<4, 4, false>(l, unused, sn4);
    int best = 0;
    double bn = 1e300, mx = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double sn = sn4[c];
        if (sn < bn) { bn = sn; best = c; }
        mx = sn > mx ? sn : mx;
    }
    // Rank-deficient neighbourhood (five collinear / coincident neighbours, exactly): fewer than three singular values above
    // 1e-13·σmax. The null space then has two or more dimensions and Eigen's choice inside it is an accident of rounding; the rule
    // used here ("rank-deficient neighbourhoods"; the CPU checker restates the same rule): the unit vector of the orthogonal complement of the dominant
    // right singular vectors closest to e4, else e3, e2, e1 (first whose projection keeps ≥ 0.4 of its squared length).
    const double thr = 1e-26 * mx;
    const int rank = (sn4[0] > thr ? 1 : 0) + (sn4[1] > thr ? 1 : 0) + (sn4[2] > thr ? 1 : 0) + (sn4[3] > thr ? 1 : 0);
    if (__builtin_expect(rank < 3, 0)) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        double nn = 0.0;
#pragma unroll 1
        for (int k = 3; k >= 0; --k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = (i == k) ? 1.0 : 0.0;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (sn4[c] > thr) {  // dominant column: l[c] = σ·v, so (t·v)v = (t·l) l / |l|²
                        const double d = ((t[0] * l[c][0] + t[1] * l[c][1]) + (t[2] * l[c][2] + t[3] * l[c][3])) / sn4[c];
#pragma unroll
                        for (int i = 0; i < 4; ++i) t[i] -= d * l[c][i];
                    }
                }
            }
            nn = (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
            if (nn >= 0.4) break;
        }
        const double rn = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
        n4[0] = t[0] * rn; n4[1] = t[1] * rn; n4[2] = t[2] * rn; n4[3] = t[3] * rn;
        return;
    }
    // the three other columns, by static selects (a dynamic register index would go through scratch)
    double u[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        u[0][r] = best == 0 ? l[1][r] : l[0][r];
        u[1][r] = best <= 1 ? l[2][r] : l[1][r];
        u[2][r] = best <= 2 ? l[3][r] : l[2][r];
    }
    // generalised cross product: component i = (−1)^i · det of the 3×3 matrix left after deleting coordinate i
    auto det3 = [&](int c0, int c1, int c2) {
        return u[0][c0] * (u[1][c1] * u[2][c2] - u[1][c2] * u[2][c1]) - u[0][c1] * (u[1][c0] * u[2][c2] - u[1][c2] * u[2][c0]) +
               u[0][c2] * (u[1][c0] * u[2][c1] - u[1][c1] * u[2][c0]);
    };
    const double w0 = det3(1, 2, 3), w1 = -det3(0, 2, 3), w2 = det3(0, 1, 3), w3 = -det3(0, 1, 2);
    const double wn = (w0 * w0 + w1 * w1) + (w2 * w2 + w3 * w3);
    const double rw = wn > 0.0 ? rsqrt_refined(wn) : 0.0;
    n4[0] = w0 * rw; n4[1] = w1 * rw; n4[2] = w2 * rw; n4[3] = w3 * rw;
}

// The same vector as plane_null_vector — the right singular vector of the smallest singular value of A = [x y z 1] — for about
// two thirds of its instructions (round 4; the fit kernel is bound by instruction issue and this is two thirds of it).
// With c = the centroid and Q = P − 1cᵀ the centred neighbours (Qᵀ1 = 0), A·(a, d) = Q·a + (c·a + d)·1, so
//     ‖A n‖² = aᵀSa + 5t²   with S = QᵀQ, t = c·a + d,   under   ‖a‖² + (t − c·a)² = 1:
// a generalised eigenproblem diag(S, 5)·x = μ·(diag(I, 0) + wwᵀ)·x, w = (c, −1), that needs only the 3×3 eigen-decomposition of S
// (three Jacobi pairs per sweep instead of six, on the triangular factor of the centred 5×3 matrix — small singular values to high
// relative accuracy, like the 4-column scheme) and the smallest root of its secular equation
//     1 = μ·( Σₖ gₖ²/(λₖ − μ) + 1/5 ),   gₖ = c·uₖ,   0 < μ < λ₁ = min λₖ.
// Multiplied by Πλₖ·Π(λₖ − μ) it is a polynomial Ψ(μ), symmetric in the three poles and free of divisions (a plane through the
// origin puts the root right below λ₁: Ψ has no pole there). Newton on Ψ from the root of the one-pole model (the two far poles
// frozen at μ = 0; hardware-precision 1/x and 1/√x are enough for a first guess) takes 2.3 iterations on average and about 4 for the
// worst lane of a wave; it stops when a step is below 1e-8·μ — quadratic convergence makes the iterate behind that step exact to
// rounding. Then a ∝ Σₖ gₖ/(λₖ − μ)·uₖ and d ∝ −1/5 − c·a/μ, again multiplied out, normalised at the end (the sign is free).
// Against a long-double Jacobi SVD on 65 536 synthetic neighbourhoods up to 500 m from the origin — planes through the origin,
// nearly collinear neighbourhoods and noise down to 20 nm included (tools/ubench/plane_fit_accuracy.hip) — |Δn4| ≤ 7e-13.
// Neighbourhoods whose centred matrix has (numerically) rank < 3 — five exactly coplanar, collinear or coincident neighbours — and
// anything that does not come out finite return false: the caller runs plane_null_vector, whose rules for those cases are the
// ones shared with the CPU checker.
__device__ __forceinline__ bool plane_null_vector_secular(const D3 (&nb)[5], double (&n4)[4]) {
#pragma clang fp contract(fast)
    const double cx = ((nb[0].x + nb[1].x) + (nb[2].x + nb[3].x) + nb[4].x) * 0.2, cy = ((nb[0].y + nb[1].y) + (nb[2].y + nb[3].y) + nb[4].y) * 0.2,
                 cz = ((nb[0].z + nb[1].z) + (nb[2].z + nb[3].z) + nb[4].z) * 0.2;
    double a[3][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { a[0][j] = nb[j].x - cx; a[1][j] = nb[j].y - cy; a[2][j] = nb[j].z - cz; }
    double l[3][3];  // l[k][i] = R(k, i) of Q = Q̂R: column k of Rᵀ
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double nrm2 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) nrm2 += a[k][i] * a[k][i];
        const double rn = nrm2 > 0.0 ? rsqrt_refined(nrm2) : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) l[k][i] = 0.0;
        l[k][k] = nrm2 * rn;
        double q[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) q[i] = a[k][i] * rn;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) r += q[i] * a[j][i];
            l[k][j] = r;
#pragma unroll
            for (int i = 0; i < 5; ++i) a[j][i] -= r * q[i];
        }
    }
    double unused[3][3];
    double sn[3];
    jacobi_svd_onesided<3, 3, false>(l, unused, sn);  // columns l[k] = σₖ·uₖ, sn[k] = σₖ² = λₖ
    const double mx = fmax(sn[0], fmax(sn[1], sn[2]));
    const bool b0 = sn[0] <= sn[1] && sn[0] <= sn[2], b1 = !b0 && sn[1] <= sn[2];
    const double l1 = b0 ? sn[0] : (b1 ? sn[1] : sn[2]);
    if (!(l1 > 1e-20 * mx)) return false;  // rank < 3 (also NaN): the 4-column scheme decides
    // everything below is scaled by λ₀λ₁λ₂: Gₖ = (c·lₖ)²·λⱼλₗ = gₖ²·Πλ (lₖ = σₖuₖ), S3 = Πλ — no division by a λ
    const double cl0 = (cx * l[0][0] + cy * l[0][1]) + cz * l[0][2], cl1 = (cx * l[1][0] + cy * l[1][1]) + cz * l[1][2],
                 cl2 = (cx * l[2][0] + cy * l[2][1]) + cz * l[2][2];
    const double s12 = sn[1] * sn[2], s02 = sn[0] * sn[2], s01 = sn[0] * sn[1];
    const double S3 = s01 * sn[2], S3_5 = 0.2 * S3;
    const double G0 = (cl0 * cl0) * s12, G1 = (cl1 * cl1) * s02, G2 = (cl2 * cl2) * s01;
    double mu;
    {   // first guess: R·μ² − (g₁² + R·λ₁ + 1)·μ + λ₁ = 0 with R = Σ_far gₖ²/λₖ + 1/5, smaller root in its stable form
        const double iS = __builtin_amdgcn_rcp(S3);
        const double i0 = __builtin_amdgcn_rcp(sn[0]), i1 = __builtin_amdgcn_rcp(sn[1]), i2 = __builtin_amdgcn_rcp(sn[2]);
        const double r0 = (G0 * iS) * i0, r1 = (G1 * iS) * i1, r2 = (G2 * iS) * i2;  // gₖ²/λₖ
        const double g1 = (b0 ? G0 : (b1 ? G1 : G2)) * iS;
        const double R = ((b0 ? 0.0 : r0) + (b1 ? 0.0 : r1)) + ((b0 || b1 ? r2 : 0.0) + 0.2);
        const double B = (g1 + R * l1) + 1.0;
        const double disc = fmax(B * B - 4.0 * (R * l1), 0.0);
        mu = (2.0 * l1) * __builtin_amdgcn_rcp(B + disc * __builtin_amdgcn_rsq(fmax(disc, 1e-300)));
        mu = fmin(mu, 0.999999 * l1);
    }
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
        const double e0 = sn[0] - mu, e1 = sn[1] - mu, e2 = sn[2] - mu;
        const double p01 = e0 * e1, p02 = e0 * e2, p12 = e1 * e2;
        const double E = p01 * e2;
        const double u = ((G0 * p12 + G1 * p02) + G2 * p01) + S3_5 * E;
        const double psi = mu * u - S3 * E;
        const double dE = -((p01 + p02) + p12);
        const double du = S3_5 * dE - ((G0 * (e1 + e2) + G1 * (e0 + e2)) + G2 * (e0 + e1));
        const double dpsi = (u + mu * du) - S3 * dE;
        double nu = mu - psi * __builtin_amdgcn_rcp(dpsi);
        if (!(nu > 0.0 && nu < l1)) nu = 0.5 * (mu + (psi < 0.0 ? l1 : 0.0));  // Newton left the interval: bisect towards the root's side
        const bool done = fabs(nu - mu) <= 1e-8 * nu;
        mu = nu;
        if (done) break;
    }
    const double e0 = sn[0] - mu, e1 = sn[1] - mu, e2 = sn[2] - mu;
    const double w0 = (cl0 * s12) * (e1 * e2), w1 = (cl1 * s02) * (e0 * e2), w2 = (cl2 * s01) * (e0 * e1);
    const double ax = (w0 * l[0][0] + w1 * l[1][0]) + w2 * l[2][0], ay = (w0 * l[0][1] + w1 * l[1][1]) + w2 * l[2][1],
                 az = (w0 * l[0][2] + w1 * l[1][2]) + w2 * l[2][2];
    const double d = -S3_5 * ((e0 * e1) * e2) - ((cx * ax + cy * ay) + cz * az);
    const double nn = (ax * ax + ay * ay) + (az * az + d * d);
    if (!(nn > 1e-290 && nn < 1e290)) return false;  // not finite, or scaled out of range: the 4-column scheme decides
    const double rn = rsqrt_refined(nn);
    n4[0] = ax * rn; n4[1] = ay * rn; n4[2] = az * rn; n4[3] = d * rn;
    return true;
}

// Right singular vector of the LARGEST singular value of the 5×3 matrix of centred neighbours — the line direction
// math::FitLine takes from JacobiSVD (math_utils.h:152-154: V.col(0)), up to its sign, which cancels in H and B. Same scheme
// as plane_null_vector: R from modified Gram–Schmidt, one-sided Jacobi on the columns of Rᵀ; the dominant right singular
// vector is the largest column of the rotated matrix, normalised (accurate precisely because its singular value is large).
__device__ __forceinline__ D3 line_direction(const D3 (&dl)[5]) {
#pragma clang fp contract(fast)
    double a[3][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { a[0][j] = dl[j].x; a[1][j] = dl[j].y; a[2][j] = dl[j].z; }
    double l[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double nrm2 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) nrm2 += a[k][i] * a[k][i];
        const double rn = nrm2 > 0.0 ? rsqrt_refined(nrm2) : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) l[k][i] = 0.0;
        l[k][k] = nrm2 * rn;
        double q[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) q[i] = a[k][i] * rn;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) r += q[i] * a[j][i];
            l[k][j] = r;
#pragma unroll
            for (int i = 0; i < 5; ++i) a[j][i] -= r * q[i];
        }
    }
    double unused[3][3];
    double sn3[3];
    jacobi_svd_onesided<3, 3, false>(l, unused, sn3);
    int best = 0;
    double bn = -1.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double sn = sn3[c];
        if (sn > bn) { bn = sn; best = c; }
    }
    const double rw = bn > 0.0 ? rsqrt_refined(bn) : 0.0;
    D3 d;
    d.x = (best == 0 ? l[0][0] : (best == 1 ? l[1][0] : l[2][0])) * rw;
    d.y = (best == 0 ? l[0][1] : (best == 1 ? l[1][1] : l[2][1])) * rw;
    d.z = (best == 0 ? l[0][2] : (best == 1 ? l[1][2] : l[2][2])) * rw;
    if (!(bn > 0.0)) d.x = 1.0;  // all five neighbours coincide: JacobiSVD of the zero matrix returns V = I, i.e. (1, 0, 0)
    return d;
}

// 6×6 partial-pivot LU: determinant and (if non-zero) solution of H x = b. What Eigen's fixed-size
// 6×6 determinant()/inverse() do (icp_registration.cpp:210,364). Runs in one thread.
// `work` (optional, 42 doubles): where the factorisation lives — on the device pass LDS (a local array indexed by the pivot search
// lives in scratch memory, ≈10× the latency).
__device__ __host__ inline double lu6_det_solve(const double* H, const double* b, double* x, double* work = nullptr) {
    double a_local[36];
    double* a = work ? work : a_local;
    for (int i = 0; i < 36; ++i) a[i] = H[i];
    int perm[6];
    for (int i = 0; i < 6; ++i) perm[i] = i;
    double det = 1.0;
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(a[6 * k + k]);
        for (int r = k + 1; r < 6; ++r) {
            const double val = fabs(a[6 * r + k]);
            if (val > best) { best = val; piv = r; }
        }
        if (piv != k) {
            for (int c = 0; c < 6; ++c) { const double tmp = a[6 * k + c]; a[6 * k + c] = a[6 * piv + c]; a[6 * piv + c] = tmp; }
            const int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
            det = -det;
        }
        const double d = a[6 * k + k];
        det *= d;
        if (d == 0.0) continue;
        for (int r = k + 1; r < 6; ++r) {
            const double f = a[6 * r + k] / d;
            a[6 * r + k] = f;
            for (int c = k + 1; c < 6; ++c) a[6 * r + c] -= f * a[6 * k + c];
        }
    }
    if (det == 0.0) return det;
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[perm[i]];
        for (int j = 0; j < i; ++j) s -= a[6 * i + j] * y[j];
        y[i] = s;
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int j = i + 1; j < 6; ++j) s -= a[6 * i + j] * x[j];
        x[i] = s / a[6 * i + i];
    }
    return det;
}

// The same factorisation, the same arithmetic in the same order, for ONE device thread with everything in registers: the loops are
// fully unrolled, the pivot row is found with compares and swapped in with selects (a row index that is only known at run time would
// put the matrix in scratch memory — ≈10× the latency — or, as rounds 2-4 had it, in LDS: ≈200 dependent LDS round trips, half of the
// solve kernel's 10 µs on the one-scan path). Bit-identical to lu6_det_solve (the host's locgpu_gn_update still uses that one).
__device__ __forceinline__ double lu6_det_solve_reg(const double* H, const double* b, double* x) {
    double a[6][6], rhs[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        rhs[r] = b[r];
#pragma unroll
        for (int c = 0; c < 6; ++c) a[r][c] = H[6 * r + c];
    }
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(a[k][k]);
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double val = fabs(a[r][k]);
            const bool g = val > best;
            best = g ? val : best;
            piv = g ? r : piv;
        }
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {  // swap rows k and piv (whole rows, like the array version; the right-hand side goes along instead of a permutation)
            const bool sel = piv == r;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double u = a[k][c], v = a[r][c];
                a[k][c] = sel ? v : u;
                a[r][c] = sel ? u : v;
            }
            const double u = rhs[k], v = rhs[r];
            rhs[k] = sel ? v : u;
            rhs[r] = sel ? u : v;
        }
        det = piv != k ? -det : det;
        const double d = a[k][k];
        det *= d;
        const bool nz = !(d == 0.0);  // `if (d == 0.0) continue;`
#pragma unroll
        for (int r = k + 1; r < 6; ++r) {
            const double f = a[r][k] / d;
            a[r][k] = nz ? f : a[r][k];
#pragma unroll
            for (int c = k + 1; c < 6; ++c) {
                const double t = a[r][c] - f * a[k][c];
                a[r][c] = nz ? t : a[r][c];
            }
        }
    }
    if (det == 0.0) return det;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = rhs[i];
#pragma unroll
        for (int j = 0; j < i; ++j) s -= a[i][j] * y[j];
        y[i] = s;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) s -= a[i][j] * x[j];
        x[i] = s / a[i][i];
    }
    return det;
}



// pose.so3() = pose.so3() * SO3::exp(dx.head<3>()); pose.translation() += dx.tail<3>()
// (icp_registration.cpp:365-366) — Sophus SO3::exp with its small-angle Taylor branch and the
// first-order renormalisation of the quaternion product.
__device__ __host__ inline void se3_apply_update(double* q, double* t, const double* dx) {
    // Sophus::SO3d::expAndTheta, the quaternion product and the renormalisation in the reference binary's own association
    // (libLocUtils.so 0x66550, 0x5b00d-0x5b10f): theta = sqrt((x² + y²) + z²), Taylor branch iff theta < 1e-10
    const double theta_sq = (dx[0] * dx[0] + dx[1] * dx[1]) + dx[2] * dx[2];
    const double theta = sqrt(theta_sq);
    double imag, real;
    if (theta < 1e-10) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double half = 0.5 * theta;
        imag = sin(half) / theta;
        real = cos(half);
    }
    const double bx = imag * dx[0], by = imag * dx[1], bz = imag * dx[2], bw = real;
    const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
    double rx = (ay * bz + aw * bx) - (az * by - ax * bw);
    double ry = (ay * bw + aw * by) + (az * bx - ax * bz);
    double rz = (aw * bz - ay * bx) + (ax * by + az * bw);
    double rw = (aw * bw - ay * by) - (ax * bx + az * bz);
    const double sq = (rz * rz + rx * rx) + (rw * rw + ry * ry);
    if (sq != 1.0) {
        const double scale = 2.0 / (sq + 1.0);
        rx *= scale; ry *= scale; rz *= scale; rw *= scale;
    }
    q[0] = rx; q[1] = ry; q[2] = rz; q[3] = rw;
    t[0] += dx[3]; t[1] += dx[4]; t[2] += dx[5];
}

// 64-lane butterfly sum of a double (two 32-bit DPP/permute moves per step).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace locgpu
