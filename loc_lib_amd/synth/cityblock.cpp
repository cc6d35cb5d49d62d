// loc_lib_amd/synth/cityblock.cpp
//
// Synthetic world "cityblock-v1" (SURVEY.md §8(d)): deterministic map and LiDAR scans regenerated
// from seeds on whichever machine runs the tests or the bench — no large fixtures travel.
// The reference ships no data (its PCDs sit behind a netdisk link, readme.md:23-27); the shapes
// mirror what its front-end feeds the matcher: a ±150 m local-map box (LocUtils/include/LocUtils/slam/3d/loc.hpp:35),
// 64-beam × 1800-azimuth sweeps with returns closer than 4 m dropped (subscriber/cloud_subscriber.cpp:13-18).
//
// Counter-based RNG (splitmix64 of (seed, stream, index, k)), so every point is independent of
// thread count and generation order. Host-only utility; not part of the hot path.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr double kHalf = 150.0;     // local-map half extent
constexpr double kWallH = 60.0;     // perimeter wall height
constexpr double kRing = 40.0;      // sensor circuit radius
constexpr double kSensorZ = 1.8;
constexpr int kNumBoxes = 200;
constexpr int kNumCyl = 300;
constexpr double kCylR = 0.15, kCylH = 6.0;
constexpr double kPi = 3.14159265358979323846;

inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t key(uint64_t seed, uint64_t stream, uint64_t index, uint64_t k) {
    return mix(mix(mix(mix(seed) ^ stream) ^ index) ^ k);
}
inline double uni(uint64_t seed, uint64_t stream, uint64_t index, uint64_t k) {  // [0,1)
    return (double)(key(seed, stream, index, k) >> 11) * (1.0 / 9007199254740992.0);
}
inline double gauss(uint64_t seed, uint64_t stream, uint64_t index, uint64_t k) {
    const double u1 = 1.0 - uni(seed, stream, index, 2 * k);  // (0,1]
    const double u2 = uni(seed, stream, index, 2 * k + 1);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * kPi * u2);
}

struct Box { double cx, cy, sx, sy, h, c, s; };  // c,s = cos/sin(yaw)
struct Cyl { double cx, cy; };

struct World {
    std::vector<Box> boxes;
    std::vector<Cyl> cyls;
    std::vector<double> cdf;  // cumulative surface areas
    // surface ids: 0 ground; 1..4 perimeter; then per box 5 faces; then cylinders
};

bool in_box_xy(const Box& b, double x, double y, double margin) {
    const double dx = x - b.cx, dy = y - b.cy;
    const double lx = b.c * dx + b.s * dy, ly = -b.s * dx + b.c * dy;
    return std::fabs(lx) <= 0.5 * b.sx + margin && std::fabs(ly) <= 0.5 * b.sy + margin;
}

World build_world(uint64_t seed) {
    World w;
    uint64_t attempt = 0;
    while ((int)w.boxes.size() < kNumBoxes) {
        const uint64_t a = attempt++;
        Box b;
        b.cx = (uni(seed, 1, a, 0) * 2 - 1) * 135.0;
        b.cy = (uni(seed, 1, a, 1) * 2 - 1) * 135.0;
        b.sx = 10.0 + 30.0 * uni(seed, 1, a, 2);
        b.sy = 10.0 + 30.0 * uni(seed, 1, a, 3);
        b.h = 5.0 + 25.0 * uni(seed, 1, a, 4);
        const double yaw = kPi * uni(seed, 1, a, 5);
        b.c = std::cos(yaw);
        b.s = std::sin(yaw);
        const double r = 0.5 * std::sqrt(b.sx * b.sx + b.sy * b.sy);
        const double rc = std::sqrt(b.cx * b.cx + b.cy * b.cy);
        if (std::fabs(rc - kRing) < r + 6.0) continue;                       // keep the 12 m corridor clear
        if (std::fabs(b.cx) + r > kHalf - 2 || std::fabs(b.cy) + r > kHalf - 2) continue;
        w.boxes.push_back(b);
    }
    attempt = 0;
    while ((int)w.cyls.size() < kNumCyl) {
        const uint64_t a = attempt++;
        Cyl c;
        c.cx = (uni(seed, 2, a, 0) * 2 - 1) * 140.0;
        c.cy = (uni(seed, 2, a, 1) * 2 - 1) * 140.0;
        const double rc = std::sqrt(c.cx * c.cx + c.cy * c.cy);
        if (std::fabs(rc - kRing) < 3.0) continue;
        bool inside = false;
        for (const Box& b : w.boxes) if (in_box_xy(b, c.cx, c.cy, 0.5)) { inside = true; break; }
        if (inside) continue;
        w.cyls.push_back(c);
    }
    double acc = 0;
    auto push = [&](double area) { acc += area; w.cdf.push_back(acc); };
    push(4 * kHalf * kHalf);                              // ground
    for (int i = 0; i < 4; ++i) push(2 * kHalf * kWallH);  // perimeter walls
    for (const Box& b : w.boxes) {
        push(b.sx * b.h); push(b.sx * b.h); push(b.sy * b.h); push(b.sy * b.h); push(b.sx * b.sy);
    }
    for (size_t i = 0; i < w.cyls.size(); ++i) push(2 * kPi * kCylR * kCylH);
    return w;
}

// One map sample: surface chosen area-weighted, uniform position, N(0, sigma) along the normal.
// With half > 0 only samples inside the square |x-cx|,|y-cy| <= half are kept (a box-cropped local map like
// Loc::ResetLocalMap, LocUtils/src/slam/3d/loc.cpp:187-194): same surface density as a full map of n/area points.
void sample_map_point(const World& w, uint64_t seed, uint64_t i, double sigma, float out[3], double cx = 0, double cy = 0, double half = -1) {
    for (uint64_t tr = 0;; ++tr) {
        const uint64_t st = 16 + tr;  // stream per retry
        const double pick = uni(seed, st, i, 0) * w.cdf.back();
        const size_t sid = std::upper_bound(w.cdf.begin(), w.cdf.end(), pick) - w.cdf.begin();
        const double u = uni(seed, st, i, 1), v = uni(seed, st, i, 2);
        const double nz = sigma * gauss(seed, st, i, 2);
        double x, y, z;
        if (sid == 0) {
            x = (u * 2 - 1) * kHalf; y = (v * 2 - 1) * kHalf; z = nz;
            bool covered = false;
            for (const Box& b : w.boxes) if (in_box_xy(b, x, y, 0.0)) { covered = true; break; }
            if (covered) continue;  // no ground under buildings: redraw
        } else if (sid <= 4) {
            const double a = (u * 2 - 1) * kHalf, hgt = v * kWallH;
            switch (sid) {
                case 1: x = kHalf + nz; y = a; break;
                case 2: x = -kHalf + nz; y = a; break;
                case 3: x = a; y = kHalf + nz; break;
                default: x = a; y = -kHalf + nz; break;
            }
            z = hgt;
        } else if (sid < 5 + 5 * w.boxes.size()) {
            const size_t bi = (sid - 5) / 5, face = (sid - 5) % 5;
            const Box& b = w.boxes[bi];
            double lx, ly;
            if (face == 0) { lx = (u - 0.5) * b.sx; ly = 0.5 * b.sy + nz; z = v * b.h; }
            else if (face == 1) { lx = (u - 0.5) * b.sx; ly = -0.5 * b.sy + nz; z = v * b.h; }
            else if (face == 2) { lx = 0.5 * b.sx + nz; ly = (u - 0.5) * b.sy; z = v * b.h; }
            else if (face == 3) { lx = -0.5 * b.sx + nz; ly = (u - 0.5) * b.sy; z = v * b.h; }
            else { lx = (u - 0.5) * b.sx; ly = (v - 0.5) * b.sy; z = b.h + nz; }
            x = b.cx + b.c * lx - b.s * ly;
            y = b.cy + b.s * lx + b.c * ly;
        } else {
            const Cyl& c = w.cyls[sid - 5 - 5 * w.boxes.size()];
            const double ang = 2 * kPi * u, r = kCylR + nz;
            x = c.cx + r * std::cos(ang); y = c.cy + r * std::sin(ang); z = v * kCylH;
        }
        if (half > 0 && (std::fabs(x - cx) > half || std::fabs(y - cy) > half)) continue;
        out[0] = (float)x; out[1] = (float)y; out[2] = (float)z;
        return;
    }
}

// Nearest hit distance along (o + t d), t > 0. Always finite thanks to ground + perimeter.
double raycast(const World& w, const double o[3], const double d[3]) {
    double best = 1e30;
    if (d[2] < 0) { const double t = -o[2] / d[2]; if (t > 0 && t < best) best = t; }
    // perimeter
    for (int ax = 0; ax < 2; ++ax) {
        if (d[ax] > 1e-12) { const double t = (kHalf - o[ax]) / d[ax]; if (t > 0 && t < best) { const double z = o[2] + t * d[2]; if (z >= 0 && z <= kWallH) best = t; } }
        if (d[ax] < -1e-12) { const double t = (-kHalf - o[ax]) / d[ax]; if (t > 0 && t < best) { const double z = o[2] + t * d[2]; if (z >= 0 && z <= kWallH) best = t; } }
    }
    for (const Box& b : w.boxes) {
        const double ox = o[0] - b.cx, oy = o[1] - b.cy;
        const double lo[3] = {b.c * ox + b.s * oy, -b.s * ox + b.c * oy, o[2]};
        const double ld[3] = {b.c * d[0] + b.s * d[1], -b.s * d[0] + b.c * d[1], d[2]};
        const double mn[3] = {-0.5 * b.sx, -0.5 * b.sy, 0.0}, mx[3] = {0.5 * b.sx, 0.5 * b.sy, b.h};
        double t0 = 0.0, t1 = best;
        bool hit = true;
        for (int a = 0; a < 3 && hit; ++a) {
            if (std::fabs(ld[a]) < 1e-12) { if (lo[a] < mn[a] || lo[a] > mx[a]) hit = false; continue; }
            double ta = (mn[a] - lo[a]) / ld[a], tb = (mx[a] - lo[a]) / ld[a];
            if (ta > tb) std::swap(ta, tb);
            if (ta > t0) t0 = ta;
            if (tb < t1) t1 = tb;
            if (t0 > t1) hit = false;
        }
        if (hit && t0 > 0 && t0 < best) best = t0;
    }
    for (const Cyl& c : w.cyls) {
        const double ox = o[0] - c.cx, oy = o[1] - c.cy;
        const double A = d[0] * d[0] + d[1] * d[1];
        if (A < 1e-14) continue;
        const double Bq = ox * d[0] + oy * d[1];
        const double C = ox * ox + oy * oy - kCylR * kCylR;
        const double disc = Bq * Bq - A * C;
        if (disc < 0) continue;
        const double t = (-Bq - std::sqrt(disc)) / A;
        if (t > 0 && t < best) { const double z = o[2] + t * d[2]; if (z >= 0 && z <= kCylH) best = t; }
    }
    return best;
}

void sensor_pose(int scan_id, double pos[3], double* yaw) {
    const double th = 2 * kPi * (double)scan_id / 256.0;
    pos[0] = kRing * std::cos(th); pos[1] = kRing * std::sin(th); pos[2] = kSensorZ;
    *yaw = th + 0.5 * kPi;
}

template <typename F>
void parallel_for(size_t n, F&& f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if (n < 4096) { f(0, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=, &f] { f(a, b); });
    }
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// Map: n area-weighted surface samples, N(0, sigma) along the normal, f32 xyz at `stride_floats`.
void cityblock_map(uint64_t seed, size_t n, double sigma, float* out, size_t stride_floats) {
    const World w = build_world(seed);
    parallel_for(n, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) sample_map_point(w, seed, i, sigma, out + i * stride_floats);
    });
}

// Local map: n samples restricted to the square of half-extent `half` around (cx, cy).
void cityblock_map_local(uint64_t seed, size_t n, double sigma, double cx, double cy, double half, float* out, size_t stride_floats) {
    const World w = build_world(seed);
    parallel_for(n, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) sample_map_point(w, seed, i, sigma, out + i * stride_floats, cx, cy, half);
    });
}

// Position of the sensor for `scan_id` (x, y, z, yaw).
void cityblock_sensor(int scan_id, double out[4]) {
    double pos[3], yaw;
    sensor_pose(scan_id, pos, &yaw);
    out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2]; out[3] = yaw;
}

// Scan `scan_id` of the 256-pose circuit, ring-major (beam outer, azimuth inner), sensor frame, f32.
// Returns the number of points written (rays with range < 4 m are dropped like cloud_subscriber.cpp:15).
size_t cityblock_scan(uint64_t world_seed, int scan_id, uint64_t noise_seed, int n_beams, int n_az, double range_sigma, float* out,
                      size_t stride_floats) {
    const World w = build_world(world_seed);
    double pos[3], yaw;
    sensor_pose(scan_id, pos, &yaw);
    const double cy = std::cos(yaw), sy = std::sin(yaw);
    const size_t total = (size_t)n_beams * n_az;
    std::vector<float> tmp(total * 3);
    std::vector<uint8_t> keep(total);
    parallel_for(total, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            const int beam = (int)(i / n_az), az = (int)(i % n_az);
            const double el = (2.0 - (n_beams > 1 ? 26.8 * beam / (n_beams - 1) : 0.0)) * kPi / 180.0;  // +2.0 … −24.8 deg
            const double azr = 2 * kPi * az / n_az;
            const double ds[3] = {std::cos(el) * std::cos(azr), std::cos(el) * std::sin(azr), std::sin(el)};
            const double dw[3] = {cy * ds[0] - sy * ds[1], sy * ds[0] + cy * ds[1], ds[2]};
            double r = raycast(w, pos, dw);
            r += range_sigma * gauss(noise_seed, 3, i, 0);
            keep[i] = r >= 4.0;
            tmp[3 * i] = (float)(r * ds[0]); tmp[3 * i + 1] = (float)(r * ds[1]); tmp[3 * i + 2] = (float)(r * ds[2]);
        }
    });
    size_t m = 0;
    for (size_t i = 0; i < total; ++i)
        if (keep[i]) { std::memcpy(out + m * stride_floats, &tmp[3 * i], 3 * sizeof(float)); ++m; }
    return m;
}

// true7 / init7: quaternion (x,y,z,w) + translation — Sophus::SE3d::data() order.
// init = T_true ∘ δ, δt ~ U(±trans_amp)^3, δrot = exp(U(±rot_amp_rad)^3).
void cityblock_pose(int scan_id, uint64_t perturb_seed, double trans_amp, double rot_amp_rad, double true7[7], double init7[7]) {
    double pos[3], yaw;
    sensor_pose(scan_id, pos, &yaw);
    const double qz = std::sin(0.5 * yaw), qw = std::cos(0.5 * yaw);
    true7[0] = 0; true7[1] = 0; true7[2] = qz; true7[3] = qw; true7[4] = pos[0]; true7[5] = pos[1]; true7[6] = pos[2];
    double dt[3], dr[3];
    for (int i = 0; i < 3; ++i) {
        dt[i] = (uni(perturb_seed, 5, scan_id, i) * 2 - 1) * trans_amp;
        dr[i] = (uni(perturb_seed, 5, scan_id, 3 + i) * 2 - 1) * rot_amp_rad;
    }
    const double th = std::sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
    double bx = 0, by = 0, bz = 0, bw = 1;
    if (th > 1e-12) { const double s = std::sin(0.5 * th) / th; bx = s * dr[0]; by = s * dr[1]; bz = s * dr[2]; bw = std::cos(0.5 * th); }
    const double ax = 0, ay = 0, az = qz, aw = qw;
    init7[3] = aw * bw - ax * bx - ay * by - az * bz;
    init7[0] = aw * bx + ax * bw + ay * bz - az * by;
    init7[1] = aw * by + ay * bw + az * bx - ax * bz;
    init7[2] = aw * bz + az * bw + ax * by - ay * bx;
    const double c = std::cos(yaw), s = std::sin(yaw);
    init7[4] = pos[0] + c * dt[0] - s * dt[1];
    init7[5] = pos[1] + s * dt[0] + c * dt[1];
    init7[6] = pos[2] + dt[2];
}

}  // extern "C"
