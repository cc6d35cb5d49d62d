// tests/cpp/stream_pipeline.cpp — BASELINE.json configs[4] driven from C++ through the C ABI, the way a slam_demo front-end would:
// per scan upload → voxel filter (VoxelGrid skips the non-finite points of a non-dense cloud itself: = RemoveNanPoint + Filter,
// lio.cpp:236) → P2Plane ScanMatch against the keyframe local map; every kf_every-th scan transform + submap update +
// SetInputTarget with the host tree build on a worker thread (Lio::AddCloud, lio.cpp:236-306).
//   mode 0: one thread, one context — the reference's sequential loop.
//   mode 1: two-stage front-end — a second thread uploads and filters scan i+1 on its own context while this one matches scan i
//           (include/locgpu.h, "Two contexts on one GPU"): same poses, bit for bit.
// Usage: stream_pipeline <scans.bin> <poses.bin> <n_scans> <pts_per_scan> <kf_every> <num_kfs> <leaf> <mode> <passes> <out_poses.bin>
//   scans.bin: n_scans x pts x 4 float32 {x, y, z, intensity}; poses.bin: n_scans x 14 float64 {truth7, init7}.
// Prints one JSON line: scans per second of every pass (wall time of the whole loop; the files are loaded beforehand).
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "locgpu.h"

#define CHECK(call) do { const int rc_ = (call); if (rc_ != LOCGPU_OK) { std::fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, locgpu_last_error(nullptr)); std::exit(3); } } while (0)

struct Pair { locgpu_cloud* raw; locgpu_cloud* filt; int scan; };

struct Channel {  // bounded hand-over between the two stages
    std::mutex m;
    std::condition_variable cv;
    std::deque<Pair> q;
    void put(const Pair& p) { { std::lock_guard<std::mutex> l(m); q.push_back(p); } cv.notify_one(); }
    Pair get() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty(); }); Pair p = q.front(); q.pop_front(); return p; }
};

int main(int argc, char** argv) {
    if (argc != 11) { std::fprintf(stderr, "usage: see the header of stream_pipeline.cpp\n"); return 2; }
    const int n_scans = std::atoi(argv[3]), kf_every = std::atoi(argv[5]), num_kfs = std::atoi(argv[6]), mode = std::atoi(argv[8]), passes = std::atoi(argv[9]);
    const size_t pts = (size_t)std::atoll(argv[4]);
    const float leaf = (float)std::atof(argv[7]);
    std::vector<float> scans((size_t)n_scans * pts * 4);
    std::vector<double> poses((size_t)n_scans * 14);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(scans.data(), 4, scans.size(), f) != scans.size()) { std::perror(argv[1]); return 2; }
    std::fclose(f);
    f = std::fopen(argv[2], "rb");
    if (!f || std::fread(poses.data(), 8, poses.size(), f) != poses.size()) { std::perror(argv[2]); return 2; }
    std::fclose(f);
    locgpu_icp_opts opts;
    locgpu_icp_opts_default(&opts);
    opts.method = LOCGPU_P2PLANE;
    std::vector<double> out((size_t)n_scans * 7);
    std::vector<double> rates;
    locgpu_ctx *ctx_m = nullptr, *ctx_f = nullptr;  // the matcher instances live as long as the front-end does
    CHECK(locgpu_create(0, &ctx_m));
    if (mode == 1) CHECK(locgpu_create(0, &ctx_f)); else ctx_f = ctx_m;
    const int n_pairs = mode == 1 ? 3 : 1;
    std::vector<Pair> pairs(n_pairs);
    for (auto& p : pairs) { CHECK(locgpu_cloud_create(ctx_f, &p.raw)); CHECK(locgpu_cloud_create(ctx_f, &p.filt)); p.scan = -1; }
    for (int pass = 0; pass <= passes; ++pass) {  // pass 0 is untimed: the library's buffers grow there
        locgpu_submap* sub = nullptr;
        CHECK(locgpu_submap_create(ctx_m, num_kfs, leaf, &sub));
        Channel free_ch, ready_ch;
        for (auto& p : pairs) free_ch.put(p);
        double t_filter = 0.0, t_match = 0.0, t_kf = 0.0, t_wait = 0.0;  // seconds inside the calls of each stage (STREAM_PIPELINE_TIMES=1 prints them)
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        auto filter_one = [&](Pair p, int s) {
            const auto ta = now();
            CHECK(locgpu_cloud_upload(p.raw, scans.data() + (size_t)s * pts * 4, pts, 16, 12, 0));
            CHECK(locgpu_cloud_voxel_filter(p.raw, leaf, p.filt, nullptr));
            t_filter += secs(ta, now());
            p.scan = s;
            return p;
        };
        const auto t0 = std::chrono::steady_clock::now();
        std::thread stage;
        if (mode == 1) stage = std::thread([&] { for (int s = 0; s < n_scans; ++s) ready_ch.put(filter_one(free_ch.get(), s)); });
        for (int s = 0; s < n_scans; ++s) {
            const auto tw = now();
            const Pair p = mode == 1 ? ready_ch.get() : filter_one(free_ch.get(), s);
            if (mode == 1) t_wait += secs(tw, now());
            const double* truth = &poses[(size_t)s * 14];
            const double* init = truth + 7;
            double* pose = &out[(size_t)s * 7];
            const locgpu_cloud* kf_src = p.raw;  // later keyframes keep the RAW scan (lio.cpp:279)
            const auto tm = now();
            if (s == 0) { std::memcpy(pose, truth, 56); kf_src = p.filt; }  // the first frame seeds the map with the FILTERED scan (lio.cpp:238-256)
            else CHECK(locgpu_icp_align_cloud(ctx_m, p.filt, init, &opts, pose, nullptr));
            const auto tk = now();
            t_match += secs(tm, tk);
            if (s % kf_every == 0) {
                CHECK(locgpu_submap_add_keyframe(sub, kf_src, pose));
                locgpu_cloud* map = nullptr;
                CHECK(locgpu_submap_cloud(sub, &map));
                CHECK(locgpu_icp_set_target_cloud_async(ctx_m, map));
                t_kf += secs(tk, now());
            }
            free_ch.put(p);
        }
        if (mode == 1) stage.join();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (pass > 0) rates.push_back(n_scans / wall);
        if (pass == passes && std::getenv("STREAM_PIPELINE_TIMES"))
            std::fprintf(stderr, "per scan [ms]: wall %.3f | upload + filter %.3f | match %.3f | keyframe + target %.3f | matcher thread waiting for a filtered scan %.3f\n",
                         1e3 * wall / n_scans, 1e3 * t_filter / n_scans, 1e3 * t_match / n_scans, 1e3 * t_kf / n_scans, 1e3 * t_wait / n_scans);
        locgpu_submap_destroy(sub);
    }
    for (auto& p : pairs) { locgpu_cloud_destroy(p.raw); locgpu_cloud_destroy(p.filt); }
    if (mode == 1) locgpu_destroy(ctx_f);
    locgpu_destroy(ctx_m);
    f = std::fopen(argv[10], "wb");
    if (!f || std::fwrite(out.data(), 8, out.size(), f) != out.size()) { std::perror(argv[10]); return 2; }
    std::fclose(f);
    std::printf("{\"mode\": \"%s\", \"scans\": %d, \"scans_per_s_all_passes\": [", mode == 1 ? "two_stage_pipeline" : "sequential", n_scans);
    for (size_t i = 0; i < rates.size(); ++i) std::printf("%s%.1f", i ? ", " : "", rates[i]);
    std::printf("]}\n");
    return 0;
}
