// tests/cpp/facade_scanmatch.cpp — drives the C++ façade exactly the way Loc::Update does (loc.cpp:194,229):
// SetInputTarget(map) once, then ScanMatch(scan, predict, out_cloud, out_pose) through a MatchingInterface pointer.
// Usage: facade_scanmatch <icp|ndt> <method 0..2> <map.bin> <scan.bin> <pose7.bin> <out.bin>
// Cloud files: raw float32 [n][3]; out.bin: 7 doubles (pose) then n*3 float32 (transformed cloud xyz).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "LocUtils/model/matching/3d/icp/icp_registration.hpp"
#include "LocUtils/model/matching/3d/loam/loam_registration.hpp"
#include "LocUtils/model/matching/3d/ndt/ndt_registration.hpp"

using namespace LocUtils;

static CloudPtr load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> raw(bytes / 4);
    if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) std::exit(2);
    std::fclose(f);
    CloudPtr c(new PointCloudType);
    c->points.resize(raw.size() / 3);
    for (size_t i = 0; i < c->points.size(); ++i) {
        c->points[i].x = raw[3 * i]; c->points[i].y = raw[3 * i + 1]; c->points[i].z = raw[3 * i + 2];
        c->points[i].intensity = (float)i;
    }
    return c;
}

// loam mode: facade_scanmatch loam <edge_map.bin> <surf_map.bin> <edge_scan.bin> <surf_scan.bin> <pose7.bin> <out.bin>
static int run_loam(char** argv) {
    CloudPtr edge_map = load(argv[2]), surf_map = load(argv[3]), edge = load(argv[4]), surf = load(argv[5]);
    SE3 predict, result;
    FILE* f = std::fopen(argv[6], "rb");
    if (!f || std::fread(predict.data(), 8, 7, f) != 7) return 2;
    std::fclose(f);
    std::shared_ptr<MatchingInterface> match_ptr = std::make_shared<LoamRegistration>(LoamOption());  // lio.cpp:51 / loc.cpp:71
    match_ptr->SetInputTarget(edge_map, surf_map);
    CloudPtr out(new PointCloudType);
    if (!match_ptr->ScanMatch(edge, surf, predict, out, result)) return 3;
    if (out->points.size() != edge->points.size() + surf->points.size()) return 4;
    f = std::fopen(argv[7], "wb");
    std::fwrite(result.data(), 8, 7, f);
    for (const auto& p : out->points) std::fwrite(&p.x, 4, 3, f);
    std::fclose(f);
    return 0;
}

// refusals mode (no GPU needed: a refusal comes before any context exists): option branches of the reference that are not on the GPU
// path make SetInputTarget / ScanMatch return false with a text in LastError() and leave the outputs alone — they are not quietly
// served by something else. facade_scanmatch refusals
static int run_refusals() {
    CloudPtr cloud(new PointCloudType);
    cloud->points.resize(8);
    CloudPtr out(new PointCloudType);
    out->points.resize(3);
    int bad = 0;
    auto check = [&](MatchingInterface& m, const char* err, const char* what) {
        SE3 predict, result;
        for (int i = 0; i < 7; ++i) { predict.data()[i] = 0.125 * (i + 1); result.data()[i] = -1.0 - i; }
        const bool a = m.SetInputTarget(cloud);
        const bool b = m.ScanMatch(cloud, predict, out, result);
        bool untouched = out->points.size() == 3;
        for (int i = 0; i < 7; ++i) untouched = untouched && result.data()[i] == -1.0 - i;
        const bool ok = !a && !b && untouched && err && std::strstr(err, what);
        std::printf("%s: SetInputTarget %d ScanMatch %d outputs untouched %d LastError \"%s\"\n", what, (int)a, (int)b, (int)untouched, err ? err : "");
        bad += ok ? 0 : 1;
    };
    { IcpOptions o(IcpMethod::PCLICP); IcpRegistration m(o); check(m, m.LastError(), "PCLICP"); }
    { IcpOptions o(IcpMethod::P2PLANE); o.use_initial_translation_ = false; IcpRegistration m(o); check(m, m.LastError(), "use_initial_translation_"); }
    { NdtOptions o; o.method_ = NdtMethod::PCL_NDT; NdtRegistration m(o); check(m, m.LastError(), "PCL_NDT"); }
    { NdtOptions o; o.remove_centroid_ = true; NdtRegistration m(o); check(m, m.LastError(), "remove_centroid_"); }
    return bad ? 5 : 0;
}

// time mode: the call slam_demo makes — match_ptr_->ScanMatch(host cloud, predict, fresh output cloud, pose)
// (loc.cpp:215,229) — timed end to end beside the C ABI's host-pointer alignment (no output cloud) on a second context.
//   facade_scanmatch time <icp|ndt> <method 0..2> <map.bin> <scans.bin> <n_scans> <poses7.bin> <reps>
// scans.bin: n_scans clouds of equal size back to back; poses7.bin: n_scans x 7 doubles. Prints one JSON line (ms per call).
#include <algorithm>
#include <chrono>
#include "../../include/locgpu.h"
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }
static int run_time(char** argv) {
    const std::string kind = argv[2];
    const int method = std::atoi(argv[3]);
    CloudPtr map = load(argv[4]), all = load(argv[5]);
    const int n_scans = std::atoi(argv[6]), reps = std::atoi(argv[8]);
    const size_t per = all->points.size() / (size_t)n_scans;
    std::vector<CloudPtr> scans;
    for (int s = 0; s < n_scans; ++s) {
        CloudPtr c(new PointCloudType);
        c->points.assign(all->points.begin() + s * per, all->points.begin() + (s + 1) * per);
        scans.push_back(c);
    }
    std::vector<SE3> predict(n_scans);
    {
        FILE* f = std::fopen(argv[7], "rb");
        if (!f) return 2;
        for (int s = 0; s < n_scans; ++s) if (std::fread(predict[s].data(), 8, 7, f) != 7) return 2;
        std::fclose(f);
    }
    std::shared_ptr<MatchingInterface> match_ptr;
    locgpu_ctx* ctx = nullptr;
    if (locgpu_create(0, &ctx) != LOCGPU_OK) return 3;
    locgpu_icp_opts io;
    locgpu_icp_opts_default(&io);
    io.method = method;
    if (kind == "icp") {
        IcpOptions o(method == 0 ? IcpMethod::P2P : (method == 1 ? IcpMethod::P2LINE : IcpMethod::P2PLANE));
        match_ptr = std::make_shared<IcpRegistration>(o);
        if (locgpu_icp_set_target(ctx, map->points.data(), map->points.size(), sizeof(PointType)) != LOCGPU_OK) return 3;
    } else {
        match_ptr = std::make_shared<NdtRegistration>(NdtOptions());
        locgpu_ndt_opts no;
        locgpu_ndt_opts_default(&no);
        if (locgpu_ndt_set_target(ctx, map->points.data(), map->points.size(), sizeof(PointType), &no) != LOCGPU_OK) return 3;
    }
    match_ptr->SetInputTarget(map);
    using clk = std::chrono::steady_clock;
    std::vector<double> t_facade, t_abi, t_abi_cloud;
    std::vector<SE3> res_f(n_scans), res_a(n_scans);
    for (int r = -1; r < reps; ++r) {  // r = -1: warm-up (buffers, first-call setup)
        for (int s = 0; s < n_scans; ++s) {
            const auto t0 = clk::now();
            CloudPtr out(new PointCloudType);  // loc.cpp:215: a fresh output cloud per scan
            match_ptr->ScanMatch(scans[s], predict[s], out, res_f[s]);
            const auto t1 = clk::now();
            if (out->points.size() != per || out->points[per - 1].intensity != scans[s]->points[per - 1].intensity) return 4;
            double pose[7];
            const auto t2 = clk::now();
            const int rc = kind == "icp" ? locgpu_icp_align(ctx, scans[s]->points.data(), per, sizeof(PointType), predict[s].data(), &io, pose, nullptr)
                                         : locgpu_ndt_align(ctx, scans[s]->points.data(), per, sizeof(PointType), predict[s].data(), pose, nullptr);
            const auto t3 = clk::now();
            if (rc != LOCGPU_OK) return 5;
            std::memcpy(res_a[s].data(), pose, sizeof(pose));
            if (r >= 0) {
                t_facade.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
                t_abi.push_back(std::chrono::duration<double, std::milli>(t3 - t2).count());
            }
        }
    }
    for (int s = 0; s < n_scans; ++s)
        if (std::memcmp(res_f[s].data(), res_a[s].data(), 56) != 0) return 6;  // the façade's pose is the C ABI's, bit for bit
    double mf = 0, ma = 0;
    for (double v : t_facade) mf += v;
    for (double v : t_abi) ma += v;
    std::printf("{\"kind\": \"%s\", \"method\": %d, \"points_per_scan\": %zu, \"map_points\": %zu, \"calls\": %zu, "
                "\"facade_scanmatch_ms_median\": %.4f, \"facade_scanmatch_ms_mean\": %.4f, \"abi_align_host_pointer_ms_median\": %.4f, \"abi_align_host_pointer_ms_mean\": %.4f}\n",
                kind.c_str(), method, per, map->points.size(), t_facade.size(), median(t_facade), mf / t_facade.size(), median(t_abi), ma / t_abi.size());
    locgpu_destroy(ctx);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 2 && std::string(argv[1]) == "refusals") return run_refusals();
    if (argc == 9 && std::string(argv[1]) == "time") return run_time(argv);
    if (argc == 8 && std::string(argv[1]) == "loam") return run_loam(argv);
    if (argc != 7) { std::fprintf(stderr, "usage\n"); return 2; }
    const std::string kind = argv[1];
    const int method = std::atoi(argv[2]);
    CloudPtr map = load(argv[3]), scan = load(argv[4]);
    SE3 predict, result;
    {
        FILE* f = std::fopen(argv[5], "rb");
        if (!f || std::fread(predict.data(), 8, 7, f) != 7) return 2;
        std::fclose(f);
    }
    std::shared_ptr<MatchingInterface> match_ptr;  // what Loc / Lio hold (loc.hpp:85)
    if (kind == "icp") {
        IcpOptions o(method == 0 ? IcpMethod::P2P : (method == 1 ? IcpMethod::P2LINE : IcpMethod::P2PLANE));
        match_ptr = std::make_shared<IcpRegistration>(o);
    } else {
        NdtOptions o;
        match_ptr = std::make_shared<NdtRegistration>(o);
    }
    match_ptr->SetInputTarget(map);
    map->points.clear();  // the matcher deep-copied it: callers may drop their cloud (lio.cpp:297 mutates it in place)
    CloudPtr out(new PointCloudType);
    const bool ok = match_ptr->ScanMatch(scan, predict, out, result);
    if (!ok || out->points.size() != scan->points.size()) return 3;
    for (size_t i = 0; i < out->points.size(); ++i)
        if (out->points[i].intensity != (float)i) return 4;  // other fields survive the transform
    FILE* f = std::fopen(argv[6], "wb");
    std::fwrite(result.data(), 8, 7, f);
    for (const auto& p : out->points) std::fwrite(&p.x, 4, 3, f);
    std::fclose(f);
    std::printf("pose %.12f %.12f %.12f %.12f %.9f %.9f %.9f\n", result.data()[0], result.data()[1], result.data()[2], result.data()[3],
                result.data()[4], result.data()[5], result.data()[6]);
    return 0;
}
