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

int main(int argc, char** argv) {
    if (argc == 2 && std::string(argv[1]) == "refusals") return run_refusals();
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
