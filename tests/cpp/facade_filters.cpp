// tests/cpp/facade_filters.cpp — drives the filter façade the way the front-ends do:
//   Loc::Update      loc.cpp:217-218   cloud = RemoveNanPoint(cloud); cur_scan_filter_ptr_->Filter(cloud, cloud);
//   Loc::ResetLocalMap loc.cpp:187-194 box_filter_ptr_->SetOrigin(origin); box_filter_ptr_->Filter(global, local);
// through std::shared_ptr<CloudFilterInterface>.
// Usage: facade_filters <in.bin> <dense 0|1> <leaf> <ox> <oy> <oz> <half> <out_prefix>
//   in.bin: float32 [n][4] (x y z intensity). Writes <prefix>.scan.bin (NaN removal + voxel filter, in place) and
//   <prefix>.box.bin (crop box around the origin), both float32 [m][4]; prints "m_scan m_box edge0..5 dense_scan dense_box".
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "LocUtils/model/cloud_filter/box_filter.hpp"
#include "LocUtils/model/cloud_filter/voxel_filter.hpp"
#include "LocUtils/model/feature_extract/loam_feature_extract.hpp"
#include "locgpu_facade/cloud_ops.hpp"

using namespace LocUtils;

static void save(const std::string& path, const CloudPtr& c) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) std::exit(2);
    for (const auto& p : c->points) {
        const float v[4] = {p.x, p.y, p.z, p.intensity};
        std::fwrite(v, 4, 4, f);
    }
    std::fclose(f);
}

// loam mode: facade_filters loam <in.bin float32 [n][4]> <ring.bin uint8 [n]> <num_scan> <out_prefix>   (Lio::AddCloud(FullCloudPtr), lio.cpp:321-323)
static int run_loam(char** argv) {
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> raw(bytes / 4);
    if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) return 2;
    std::fclose(f);
    const size_t n = raw.size() / 4;
    std::vector<unsigned char> ring(n);
    f = std::fopen(argv[3], "rb");
    if (!f || std::fread(ring.data(), 1, n, f) != n) return 2;
    std::fclose(f);
    FullCloudPtr scan(new FullPointCloudType);
    scan->points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        auto& p = scan->points[i];
        p.x = raw[4 * i]; p.y = raw[4 * i + 1]; p.z = raw[4 * i + 2];
        p.intensity = (unsigned char)raw[4 * i + 3];
        p.ring = ring[i];
    }
    LoamFeatureOptions opt;
    opt.num_scan_ = (size_t)std::atoi(argv[4]);
    std::shared_ptr<LoamFeatureExtract> loam_feature_ptr = std::make_shared<LoamFeatureExtract>(opt);  // lio.cpp:52
    CloudPtr edge_cloud(new PointCloudType), surf_cloud(new PointCloudType);
    loam_feature_ptr->Extract(scan, edge_cloud, surf_cloud);
    const std::string prefix = argv[5];
    save(prefix + ".edge.bin", edge_cloud);
    save(prefix + ".surf.bin", surf_cloud);
    std::printf("%zu %zu\n", edge_cloud->points.size(), surf_cloud->points.size());
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 6 && std::string(argv[1]) == "loam") return run_loam(argv);
    if (argc != 9) { std::fprintf(stderr, "usage\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> raw(bytes / 4);
    if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) return 2;
    std::fclose(f);
    CloudPtr cloud(new PointCloudType);
    cloud->points.resize(raw.size() / 4);
    for (size_t i = 0; i < cloud->points.size(); ++i) {
        cloud->points[i].x = raw[4 * i]; cloud->points[i].y = raw[4 * i + 1]; cloud->points[i].z = raw[4 * i + 2];
        cloud->points[i].intensity = raw[4 * i + 3];
    }
    cloud->is_dense = std::atoi(argv[2]) != 0;
    const float leaf = (float)std::atof(argv[3]);
    const std::vector<float> origin = {(float)std::atof(argv[4]), (float)std::atof(argv[5]), (float)std::atof(argv[6])};
    const float half = (float)std::atof(argv[7]);
    const std::string prefix = argv[8];

    std::shared_ptr<CloudFilterInterface> cur_scan_filter_ptr = std::make_shared<VoxelFilter>(leaf);   // loc.cpp:113
    std::shared_ptr<BoxFilter> box_filter_ptr = std::make_shared<BoxFilter>(half, half, half);           // loc.cpp:115

    // crop first, from the untouched input (global map → local map)
    CloudPtr local(new PointCloudType);
    box_filter_ptr->SetOrigin(origin);
    box_filter_ptr->Filter(cloud, local);
    const std::vector<float> edge = box_filter_ptr->GetEdge();

    CloudPtr no_nan = gpu::RemoveNanPoint(cloud);
    if (!cur_scan_filter_ptr->Filter(no_nan, no_nan)) return 3;  // in place, loc.cpp:218

    save(prefix + ".scan.bin", no_nan);
    save(prefix + ".box.bin", local);
    std::printf("%zu %zu %.9g %.9g %.9g %.9g %.9g %.9g %d %d\n", no_nan->points.size(), local->points.size(), edge[0], edge[1], edge[2], edge[3], edge[4],
                edge[5], no_nan->is_dense ? 1 : 0, local->is_dense ? 1 : 0);
    // filtering a cloud into itself through BoxFilter empties it (clear() precedes the read, box_filter.cpp:27)
    box_filter_ptr->Filter(local, local);
    if (!local->points.empty()) return 4;
    return 0;
}
