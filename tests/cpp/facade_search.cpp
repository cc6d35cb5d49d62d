// tests/cpp/facade_search.cpp — drives the search plug-ins of the façade the way the reference's matcher does
// (SearchPointInterface, search_point_interface.h:9-24): KdtreeRegistration (kdtree.h:134-156, kdtree.cpp:261-288) and
// BfnnRegistration (bfnn.h:11-37, bfnn.cpp:14-50) through a SearchPointInterface pointer.
//   facade_search <map.bin> <queries.bin> <k> <out.bin>
// map / queries: raw float32 [n][3]. out.bin: int32 [5][nq][k] — the lists of
//   0: KdtreeRegistration as constructed (ANN on, alpha 0.1: kdtree.h:128-129), FindNearstPoints query by query
//   1: after SetEnableANN(false) (exact: kdtree.cpp:285-288), query by query
//   2: after SetEnableANN(true, 0.3f), the batch entry point
//   3: BfnnRegistration, query by query
//   4: BfnnRegistration, the batch entry point
// (-1 where a list is shorter). Also checks what the reference pins: an empty cloud is refused by both (kdtree.cpp:263-266, bfnn.cpp:16-19),
// k larger than the tree gives an empty result (kdtree.cpp:149-153), FindCloud leaves its output alone.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "LocUtils/model/search_point/bfnn/bfnn.h"
#include "LocUtils/model/search_point/kdtree/kdtree.h"

using namespace LocUtils;

static std::vector<float> load(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> raw(bytes / 4);
    if (std::fread(raw.data(), 4, raw.size(), f) != raw.size()) std::exit(2);
    std::fclose(f);
    return raw;
}

int main(int argc, char** argv) {
    if (argc != 5) { std::fprintf(stderr, "usage: facade_search <map.bin> <queries.bin> <k> <out.bin>\n"); return 2; }
    const std::vector<float> m = load(argv[1]), q = load(argv[2]);
    const int k = std::atoi(argv[3]);
    const size_t nq = q.size() / 3;
    CloudPtr cloud(new PointCloudType);
    cloud->points.resize(m.size() / 3);
    for (size_t i = 0; i < cloud->points.size(); ++i) { cloud->points[i].x = m[3 * i]; cloud->points[i].y = m[3 * i + 1]; cloud->points[i].z = m[3 * i + 2]; }
    std::vector<int> out(5 * nq * k, -1);
    auto one_by_one = [&](SearchPointInterface& s, int slot) {
        for (size_t i = 0; i < nq; ++i) {
            Vec3f p;
            p.v[0] = q[3 * i]; p.v[1] = q[3 * i + 1]; p.v[2] = q[3 * i + 2];
            const std::vector<int> r = s.FindNearstPoints(p, k);
            if (r.size() > (size_t)k) std::exit(3);
            for (size_t j = 0; j < r.size(); ++j) out[((size_t)slot * nq + i) * k + j] = r[j];
        }
    };
    {
        std::shared_ptr<SearchPointInterface> kd = std::make_shared<KdtreeRegistration>();  // icp_registration.cpp:12: the matcher's search plug-in
        CloudPtr empty(new PointCloudType);
        if (kd->SetTargetCloud(empty)) return 4;   // kdtree.cpp:263-266
        if (!kd->SetTargetCloud(cloud)) return 4;
        one_by_one(*kd, 0);
        kd->SetEnableANN(false);
        one_by_one(*kd, 1);
        kd->SetEnableANN(true, 0.3f);
        std::vector<int> b;
        if (!static_cast<KdtreeRegistration&>(*kd).FindNearstPointsBatch(q.data(), nq, k, b) || b.size() != nq * k) return 5;
        std::copy(b.begin(), b.end(), out.begin() + 2 * nq * k);
        Vec3f p;
        if (!kd->FindNearstPoints(p, (int)cloud->points.size() + 1).empty()) return 6;  // k > size_: error, empty result (kdtree.cpp:149-153)
        std::vector<std::pair<size_t, size_t>> matches(3);
        kd->FindCloud(cloud, matches);
        if (matches.size() != 3) return 7;  // kdtree.cpp:290-293: an empty body
    }
    {
        std::shared_ptr<SearchPointInterface> bf = std::make_shared<BfnnRegistration>();
        CloudPtr empty(new PointCloudType);
        if (bf->SetTargetCloud(empty)) return 8;   // bfnn.cpp:16-19
        if (!bf->SetTargetCloud(cloud)) return 8;
        one_by_one(*bf, 3);
        std::vector<int> b;
        if (!static_cast<BfnnRegistration&>(*bf).FindNearstPointsBatch(q.data(), nq, k, b) || b.size() != nq * k) return 9;
        std::copy(b.begin(), b.end(), out.begin() + 4 * nq * k);
    }
    FILE* f = std::fopen(argv[4], "wb");
    if (!f || std::fwrite(out.data(), 4, out.size(), f) != out.size()) return 10;
    std::fclose(f);
    return 0;
}
