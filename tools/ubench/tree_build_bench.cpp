// tools/ubench/tree_build_bench.cpp — host-only: how long does the packed KD-tree build of a keyframe local map take?
// (the streaming loop's matcher waits for it once per keyframe: tools/stream_trace.py)
//   g++ -O3 -std=c++17 -ffp-contract=off -I../../loc_lib_amd/csrc -o tree_build_bench tree_build_bench.cpp ../../loc_lib_amd/csrc/kdtree_build.cpp -pthread
//   ./tree_build_bench [points=35133] [reps=200] [verify=0]
// Prints the median / min build time and a checksum of the slots (the tree must not change with the thread count: LOCGPU_BUILD_THREADS).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kdtree_build.hpp"

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)std::atoll(argv[1]) : 35133;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 200;
    const bool verify = argc > 3 && std::atoi(argv[3]) != 0;  // every build's slots, leaf list, leaf count and depth against the first build's
    int bad = 0;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::normal_distribution<float> g(0.f, 0.02f);
    std::vector<float> xyz(3 * n);
    for (size_t i = 0; i < n; ++i) {  // ground + two walls, voxel-filter-like density
        const int kind = (int)(i % 4);
        float x = 40.f * u(rng), y = 40.f * u(rng), z = g(rng);
        if (kind == 2) { z = 3.f * (u(rng) + 1.f); y = 12.f + g(rng); }
        if (kind == 3) { z = 3.f * (u(rng) + 1.f); x = -9.f + g(rng); }
        xyz[3 * i] = x; xyz[3 * i + 1] = y; xyz[3 * i + 2] = z;
    }
    std::vector<double> t;
    uint64_t sum = 0;
    locgpu::PackedKdTree tree;  // reused, like the context's (its vectors keep their capacity)
    for (int r = 0; r < reps + 5; ++r) {
        std::string err;
        const auto a = std::chrono::steady_clock::now();
        if (!locgpu::build_packed_kdtree(xyz.data(), n, tree, err)) { std::fprintf(stderr, "build failed: %s\n", err.c_str()); return 1; }
        const auto b = std::chrono::steady_clock::now();
        if (r >= 5) t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
        uint64_t h = 0;
        if (r == 0 || verify) for (size_t i = 0; i < tree.slots.size(); ++i) h = h * 1099511628211ull + tree.slots[i];
        if (verify) for (size_t i = 0; i < tree.leaf_slots.size(); ++i) h = h * 1099511628211ull + tree.leaf_slots[i];
        if (verify) h = h * 1099511628211ull + (uint64_t)tree.num_leaves * 31u + (uint64_t)tree.depth;
        if (r == 0) sum = h;
        else if (verify && h != sum) { ++bad; std::fprintf(stderr, "build %d differs from the first: %016llx vs %016llx\n", r, (unsigned long long)h, (unsigned long long)sum); }
    }
    if (verify) std::printf("verify: %d of %d builds differ from the first\n", bad, reps + 4);
    std::sort(t.begin(), t.end());
    std::printf("%zu points, %d builds: median %.1f us, min %.1f us, p90 %.1f us; slots checksum %016llx\n", n, reps, t[t.size() / 2], t[0], t[t.size() * 9 / 10],
                (unsigned long long)sum);
    return 0;
}
