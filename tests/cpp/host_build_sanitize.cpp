// tests/cpp/host_build_sanitize.cpp — the host-side target ingest (packed KD-tree + exact-search grid builders) on random, duplicate-heavy, planar
// and tiny clouds, meant to be compiled with -fsanitize=address,undefined (tests/test_abi_and_host.py). GPU sanitizers do not exist on this pool.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "kdtree_build.hpp"
#include "grid_build.hpp"
int main() {
    std::mt19937 rng(7);
    for (int trial = 0; trial < 40; ++trial) {
        size_t n = trial < 5 ? (size_t)trial + 1 : (size_t)(rng() % 200000) + 1;
        std::vector<float> xyz(3 * n);
        std::normal_distribution<float> g(0.f, trial % 3 == 0 ? 0.01f : 20.f);
        for (auto& v : xyz) v = g(rng);
        if (trial % 4 == 1) for (size_t i = 0; i < n; i += 3) { xyz[3*i] = 1.f; xyz[3*i+1] = 2.f; xyz[3*i+2] = 3.f; }   // duplicates
        if (trial % 7 == 2) for (size_t i = 0; i < n; ++i) xyz[3*i+2] = 0.f;                                            // planar
        locgpu::PackedKdTree t; std::string err;
        if (!locgpu::build_packed_kdtree(xyz.data(), n, t, err)) { std::printf("build failed: %s\n", err.c_str()); return 1; }
        locgpu::SearchGrid gr;
        if (!locgpu::build_search_grid(t.slots.data(), t.slots.size(), gr, err)) { std::printf("grid failed: %s\n", err.c_str()); return 1; }
        if (gr.num_points != t.num_leaves) { std::printf("leaf mismatch\n"); return 1; }
    }
    std::puts("asan harness ok");
    return 0;
}
