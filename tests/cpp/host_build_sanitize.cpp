// tests/cpp/host_build_sanitize.cpp — the host-side target ingest (packed KD-tree builder and its leaf list) on random, duplicate-heavy, planar
// and tiny clouds, meant to be compiled with -fsanitize=address,undefined (tests/test_abi_and_host.py). GPU sanitizers do not exist on this pool.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "kdtree_build.hpp"
int main() {
    std::mt19937 rng(7);
    uint64_t sum = 1469598103934665603ull;  // FNV-1a over every tree's slots: two builds of the builder (SSE / lane-by-lane) must print the same
    for (int trial = 0; trial < 40; ++trial) {
        size_t n = trial < 5 ? (size_t)trial + 1 : (size_t)(rng() % 200000) + 1;
        std::vector<float> xyz(3 * n);
        std::normal_distribution<float> g(0.f, trial % 3 == 0 ? 0.01f : 20.f);
        for (auto& v : xyz) v = g(rng);
        if (trial % 4 == 1) for (size_t i = 0; i < n; i += 3) { xyz[3*i] = 1.f; xyz[3*i+1] = 2.f; xyz[3*i+2] = 3.f; }   // duplicates
        if (trial % 7 == 2) for (size_t i = 0; i < n; ++i) xyz[3*i+2] = 0.f;                                            // planar
        locgpu::PackedKdTree t; std::string err;
        if (!locgpu::build_packed_kdtree(xyz.data(), n, t, err)) { std::printf("build failed: %s\n", err.c_str()); return 1; }
        if (t.leaf_slots.size() != t.num_leaves) { std::printf("leaf list mismatch\n"); return 1; }
        for (uint32_t sl : t.leaf_slots) if (((uint32_t)(t.slots[sl] >> 32) >> 30) != 3u) { std::printf("leaf list points at a non-leaf\n"); return 1; }
        for (uint64_t w : t.slots) { sum ^= w; sum *= 1099511628211ull; }
    }
    std::printf("tree checksum %016llx\n", (unsigned long long)sum);
    std::puts("asan harness ok");
    return 0;
}
