// tests/cpp/host_build_tsan.cpp — the persistent thread pool of the host-side target ingest under concurrent callers: several threads (two
// contexts ingesting at once, as N ranks' worth of callers would inside one process) build trees of different sizes at the same
// time, each result is compared with a single-threaded build of the same cloud. Meant for -fsanitize=thread (tests/test_abi_and_host.py);
// also exercises the fork path: a child process forked after the pool exists builds a tree with a pool of its own.
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "kdtree_build.hpp"

static std::vector<float> cloud(size_t n, unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> g(0.f, 25.f);
    std::vector<float> xyz(3 * n);
    for (auto& v : xyz) v = g(rng);
    return xyz;
}

int main() {
    const size_t sizes[4] = {150000, 35000, 90000, 1200};
    std::vector<locgpu::PackedKdTree> want(4);
    for (int i = 0; i < 4; ++i) {
        std::string err;
        const auto xyz = cloud(sizes[i], 100 + i);
        if (!locgpu::build_packed_kdtree(xyz.data(), sizes[i], want[i], err)) { std::printf("reference build failed: %s\n", err.c_str()); return 1; }
    }
    int bad = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([&, t] {
            for (int rep = 0; rep < 6; ++rep) {
                const int i = (t + rep) % 4;
                const auto xyz = cloud(sizes[i], 100 + i);
                locgpu::PackedKdTree got;
                std::string err;
                if (!locgpu::build_packed_kdtree(xyz.data(), sizes[i], got, err) || got.slots != want[i].slots || got.leaf_slots != want[i].leaf_slots || got.depth != want[i].depth)
                    __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            }
        });
    for (auto& t : th) t.join();
    if (bad) { std::printf("%d concurrent builds differ from the single-caller result\n", bad); return 1; }
    const pid_t child = fork();
    if (child == 0) {  // the parent's pool threads do not exist here
        const auto xyz = cloud(sizes[1], 101);
        locgpu::PackedKdTree got;
        std::string err;
        const bool ok = locgpu::build_packed_kdtree(xyz.data(), sizes[1], got, err) && got.slots == want[1].slots;
        _exit(ok ? 0 : 3);
    }
    int status = 0;
    if (child < 0 || waitpid(child, &status, 0) != child || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { std::printf("forked child failed (%d)\n", status); return 1; }
    std::puts("tsan harness ok");
    return 0;
}
