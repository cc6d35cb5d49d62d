// tests/cpp/host_worker_tsan.cpp — the context's helper thread (loc_lib_amd/csrc/host_worker.hpp) under ThreadSanitizer, used the way
// locgpu_*_scan_match uses it: run() hands over a job that writes memory the caller reads after wait(); a second run() while the first
// job is still busy waits for it; jobs capture by reference things that live on the caller's stack; wait() without a job returns at
// once; the destructor joins a thread that is idle, busy or was never started.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "host_worker.hpp"

int main() {
    long bad = 0;
    {
        locgpu::HostWorker w;
        w.wait();  // nothing to wait for
        std::vector<int> src(1 << 16), dst(1 << 16, -1);
        for (int rep = 0; rep < 2000; ++rep) {
            for (size_t i = 0; i < src.size(); i += 4099) src[i] = rep;
            void* out = nullptr;
            w.run([&] { std::memcpy(dst.data(), src.data(), src.size() * sizeof(int)); out = dst.data(); });  // "size the cloud, copy the fields"
            // the caller does its own work meanwhile (the alignment), then needs the job's result
            volatile int spin = 0;
            for (int k = 0; k < (rep % 7) * 50; ++k) spin = spin + k;
            w.wait();
            if (out != dst.data() || std::memcmp(dst.data(), src.data(), src.size() * sizeof(int)) != 0) ++bad;
            // two-thread scatter: odd pieces on the helper, even ones here
            std::atomic<int> done{0};
            w.run([&] { for (size_t i = 1; i < dst.size(); i += 2) dst[i] = -rep; done.fetch_add(1); });
            for (size_t i = 0; i < dst.size(); i += 2) dst[i] = rep;
            w.wait();
            if (done.load() != 1 || dst[1] != -rep || dst[2] != rep) ++bad;
        }
        // a second run() while the first is busy: it waits, the jobs do not overlap
        std::atomic<int> inside{0}, overlap{0};
        for (int rep = 0; rep < 200; ++rep) {
            auto job = [&] { if (inside.fetch_add(1) != 0) overlap.fetch_add(1); std::this_thread::sleep_for(std::chrono::microseconds(50)); inside.fetch_sub(1); };
            w.run(job);
            w.run(job);
        }
        w.wait();
        bad += overlap.load();
        w.run([] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); });  // destroyed while busy: the destructor joins
    }
    { locgpu::HostWorker never_started; }
    { locgpu::HostWorker idle; idle.run([] {}); idle.wait(); }
    std::printf("host_worker: %ld problems\n", bad);
    return bad ? 1 : 0;
}
