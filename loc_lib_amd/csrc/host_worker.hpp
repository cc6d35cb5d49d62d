// loc_lib_amd/csrc/host_worker.hpp — one helper thread per context for host work that runs beside a blocking call's GPU work. Plain C++,
// no HIP in it: the CPU suite drives it under ThreadSanitizer (tests/cpp/host_worker_tsan.cpp).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace locgpu {
// One helper thread per context for host work that can run beside a blocking call's GPU work (round 6: the copy of the source cloud's
// non-coordinate fields into the caller's output cloud during ScanMatch). run() hands over one job, wait() returns when it is done;
// the caller's thread is the only one that calls either.
class HostWorker {
public:
    ~HostWorker() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        if (t_.joinable()) t_.join();
    }
    void run(std::function<void()> job) {
        wait();
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); busy_ = true; }
        if (!t_.joinable()) t_ = std::thread([this] { loop(); });
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !busy_; });
    }
private:
    void loop() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [this] { return stop_ || (busy_ && job_); });
            if (stop_) return;
            std::function<void()> job = std::move(job_);
            job_ = nullptr;
            g.unlock();
            job();
            g.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::thread t_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, stop_ = false;
};
}  // namespace locgpu

