// tests/cpp/pool_sched_sanitize.cpp — the open-scan pool's bookkeeping (loc_lib_amd/csrc/pool_sched.hpp) on the CPU, under
// AddressSanitizer / UBSan (tests/test_abi_and_host.py). No GPU here: the "device" is a table of how many Gauss–Newton iterations
// each scan needs, the convergence flags every rank would read back after the all-reduce.
//
//   ranks R   R simulated ranks, each with a PoolSched of its own and a different shard [first, first + n_local) of every job, make
//             the calls scan_pool.hip makes (accept at submit; per chunk: finish the scans whose flag is set, admit, run `chunk`
//             iterations = `chunk` collectives when anything is open). Checked after every call: every rank took the SAME decisions
//             (slot, job, scan, region of every admission; free slots / regions; collectives issued) — what RCCL needs not to hang
//             and what makes a scan's sums land in the same row of the [slots][32] exchange buffer on every rank.
//   stall     one rank, admission with copies that are "still on their way" at random: FIFO (no job overtakes an older one), no slot
//             or region held twice, every job completes, everything is free at the end, cancel_last gives the regions back.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "pool_sched.hpp"

using namespace locgpu;

namespace {

uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

#define CHECK(c, ...)                                                              \
    do {                                                                           \
        if (!(c)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } \
    } while (0)

struct Rank {
    std::unique_ptr<PoolSched> sched;
    std::map<int64_t, std::unique_ptr<PoolSchedJob>> jobs;
    std::vector<int> left;  // [slots] iterations the scan in the slot still needs (replicated: every rank solves every slot)
    long long collectives = 0;
    std::vector<PoolAdmitted> last;
};

int iterations_of(int64_t ticket, int idx) { return 1 + (int)(mix((uint64_t)ticket * 1315423911ULL + (uint64_t)idx) % 19); }

// no slot and no region is held by two live scans
void check_exclusive(const PoolSched& s, const std::map<int64_t, std::unique_ptr<PoolSchedJob>>& jobs, int regions) {
    std::set<int> reg;
    int live = 0;
    for (int sl = 0; sl < s.slots(); ++sl) {
        PoolSchedJob* j = s.job_of(sl);
        if (!j) continue;
        ++live;
        const int r = j->region[(size_t)s.idx_of(sl)];
        CHECK(r >= 0 && r < regions, "region %d out of range", r);
        CHECK(reg.insert(r).second, "region %d held by two live scans", r);
    }
    CHECK(live + s.free_slots() == s.slots(), "slots leak: %d live + %d free != %d", live, s.free_slots(), s.slots());
    // regions: free + those of scans not finished (live or still waiting) = all
    int held = 0;
    for (auto& kv : jobs) held += kv.second->remaining;
    CHECK(held + s.free_regions() == regions, "regions leak: %d held + %d free != %d", held, s.free_regions(), regions);
}

int run_ranks(int R, int slots, int prefetch, int chunk, int n_jobs, uint64_t seed) {
    const int regions = slots + prefetch;
    std::vector<Rank> rk((size_t)R);
    for (auto& r : rk) { r.sched.reset(new PoolSched(slots, regions)); r.left.assign((size_t)slots, 0); }
    int64_t next_ticket = 1;
    int submitted = 0;
    long long total_scans = 0, finished = 0;
    std::vector<int64_t> order;  // admission order of (ticket << 20 | idx) on rank 0: FIFO check
    auto turn = [&]() {  // one pool_pump on every rank: collect, admit, run a chunk
        for (int q = 0; q < R; ++q) {
            Rank& r = rk[(size_t)q];
            for (int sl = 0; sl < slots; ++sl)
                if (r.sched->job_of(sl) && r.left[(size_t)sl] == 0) { r.sched->finish(sl); if (q == 0) ++finished; }
            r.last.clear();
            r.sched->admit([](PoolSchedJob*, int) { return true; }, r.last);  // several ranks: never defer
            for (const PoolAdmitted& a : r.last) r.left[(size_t)a.slot] = iterations_of(a.job->ticket, a.idx);
            int open = 0;
            for (int sl = 0; sl < slots; ++sl) open += r.sched->job_of(sl) ? 1 : 0;
            if (open) {
                r.collectives += chunk;
                for (int sl = 0; sl < slots; ++sl)
                    if (r.sched->job_of(sl)) r.left[(size_t)sl] = r.left[(size_t)sl] > chunk ? r.left[(size_t)sl] - chunk : 0;
            }
            check_exclusive(*r.sched, r.jobs, regions);
        }
        // every rank decided the same
        for (int q = 1; q < R; ++q) {
            CHECK(rk[(size_t)q].last.size() == rk[0].last.size(), "rank %d admitted %zu scans, rank 0 %zu", q, rk[(size_t)q].last.size(), rk[0].last.size());
            for (size_t i = 0; i < rk[0].last.size(); ++i) {
                const PoolAdmitted &a = rk[0].last[i], &b = rk[(size_t)q].last[i];
                CHECK(a.slot == b.slot && a.idx == b.idx && a.job->ticket == b.job->ticket, "rank %d: admission %zu differs", q, i);
                CHECK(a.job->region[(size_t)a.idx] == b.job->region[(size_t)b.idx], "rank %d: region of admission %zu differs", q, i);
            }
            CHECK(rk[(size_t)q].collectives == rk[0].collectives, "rank %d issued %lld collectives, rank 0 %lld", q, rk[(size_t)q].collectives, rk[0].collectives);
            CHECK(rk[(size_t)q].sched->free_slots() == rk[0].sched->free_slots() && rk[(size_t)q].sched->free_regions() == rk[0].sched->free_regions(), "rank %d: free lists differ", q);
        }
        for (const PoolAdmitted& a : rk[0].last) order.push_back((a.job->ticket << 20) | a.idx);
        // smallest free slot first: the admissions of one turn take ascending slots
        for (size_t i = 1; i < rk[0].last.size(); ++i) CHECK(rk[0].last[i].slot > rk[0].last[i - 1].slot, "slots not handed out smallest first");
    };
    while (submitted < n_jobs) {
        const int n_total = 1 + (int)(mix(seed + 977 * (uint64_t)submitted) % (uint64_t)(regions < 48 ? regions : 48));
        // room: let scans finish (locgpu_pool_submit's loop)
        int guard = 0;
        while (rk[0].sched->free_regions() < n_total) { turn(); CHECK(++guard < 100000, "no progress while waiting for regions"); }
        const int64_t ticket = next_ticket++;
        for (int q = 0; q < R; ++q) {
            auto j = std::make_unique<PoolSchedJob>();
            j->ticket = ticket;
            j->n_total = n_total;
            // contiguous shards, the last ranks possibly empty (multi_gpu.shard_range's shape)
            const int per = (n_total + R - 1) / R;
            j->first = std::min(n_total, q * per);
            j->n_local = std::min(n_total, (q + 1) * per) - j->first;
            CHECK(rk[(size_t)q].sched->accept(j.get()), "accept refused with %d free regions for %d scans", rk[(size_t)q].sched->free_regions(), n_total);
            rk[(size_t)q].jobs[ticket] = std::move(j);
        }
        for (int q = 1; q < R; ++q) CHECK(rk[(size_t)q].jobs[ticket]->region == rk[0].jobs[ticket]->region, "rank %d: regions of job %lld differ", q, (long long)ticket);
        total_scans += n_total;
        ++submitted;
        if (mix(seed ^ (uint64_t)submitted) % 3 == 0) turn();  // the caller submits several jobs between turns, at times
        // hand back finished jobs (locgpu_pool_wait)
        for (int q = 0; q < R; ++q)
            for (auto it = rk[(size_t)q].jobs.begin(); it != rk[(size_t)q].jobs.end();) it = it->second->remaining == 0 ? rk[(size_t)q].jobs.erase(it) : std::next(it);
    }
    int guard = 0;
    while (finished < total_scans) { turn(); CHECK(++guard < 1000000, "the pool stopped with %lld of %lld scans finished", finished, total_scans); }
    for (int q = 0; q < R; ++q) {
        CHECK(rk[(size_t)q].sched->free_slots() == slots && rk[(size_t)q].sched->free_regions() == regions && rk[(size_t)q].sched->waiting() == 0, "rank %d: not everything returned", q);
        for (auto& kv : rk[(size_t)q].jobs) CHECK(kv.second->remaining == 0, "job unfinished");
    }
    for (size_t i = 1; i < order.size(); ++i) CHECK(order[i] > order[i - 1], "admission order is not FIFO at %zu", i);
    CHECK((long long)order.size() == total_scans, "admitted %zu of %lld", order.size(), total_scans);
    printf("ranks=%d slots=%d regions=%d chunk=%d jobs=%d scans=%lld collectives=%lld OK\n", R, slots, regions, chunk, n_jobs, total_scans, rk[0].collectives);
    return 0;
}

int run_stall(int slots, int prefetch, int n_jobs, uint64_t seed) {
    const int regions = slots + prefetch;
    PoolSched s(slots, regions);
    std::map<int64_t, std::unique_ptr<PoolSchedJob>> jobs;
    std::vector<int> left((size_t)slots, 0);
    std::vector<int64_t> order;
    uint64_t rng = seed;
    long long total = 0, finished = 0, stalls = 0, cancelled = 0;
    auto turn = [&]() {
        for (int sl = 0; sl < slots; ++sl)
            if (s.job_of(sl) && left[(size_t)sl] == 0) { s.finish(sl); ++finished; }
        std::vector<PoolAdmitted> adm;
        s.admit([&](PoolSchedJob*, int) { rng = mix(rng + 1); const bool ok = rng % 4 != 0; stalls += ok ? 0 : 1; return ok; }, adm);
        for (const PoolAdmitted& a : adm) { left[(size_t)a.slot] = iterations_of(a.job->ticket, a.idx); order.push_back((a.job->ticket << 20) | a.idx); }
        for (int sl = 0; sl < slots; ++sl)
            if (s.job_of(sl)) left[(size_t)sl] = left[(size_t)sl] > 4 ? left[(size_t)sl] - 4 : 0;
        check_exclusive(s, jobs, regions);
    };
    int64_t ticket = 1;
    for (int n = 0; n < n_jobs; ++n) {
        const int n_total = 1 + (int)(mix(seed + 31 * (uint64_t)n) % (uint64_t)(regions < 40 ? regions : 40));
        int guard = 0;
        while (s.free_regions() < n_total) { turn(); CHECK(++guard < 100000, "no progress"); }
        auto j = std::make_unique<PoolSchedJob>();
        j->ticket = ticket;
        j->n_total = j->n_local = n_total;
        const int before = s.free_regions();
        CHECK(s.accept(j.get()), "accept refused");
        if (mix(seed * 7 + (uint64_t)n) % 5 == 0) {  // its copy could not be started: the job is taken back
            s.cancel_last(j.get());
            CHECK(s.free_regions() == before, "cancel_last did not give the regions back");
            ++cancelled;
            // the next job gets the same regions again, smallest first
            continue;
        }
        total += n_total;
        jobs[ticket++] = std::move(j);
        if (n % 2) turn();
        for (auto it = jobs.begin(); it != jobs.end();) it = it->second->remaining == 0 ? jobs.erase(it) : std::next(it);
    }
    int guard = 0;
    while (finished < total) { turn(); CHECK(++guard < 1000000, "stopped with %lld of %lld", finished, total); }
    CHECK(s.free_slots() == slots && s.free_regions() == regions && s.waiting() == 0, "not everything returned");
    for (size_t i = 1; i < order.size(); ++i) CHECK(order[i] > order[i - 1], "a job overtook an older one at %zu", i);
    // refusals: a job larger than the arena, an empty job
    PoolSchedJob big;
    big.n_total = regions + 1;
    CHECK(!s.accept(&big), "a job larger than the arena was accepted");
    PoolSchedJob none;
    CHECK(!s.accept(&none), "an empty job was accepted");
    s.finish(0);  // a free slot: nothing happens
    CHECK(s.free_slots() == slots, "finish() on a free slot changed the free list");
    printf("stall slots=%d regions=%d jobs=%d scans=%lld stalls=%lld cancelled=%lld OK\n", slots, regions, n_jobs, total, stalls, cancelled);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "ranks")) {
        const int R = argc > 2 ? atoi(argv[2]) : 2;
        int rc = 0;
        rc |= run_ranks(R, 256, 256, 4, 300, 1);
        rc |= run_ranks(R, 64, 0, 4, 200, 2);   // no prefetch regions: a job waits for whole scans to leave
        rc |= run_ranks(R, 7, 3, 1, 200, 3);    // tiny pool, jobs larger than the slots
        rc |= run_ranks(R, 256, 64, 8, 150, 4);
        return rc;
    }
    if (argc >= 2 && !strcmp(argv[1], "stall")) return run_stall(64, 32, 400, 5) | run_stall(5, 0, 300, 6) | run_stall(256, 256, 300, 7);
    fprintf(stderr, "usage: pool_sched_sanitize ranks R | stall\n");
    return 2;
}
