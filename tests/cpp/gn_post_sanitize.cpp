// tests/cpp/gn_post_sanitize.cpp — the host side of the finished-scan post (loc_lib_amd/csrc/gn_post.hpp) against TORN READS.
// The solve kernel's stores to pinned host memory carry no fence: the record's ten words and the {tag, checksum} pair may become
// visible in any order. This driver plays the device: it makes the words of a new post visible one at a time in EVERY order it can
// afford (all 12! is too many: every prefix order of a few thousand random permutations, plus the adversarial ones — pair first,
// pair last, the checksum before the tag), on top of the previous post's words, and after every single word asks gn_post_take():
// it may only ever say "yes" when every word it hands back is the new post's; it must say "no" for a stale call number and for an
// iteration already seen; it must say "yes" once everything has arrived. A second part runs a writer thread against a polling
// reader (relaxed atomics both sides, as the hardware gives them). Built with -fsanitize=address,undefined by the CPU suite.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

#include "gn_post.hpp"

using locgpu::GnPostRecord;
using locgpu::gn_post_sum;
using locgpu::gn_post_take;

constexpr int kWordAt = 16;  // locgpu_batch::kPostWord
static unsigned long long area[32];

struct Post { unsigned long long tag; GnPostRecord rec; unsigned long long sum; };
static Post make_post(std::mt19937_64& rng, unsigned int call, int iterations, bool done) {
    Post p;
    p.tag = ((unsigned long long)call << 32) | ((unsigned long long)(unsigned)iterations << 1) | (done ? 1ull : 0ull);
    for (auto& w : p.rec.w) w = rng();
    if (rng() % 4 == 0) p.rec.w[rng() % 10] = 0;  // words that happen to equal what was there before
    p.sum = gn_post_sum(p.tag, p.rec);
    return p;
}
static void store(int slot, const Post& p) {  // slot 0..9 record words, 10 the tag, 11 the checksum
    unsigned long long* at = slot < 10 ? &area[slot] : &area[kWordAt + (slot - 10)];
    const unsigned long long v = slot < 10 ? p.rec.w[slot] : (slot == 10 ? p.tag : p.sum);
    __atomic_store_n(at, v, __ATOMIC_RELAXED);
}

int main() {
    std::mt19937_64 rng(20261004);
    long accepted_early = 0, wrong = 0, missed = 0, checks = 0;
    for (int trial = 0; trial < 6000; ++trial) {
        const unsigned int call = 1 + (unsigned)(rng() % 1000);
        const int it_old = 1 + (int)(rng() % 15), it_new = it_old + 1 + (int)(rng() % 3);
        // what the previous finished post of this buffer left behind (an older call), then this call's progress word of iteration it_old
        const Post prev = make_post(rng, call - 1, 1 + (int)(rng() % 19), true);
        for (int s = 0; s < 12; ++s) store(s, prev);
        unsigned long long tag;
        GnPostRecord rec;
        if (gn_post_take(area, kWordAt, call, 0, &tag, &rec)) { ++wrong; }  // a stale call's word is not this call's
        const Post prog = make_post(rng, call, it_old, false);
        store(10, prog);
        if (!gn_post_take(area, kWordAt, call, it_old - 1, &tag, &rec) || tag != prog.tag) ++missed;  // a progress word needs no record
        if (gn_post_take(area, kWordAt, call, it_old, &tag, &rec)) ++wrong;                            // ... and is seen once
        // the finished post arrives word by word in some order
        const Post fin = make_post(rng, call, it_new, true);
        std::vector<int> order(12);
        std::iota(order.begin(), order.end(), 0);
        switch (trial % 4) {
            case 0: std::shuffle(order.begin(), order.end(), rng); break;
            case 1: order = {10, 11, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9}; std::shuffle(order.begin() + 2, order.end(), rng); break;   // the pair overtakes the record
            case 2: order = {11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0}; break;                                                       // checksum before tag, record backwards
            case 3: std::shuffle(order.begin(), order.begin() + 10, rng); break;                                                  // the order the kernel issues them in
        }
        for (int n = 0; n < 12; ++n) {
            store(order[n], fin);
            ++checks;
            const bool yes = gn_post_take(area, kWordAt, call, it_old, &tag, &rec);
            // what the host would now read: complete iff every word equals the new post's (a word that did not change counts as arrived)
            bool complete = __atomic_load_n(&area[kWordAt], __ATOMIC_RELAXED) == fin.tag && __atomic_load_n(&area[kWordAt + 1], __ATOMIC_RELAXED) == fin.sum;
            for (int i = 0; i < 10; ++i) complete = complete && area[i] == fin.rec.w[i];
            if (yes && (tag != fin.tag || std::memcmp(rec.w, fin.rec.w, sizeof(rec.w)) != 0)) ++wrong;
            if (yes && !complete) ++accepted_early;
            if (!yes && complete) ++missed;
        }
    }
    // writer thread against a polling reader: 20 000 posts, each record's words stored in a random order, the pair at a random place
    std::atomic<int> go{0};
    long thread_wrong = 0, taken = 0;
    std::vector<Post> posts;
    for (int i = 0; i < 20000; ++i) posts.push_back(make_post(rng, 7, i + 1, true));
    std::memset(area, 0, sizeof(area));
    std::thread writer([&] {
        std::mt19937_64 r2(99);
        for (size_t i = 0; i < posts.size(); ++i) {
            while (go.load(std::memory_order_acquire) != (int)i) {}
            std::vector<int> order(12);
            std::iota(order.begin(), order.end(), 0);
            std::shuffle(order.begin(), order.end(), r2);
            for (int s : order) { store(s, posts[i]); if (r2() % 3 == 0) std::this_thread::yield(); }
        }
    });
    for (size_t i = 0; i < posts.size(); ++i) {
        go.store((int)i, std::memory_order_release);
        unsigned long long tag;
        GnPostRecord rec;
        while (!gn_post_take(area, kWordAt, 7, (int)i, &tag, &rec)) {}
        ++taken;
        if (tag != posts[i].tag || std::memcmp(rec.w, posts[i].rec.w, sizeof(rec.w)) != 0) ++thread_wrong;
    }
    writer.join();
    std::printf("gn_post: %ld single-word checks, accepted early %ld, wrong content %ld, missed %ld; threaded: %ld taken, %ld wrong\n", checks, accepted_early, wrong, missed,
                taken, thread_wrong);
    return (accepted_early || wrong || missed || thread_wrong) ? 1 : 0;
}
