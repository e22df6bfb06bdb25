// The hand-over of candidate rectangles from the cluster form's main wave to a concurrent NFA stage (csrc/lsd_cluster.h cl_main<G, true> ->
// csrc/lsd_nfa.h k_nfa_stream; SSLAM_NFA_STREAM=1) as a CPU model with real threads: one producer writes records, publishes the number of complete
// records with every record (lagging by one) and a final count; consumer threads claim 1 .. BLOCK published records by CAS, copy and "evaluate" them, and may give up waiting; a second pass
// behind the producer (the launch behind the core) takes what is unclaimed.  Checked per run: every record processed exactly once, by a thread
// that saw its final contents, nothing beyond the final count touched.  Relaxed atomics + a release/acquire pair on the counters stand for the
// kernel's sc1 accesses and its "records, s_waitcnt vmcnt(0), counter" order.
// usage: nfa_stream_proto <runs> <max records> <consumers> <expire 0|1> [break]
//        (break = 1, the negative control: the counter includes the record that is being written, and the producer HOLDS that record back until a consumer has
//         taken it -- a forced interleaving, so the faulty order is caught in every run that has a record, not with some probability)
#include <atomic>
#include <thread>
#include <vector>
#include <random>
#include <cstdio>
#include <cstdlib>
#include <chrono>

static const int BLOCK = 8, MAXREC = 8192;
struct Ctl { std::atomic<int> candReady{0}, candFinal{0}, claim{0}, expired{0}; };

static std::atomic<unsigned long long> staged[MAXREC];      // what the main wave writes (one word per record here: index ^ salt)
static std::atomic<int> processed[MAXREC + 64];
static std::atomic<int> badValue{0};

static void consume(Ctl& c, unsigned long long salt, bool mayWait, int patience, unsigned seed) {
    std::minstd_rand rng(seed);
    int waited = 0;
    for (;;) {
        int c0 = -1, c1 = 0;
        for (;;) {
            const int fin = c.candFinal.load(std::memory_order_acquire);
            const int ready = fin ? fin - 1 : c.candReady.load(std::memory_order_acquire);
            const int cur = c.claim.load(std::memory_order_relaxed);
            if (cur < ready) {
                const int take = std::min(BLOCK, std::max(1, (ready - cur) >> 2));
                int expect = cur;
                if (c.claim.compare_exchange_strong(expect, cur + take)) { c0 = cur; c1 = cur + take; break; }
                continue;
            }
            if (fin) break;
            if (!mayWait || ++waited > patience) { c.expired.fetch_add(1); break; }
            if (rng() % 4 == 0) std::this_thread::yield();
        }
        if (c0 < 0) return;
        for (int i = c0; i < c1; ++i) {
            if (staged[i].load(std::memory_order_relaxed) != ((unsigned long long)i ^ salt)) badValue.fetch_add(1);
            processed[i].fetch_add(1);
        }
        waited = 0;
    }
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 100, maxRec = argc > 2 ? atoi(argv[2]) : 300, nCons = argc > 3 ? atoi(argv[3]) : 8;
    const bool expire = argc > 4 && atoi(argv[4]) != 0, broken = argc > 5 && atoi(argv[5]) != 0;
    int bad = 0; long long expiredTotal = 0, tailBlocks = 0;
    std::mt19937 top(12345);
    for (int run = 0; run < runs; ++run) {
        const int n = run == 0 ? 0 : run == 1 ? BLOCK : run == 2 ? BLOCK + 1 : (int)(top() % (unsigned)(maxRec + 1));
        const unsigned long long salt = ((unsigned long long)top() << 32) | top();
        Ctl c;
        for (int i = 0; i < n + 64; ++i) { processed[i].store(0); if (i < MAXREC) staged[i].store(~0ull); }
        badValue.store(0);
        std::vector<std::thread> th;
        for (int k = 0; k < nCons; ++k) th.emplace_back(consume, std::ref(c), salt, true, expire ? 50 + (int)(top() % 2000) : 1 << 30, top());
        {   // the main wave
            std::minstd_rand rng(top());
            for (int i = 0; i < n; ++i) {
                c.candReady.store(broken ? i + 1 : i, std::memory_order_release);      // the kernel publishes with a lag of one record: [0, i) are complete when record i is written
                if (broken && i == n / 2) {          // the faulty order made visible: record i is announced and not yet written -- wait (bounded) until a consumer has read it
                    const auto t0 = std::chrono::steady_clock::now();
                    while (processed[i].load() == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) std::this_thread::yield();
                }
                if (rng() % 8 == 0) std::this_thread::yield();
                staged[i].store((unsigned long long)i ^ salt, std::memory_order_relaxed);
                if (rng() % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 30));
            }
            c.candFinal.store(1 + n, std::memory_order_release);
        }
        for (auto& t : th) t.join();
        const int claimedBefore = c.claim.load();
        {   // the launch behind the core: no waiting
            std::vector<std::thread> t2;
            for (int k = 0; k < 4; ++k) t2.emplace_back(consume, std::ref(c), salt, false, 0, top());
            for (auto& t : t2) t.join();
        }
        tailBlocks += c.claim.load() - claimedBefore; expiredTotal += c.expired.load();
        bool ok = badValue.load() == 0 && c.claim.load() == n;
        for (int i = 0; i < n + 64 && ok; ++i) ok = processed[i].load() == (i < n ? 1 : 0);
        if (!ok) ++bad;
    }
    printf("expired waits %lld, records left to the second pass %lld\n", expiredTotal, tailBlocks);
    printf("bad runs: %d of %d\n", bad, runs);
    return 0;
}
