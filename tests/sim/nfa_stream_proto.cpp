// The hand-over of candidate rectangles from the cluster form's main wave to a concurrent NFA stage (csrc/lsd_cluster.h cl_main<G, true> ->
// csrc/lsd_nfa.h k_nfa_stream; SSLAM_NFA_STREAM=1) as a CPU model with real threads: one producer writes records, publishes the number of complete
// records with every record (lagging by one) and a final count; consumer threads claim 1 .. BLOCK published records by CAS, copy and "evaluate" them, and may give up waiting; a second pass
// behind the producer (the launch behind the core) takes what is unclaimed.  Checked per run: every record processed exactly once, by a thread
// that saw its final contents, nothing beyond the final count touched.  Relaxed atomics + a release/acquire pair on the counters stand for the
// kernel's sc1 accesses and its "records, s_waitcnt vmcnt(0), counter" order.
// usage: nfa_stream_proto <runs> <max records> <consumers> <expire 0|1> [break] [ring]     (ring = 1: the hand-over through an LDS ring and a publisher, STREAM == 2)
//        (break = 1: the counter includes the record that is being written -> must fail)
#include <atomic>
#include <thread>
#include <vector>
#include <random>
#include <cstdio>
#include <cstdlib>
#include <chrono>

static const int BLOCK = 8, MAXREC = 8192;
struct Ctl { std::atomic<int> candReady{0}, candFinal{0}, claim{0}, expired{0}; };

static const int RING = 64;
struct Ring { std::atomic<int> produced{0}, consumed{0}, finished{0}; std::atomic<unsigned long long> rec[RING]; };
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

// the publisher wave of STREAM == 2 (csrc/lsd_cluster.h cl_publisher): drains the LDS ring into the staging array and publishes the counters
static void publish(Ring& r, Ctl& c, unsigned seed) {
    std::minstd_rand rng(seed);
    int published = 0;
    for (;;) {
        const int fin = r.finished.load(std::memory_order_acquire);
        const int p = r.produced.load(std::memory_order_acquire);
        if (p > published) {
            const int n = std::min(p - published, 5);
            for (int i = published; i < published + n; ++i) staged[i].store(r.rec[i & (RING - 1)].load(std::memory_order_relaxed), std::memory_order_relaxed);
            published += n;
            c.candReady.store(published, std::memory_order_release);
            r.consumed.store(published, std::memory_order_release);
            continue;
        }
        if (fin) break;
        if (rng() % 4 == 0) std::this_thread::yield();
    }
    c.candFinal.store(1 + published, std::memory_order_release);
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 100, maxRec = argc > 2 ? atoi(argv[2]) : 300, nCons = argc > 3 ? atoi(argv[3]) : 8;
    const bool expire = argc > 4 && atoi(argv[4]) != 0, broken = argc > 5 && atoi(argv[5]) != 0, ring = argc > 6 && atoi(argv[6]) != 0;
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
        if (ring) {   // the main wave hands over through the ring; a publisher thread stands for wave 2 of its workgroup
            Ring r; std::thread pub(publish, std::ref(r), std::ref(c), top());
            std::minstd_rand rng(top());
            for (int i = 0; i < n; ++i) {
                while (i - r.consumed.load(std::memory_order_acquire) >= RING) std::this_thread::yield();
                if (broken) r.produced.store(i + 1, std::memory_order_release);
                r.rec[i & (RING - 1)].store((unsigned long long)i ^ salt, std::memory_order_relaxed);
                if (!broken) r.produced.store(i + 1, std::memory_order_release);
                if (rng() % 64 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 30));
            }
            r.finished.store(1, std::memory_order_release);
            pub.join();
        } else
        {   // the main wave
            std::minstd_rand rng(top());
            for (int i = 0; i < n; ++i) {
                c.candReady.store(broken ? i + 1 : i, std::memory_order_release);      // the kernel publishes with a lag of one record: [0, i) are complete when record i is written
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
