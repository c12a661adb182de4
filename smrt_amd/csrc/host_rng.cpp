// Host-side draws of smrt_balance on numpy's LEGACY stream (plain C++, no device code).
//
// The reference draws the latent normals of every class (smrt/autoencode/autoencoder.py:549-572 via
// smrt/balance/smrt.py:226-238: random_state.normal(size=(n, n_latent))) and the output permutation
// (smrt/balance/smrt.py:244: random_state.permutation) from ONE numpy RandomState, in a fixed order; a drop-in must
// consume that stream exactly.  numpy's legacy generator is a scalar loop: Mersenne Twister word by word, the polar
// method's log + sqrt per accepted pair in the same loop.  On C2 (2.8 M normals + a 1.18 M-element shuffle) that loop IS
// the wall-clock of smrt_balance once the VAE trains on the GPU behind it (47 of 86 ms, tools/c2_timing.py).
//
// Here the same numbers, bit for bit, in two phases:
//   1. sequential: MT19937 regenerated a block of 624 words at a time (the twist and the tempering as loops over the
//      block, which the compiler vectorises), pairs (x1, x2) accepted / rejected exactly as legacy_gauss does;
//   2. parallel: f = sqrt(-2 log(r2) / r2), f * x2, f * x1 per accepted pair -- the same libm calls per element as
//      numpy's, over a pool of threads.
// The state (key[624], pos, has_gauss, cached gauss) goes in and out in RandomState.get_state() / set_state() form.
// Algorithms: numpy/random/src/mt19937/mt19937.c (mt19937_gen, mt19937_next, mt19937_next_double),
// numpy/random/src/legacy/legacy-distributions.c (legacy_gauss, legacy_normal), numpy/random/src/distributions/
// distributions.c (random_interval) and numpy/random/mtrand.pyx (shuffle's 1-d path) of the numpy releases 1.17 - 2.x.
#include "../../include/smrt_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MT_N = 624, MT_M = 397;

// The generator with a block of tempered outputs: out[i] is what mt19937_next returns for key[i].
struct Mt {
    uint32_t key[MT_N];
    uint32_t out[MT_N];
    int pos;

    void temper_block() {
        for (int i = 0; i < MT_N; ++i) {
            uint32_t y = key[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out[i] = y;
        }
    }
    void regenerate() {                                   // mt19937_gen
        int kk = 0;
        for (; kk < MT_N - MT_M; ++kk) {
            const uint32_t y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
            key[kk] = key[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; ++kk) {
            const uint32_t y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
            key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        const uint32_t y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        pos = 0;
        temper_block();
    }
    inline uint32_t next() {
        if (pos == MT_N) regenerate();
        return out[pos++];
    }
    inline double next_double() {                         // mt19937_next_double
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

bool load_state(Mt& g, const uint32_t* key, int32_t pos) {
    if (!key || pos < 0 || pos > MT_N) return false;
    std::memcpy(g.key, key, sizeof(g.key));
    g.pos = pos;
    g.temper_block();
    return true;
}

// Work arrays of the two entry points: kept per calling thread and only ever grown (up to SCRATCH_KEEP bytes; a larger one is
// released when its call ends).  A fresh std::vector is zero-filled page by page: for C2's 2.8 M normals that was 34 MB of
// page faults per call, as long as the draws themselves.
constexpr size_t SCRATCH_KEEP = size_t(256) << 20;
struct Scratch {
    void* p = nullptr; size_t cap = 0;
    void* get(size_t bytes) {
        if (bytes > cap) { std::free(p); p = std::malloc(bytes); cap = p ? bytes : 0; }
        return p;
    }
    void done() { if (cap > SCRATCH_KEEP) { std::free(p); p = nullptr; cap = 0; } }
    ~Scratch() { std::free(p); }
};
thread_local Scratch g_scratch;

template <class F>
void parallel_for(int64_t n, int n_threads, int64_t min_chunk, F&& f) {
    const int nt = std::max(1, (int)std::min<int64_t>(n_threads, (n + min_chunk - 1) / min_chunk));
    if (nt == 1) { f((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * per, hi = std::min<int64_t>(n, lo + per);
        if (lo < hi) pool.emplace_back([=, &f]() { f(lo, hi); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int smrt_rng_normal(uint32_t* key, int32_t* pos, int32_t* has_gauss, double* cached_gauss, double loc, double scale,
                    double* out, int64_t n, int32_t n_threads) {
    if (!pos || !has_gauss || !cached_gauss || (!out && n > 0) || n < 0) return -1;
    Mt g;
    if (!load_state(g, key, *pos)) return -1;
    int64_t done = 0;
    double* x1 = nullptr;
    {   // work arrays first: a failed allocation must leave the caller's state as it was
        const int64_t m0 = n - ((n > 0 && *has_gauss) ? 1 : 0), pairs0 = (m0 + 1) / 2;
        if (pairs0 > 0 && !(x1 = static_cast<double*>(g_scratch.get(sizeof(double) * 3 * (size_t)pairs0)))) return -2;
    }
    if (n > 0 && *has_gauss) {                           // legacy_gauss hands out the kept value first
        out[0] = loc + scale * *cached_gauss;
        *has_gauss = 0; *cached_gauss = 0.0;
        done = 1;
    }
    const int64_t m = n - done, pairs = (m + 1) / 2;
    if (pairs > 0) {
        // phase 1 (sequential): accepted pairs, in stream order; x2 comes out first, x1 is the kept value
        double* const x2 = x1 + pairs;
        double* const r2v = x2 + pairs;
        // (every candidate is stored and the cursor advances only past an accepted one: the 21 % rejections of the polar
        // method cost no mispredicted branch; the draws consumed are the rejection loop's, in the same order)
        for (int64_t p = 0; p < pairs;) {
            const double a = 2.0 * g.next_double() - 1.0;
            const double b = 2.0 * g.next_double() - 1.0;
            const double r2 = a * a + b * b;
            x1[p] = a; x2[p] = b; r2v[p] = r2;
            p += (int64_t)((r2 < 1.0) & (r2 != 0.0));
        }
        // phase 2 (parallel): the polar method's transcendental half, the same libm calls per element as numpy's loop
        double last_kept = 0.0;
        double* dst = out + done;
        parallel_for(pairs, n_threads, 16384, [&](int64_t lo, int64_t hi) {
            for (int64_t p = lo; p < hi; ++p) {
                const double f = std::sqrt(-2.0 * std::log(r2v[p]) / r2v[p]);
                dst[2 * p] = loc + scale * (f * x2[p]);
                const double kept = f * x1[p];
                if (2 * p + 1 < m) dst[2 * p + 1] = loc + scale * kept;
                else last_kept = kept;                   // (only the last pair of an odd count gets here: one writer)
            }
        });
        if (m & 1) { *has_gauss = 1; *cached_gauss = last_kept; }
        g_scratch.done();
    }
    std::memcpy(key, g.key, sizeof(g.key));
    *pos = g.pos;
    return 0;
}

int smrt_rng_permutation(uint32_t* key, int32_t* pos, int64_t* out, int64_t n) {
    if (!pos || (!out && n > 0) || n < 0) return -1;
    Mt g;
    if (!load_state(g, key, *pos)) return -1;
    if (n >= 2 && n <= 0x7fffffffll) {
        // The common size (every row count this library meets).  _shuffle_raw is "j = random_interval(i); swap(a[i], a[j])" for
        // i = n-1 .. 1: the draws depend on nothing but the generator, the swaps on nothing but the draws.  So: all draws
        // first, a sequential pass over the generator with no memory traffic but its own output; then the swaps, on 4-byte
        // elements (half the footprint of the int64 result: 4.7 MB for C2's 1.18 M rows) with the partner's line requested
        // a few dozen swaps ahead -- as one loop every swap was a cache miss behind a rejection loop.  Same values, same
        // generator state as numpy's loop (tests/test_host_rng.py).
        uint32_t* const js = static_cast<uint32_t*>(g_scratch.get(sizeof(uint32_t) * 2 * (size_t)n));
        if (!js) return -2;
        uint32_t* const a = js + n;
        uint32_t mask = 1;
        while (mask < (uint32_t)(n - 1)) mask = (mask << 1) | 1u;        // smallest 2^k - 1 >= n - 1
        for (int64_t i = n - 1; i >= 1;) {                               // (branch-free rejection, as in smrt_rng_normal)
            const uint32_t max = (uint32_t)i;
            while ((mask >> 1) >= max) mask >>= 1;                       // ... >= i
            const uint32_t value = g.next() & mask;
            js[(size_t)i] = value;
            i -= (int64_t)(value <= max);
        }
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = (uint32_t)i;
        constexpr int64_t AHEAD = 48;
        for (int64_t i = n - 1; i >= 1; --i) {
            if (i > AHEAD) __builtin_prefetch(&a[js[(size_t)(i - AHEAD)]], 1, 1);
            std::swap(a[(size_t)i], a[js[(size_t)i]]);
        }
        for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)a[(size_t)i];
        g_scratch.done();
        std::memcpy(key, g.key, sizeof(g.key));
        *pos = g.pos;
        return 0;
    }
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {               // _shuffle_raw: j = random_interval(i), swap
        const uint64_t max = (uint64_t)i;
        uint64_t mask = max, value;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((value = ((uint64_t)g.next() & mask)) > max) {}
        } else {                                         // next_uint64 = (next32 << 32) | next32
            for (;;) {
                const uint64_t hi = g.next(), lo = g.next();
                value = ((hi << 32) | lo) & mask;
                if (value <= max) break;
            }
        }
        std::swap(out[i], out[(int64_t)value]);
    }
    std::memcpy(key, g.key, sizeof(g.key));
    *pos = g.pos;
    return 0;
}

}  // extern "C"
