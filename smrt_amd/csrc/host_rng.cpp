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
    if (n > 0 && *has_gauss) {                           // legacy_gauss hands out the kept value first
        out[0] = loc + scale * *cached_gauss;
        *has_gauss = 0; *cached_gauss = 0.0;
        done = 1;
    }
    const int64_t m = n - done, pairs = (m + 1) / 2;
    if (pairs > 0) {
        // phase 1 (sequential): accepted pairs, in stream order; x2 comes out first, x1 is the kept value
        std::vector<double> x1(pairs), x2(pairs), r2v(pairs);
        for (int64_t p = 0; p < pairs; ++p) {
            double a, b, r2;
            do {
                a = 2.0 * g.next_double() - 1.0;
                b = 2.0 * g.next_double() - 1.0;
                r2 = a * a + b * b;
            } while (r2 >= 1.0 || r2 == 0.0);
            x1[p] = a; x2[p] = b; r2v[p] = r2;
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
    }
    std::memcpy(key, g.key, sizeof(g.key));
    *pos = g.pos;
    return 0;
}

int smrt_rng_permutation(uint32_t* key, int32_t* pos, int64_t* out, int64_t n) {
    if (!pos || (!out && n > 0) || n < 0) return -1;
    Mt g;
    if (!load_state(g, key, *pos)) return -1;
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
