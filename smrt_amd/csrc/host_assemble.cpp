// Host-side output assembly of smrt_balance (plain C++, no device code).
//
// The reference builds its result with two full fp64 copies of X (smrt/balance/base.py:24, smrt/balance/smrt.py:184), one
// np.vstack per class (smrt.py:237) and a fancy-index gather for the shuffle (smrt.py:248-250): at 1M+ rows that
// is what is left of the wall-clock once the VAE runs on the GPU.  Here the balanced matrix is written ONCE:
// output row i is row order[i] of the virtual concatenation [X ; synthetic], converted to fp64 on the fly, by a
// pool of threads that each own a contiguous range of output rows.
#include "../../include/smrt_hip.h"

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

template <class SRC>
inline void put_row(const SRC* src, double* dst, int64_t d) {
    for (int64_t j = 0; j < d; ++j) dst[j] = (double)src[j];
}
template <>
inline void put_row<double>(const double* src, double* dst, int64_t d) {
    std::memcpy(dst, src, sizeof(double) * (size_t)d);
}

template <class SRC, class SYN>
void assemble_range(const SRC* X, int64_t n_orig, int64_t d, int64_t ldx, const SYN* synth, int64_t ld_synth,
                    const int64_t* order, int64_t lo, int64_t hi, double* out, int which) {
    for (int64_t i = lo; i < hi; ++i) {
        const int64_t r = order ? order[i] : i;
        double* dst = out + i * d;
        if (r < n_orig) { if (which != 2) put_row<SRC>(X + r * ldx, dst, d); }
        else            { if (which != 1) put_row<SYN>(synth + (r - n_orig) * ld_synth, dst, d); }
    }
}

template <class SYN>
int assemble(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
             const SYN* synth, int64_t n_synth, int64_t ld_synth,
             const int64_t* order, int64_t n_out, double* out, int32_t n_threads, int which = 0) {
    if (which < 0 || which > 2) return -1;
    if ((!X && n_orig > 0 && which != 2) || (!synth && n_synth > 0 && which != 1) || !out) return -1;
    if (n_orig < 0 || n_synth < 0 || d < 1 || ldx < d || (n_synth > 0 && ld_synth < d) || n_out < 0) return -1;
    if (!order && n_out != n_orig + n_synth) return -1;
    if (order)
        for (int64_t i = 0; i < n_out; ++i)
            if (order[i] < 0 || order[i] >= n_orig + n_synth) return -1;      // never read outside the inputs
    const int64_t min_rows = 4096;                                        // below this a thread is not worth starting
    int nt = std::max(1, (int)std::min<int64_t>(n_threads, (n_out + min_rows - 1) / min_rows));
    auto run = [&](int64_t lo, int64_t hi) {
        if (x_is_f32) assemble_range<float, SYN>((const float*)X, n_orig, d, ldx, synth, ld_synth, order, lo, hi, out, which);
        else          assemble_range<double, SYN>((const double*)X, n_orig, d, ldx, synth, ld_synth, order, lo, hi, out, which);
    };
    if (nt == 1) { run(0, n_out); return 0; }
    std::vector<std::thread> pool;
    const int64_t per = (n_out + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * per, hi = std::min<int64_t>(n_out, lo + per);
        if (lo < hi) pool.emplace_back(run, lo, hi);
    }
    for (auto& th : pool) th.join();
    return 0;
}

}  // namespace

extern "C" int smrt_assemble_rows(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                                  const float* synth, int64_t n_synth, int64_t ld_synth,
                                  const int64_t* order, int64_t n_out, double* out, int32_t n_threads) {
    return assemble<float>(X, x_is_f32, n_orig, d, ldx, synth, n_synth, ld_synth, order, n_out, out, n_threads);
}

// Same pass with fp64 appended rows: smote_balance (smote.py:93,254-260), and -- with no appended rows at all -- the
// pure row gathers of over_sample_balance (over.py:77-81) and under_sample_balance (under.py:69-78).
extern "C" int smrt_assemble_rows_f64(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                                      const double* synth, int64_t n_synth, int64_t ld_synth,
                                      const int64_t* order, int64_t n_out, double* out, int32_t n_threads) {
    return assemble<double>(X, x_is_f32, n_orig, d, ldx, synth, n_synth, ld_synth, order, n_out, out, n_threads);
}

// The same pass in two halves, so that the rows that come from X (85 % of the output on C2) are written while the GPU is
// still training: which = 1 writes only the output rows whose source is X (synth may be NULL), which = 2 only those whose
// source is a synthetic row (X may be NULL), 0 everything.  Rows of the other kind are left untouched.
extern "C" int smrt_assemble_rows_part(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                                       const float* synth, int64_t n_synth, int64_t ld_synth,
                                       const int64_t* order, int64_t n_out, double* out, int32_t n_threads, int32_t which) {
    return assemble<float>(X, x_is_f32, n_orig, d, ldx, synth, n_synth, ld_synth, order, n_out, out, n_threads, which);
}

// The synthetic half in pieces (no counterpart in the reference): only the output rows whose source is one of the synthetic
// rows [s_lo, s_hi) -- smrt_balance writes a class's rows while the next class's are still being decoded and downloaded.
extern "C" int smrt_assemble_rows_synth_range(const float* synth, int64_t n_orig, int64_t n_synth, int64_t ld_synth, int64_t d,
                                              const int64_t* order, int64_t n_out, double* out, int32_t n_threads,
                                              int64_t s_lo, int64_t s_hi) {
    if (!synth || !out || n_orig < 0 || n_synth < 0 || d < 1 || ld_synth < d || n_out < 0) return -1;
    if (s_lo < 0 || s_hi < s_lo || s_hi > n_synth) return -1;
    if (!order && n_out != n_orig + n_synth) return -1;
    if (s_lo == s_hi) return 0;
    const int64_t r_lo = n_orig + s_lo, r_hi = n_orig + s_hi;
    // without an order the rows are where they are in the concatenation; with one, every thread scans its share of it
    const int64_t i_lo = order ? 0 : r_lo, i_hi = order ? n_out : r_hi;
    auto run = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t r = order ? order[i] : i;
            if (r >= r_lo && r < r_hi) put_row<float>(synth + (r - n_orig) * ld_synth, out + i * d, d);
        }
    };
    const int64_t n = i_hi - i_lo, min_rows = order ? 16384 : 4096;
    const int nt = std::max(1, (int)std::min<int64_t>(n_threads, (n + min_rows - 1) / min_rows));
    if (nt == 1) { run(i_lo, i_hi); return 0; }
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = i_lo + t * per, hi = std::min<int64_t>(i_hi, lo + per);
        if (lo < hi) pool.emplace_back(run, lo, hi);
    }
    for (auto& th : pool) th.join();
    return 0;
}

namespace {
// An IEEE value is NaN or infinite exactly when its exponent field is all ones: the largest |bit pattern| of a range says
// whether any element is.  Integer max over the words vectorises (AVX2 clone picked at load time where the CPU has it).
__attribute__((target_clones("avx2", "default")))
bool finite_f32(const uint32_t* w, int64_t n) {
    uint32_t m = 0;
    for (int64_t i = 0; i < n; ++i) { const uint32_t a = w[i] & 0x7fffffffu; m = a > m ? a : m; }
    return m < 0x7f800000u;
}
__attribute__((target_clones("avx2", "default")))
bool finite_f64(const uint32_t* w, int64_t n) {          // little-endian: the high word of element i is w[2 i + 1]
    uint32_t m = 0;
    for (int64_t i = 0; i < n; ++i) { const uint32_t a = w[2 * i + 1] & 0x7fffffffu; m = a > m ? a : m; }
    return m < 0x7ff00000u;
}
}  // namespace

// check_array's finiteness sweep over X (smrt/balance/base.py:24 via sklearn), on every core instead of one: 1 when every
// element is finite, 0 when one is NaN or infinite, -1 on a bad argument.  The caller lets sklearn produce the error.
extern "C" int smrt_all_finite(const void* X, int32_t x_is_f32, int64_t n, int32_t n_threads) {
    if (!X || n < 0) return -1;
    std::atomic<int> ok{1};
    auto run = [&](int64_t lo, int64_t hi) {
        if (!(x_is_f32 ? finite_f32((const uint32_t*)X + lo, hi - lo) : finite_f64((const uint32_t*)X + 2 * lo, hi - lo))) ok.store(0);
    };
    const int64_t min_elems = 1 << 18;
    const int nt = std::max(1, (int)std::min<int64_t>(n_threads, (n + min_elems - 1) / min_elems));
    if (nt == 1) { run(0, n); return ok.load(); }
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = t * per, hi = std::min<int64_t>(n, lo + per);
        if (lo < hi) pool.emplace_back(run, lo, hi);
    }
    for (auto& th : pool) th.join();
    return ok.load();
}
