// Exact fp64 k-nearest-neighbours and the SMOTE row construction (smrt/balance/smote.py:39-98).
//
// The reference asks sklearn's kd-tree for the k+1 nearest rows of every minority row; its distances are the fp64 sums
// sum_j (a_j - b_j)^2 accumulated in feature order.  This file computes the same sums, in the same order and without
// fused multiply-adds (the file is compiled with -ffp-contract=off), for every (query, candidate) pair -- a brute-force
// sweep is a few milliseconds at these sizes on 256 CUs -- and keeps the k smallest per query, ties to the lower row
// number.  Work is fp64 VALU bound; there is no fp64 contraction to hand to the matrix cores because (a-b)^2 summed in
// order is not a dot product once bit-equality with the reference's ordering is the requirement.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/smrt_hip.h"

extern int smrt_set_error(int code, const std::string& m);

namespace {

constexpr int TQ = 64, TC = 64, DK = 16, KNN_THREADS = 256, KNN_KMAX = 64;
constexpr int SLAB = TQ + 1;            // LDS row pitch of a staged slab (doubles): odd, so the transposing store is conflict-free

// LDS: qs[DK][SLAB] cs[DK][SLAB] | dist[TQ][TC+1] | topd[TQ][k] | topi[TQ][k]
__global__ void __launch_bounds__(KNN_THREADS)
knn_kernel(const double* __restrict__ Q, long nq, long ldq, const double* __restrict__ Cm, long nc, long ldc, int d, int k,
           long long* __restrict__ idx_out, double* __restrict__ dist_out) {
    extern __shared__ double lds[];
    __shared__ unsigned flags[KNN_THREADS];
    double* qs = lds;
    double* cs = qs + DK * SLAB;
    double* dist = cs + DK * SLAB;
    double* topd = dist + TQ * (TC + 1);
    long long* topi = reinterpret_cast<long long*>(topd + TQ * k);

    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;             // thread owns queries 4 ty + a, candidates tx + 16 b
                                                                     // (16 lanes read 16 consecutive doubles: no bank conflict)
    const long q0 = (long)blockIdx.x * TQ;

    for (int i = t; i < TQ * k; i += KNN_THREADS) { topd[i] = __builtin_inf(); topi[i] = -1; }

    // Global -> LDS staging is software-pipelined: the slab of stage s + 1 (the next 16 features, or the first 16 of the
    // next candidate tile) is requested into registers right after the barrier that publishes stage s, and lands while
    // stage s is being multiplied.  Each thread carries 4 + 4 doubles; element e = t + 256 u is row e / 16, feature e % 16.
    const int nd = (d + DK - 1) / DK;
    const long n_ct = (nc + TC - 1) / TC;
    double pq[4], pc[4];
    auto fetch = [&](long ct, int js) {
        const long c0f = ct * TC;
        const int j0f = js * DK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + KNN_THREADS * u;
            const int r = e / DK, j = e % DK;
            const long qr = q0 + r, cr = c0f + r;
            const bool jv = j0f + j < d;
            pq[u] = (jv && qr < nq) ? Q[qr * ldq + j0f + j] : 0.0;
            pc[u] = (jv && cr < nc) ? Cm[cr * ldc + j0f + j] : 0.0;
        }
    };
    fetch(0, 0);
    for (long ct = 0; ct < n_ct; ++ct) {
        const long c0 = ct * TC;
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;

        for (int js = 0; js < nd; ++js) {
            const int j0 = js * DK;
            __syncthreads();                                   // the previous stage's readers are done
#pragma unroll
            for (int u = 0; u < 4; ++u) {                      // registers -> LDS, transposed to [feature][row]
                const int e = t + KNN_THREADS * u;
                const int r = e / DK, j = e % DK;
                qs[j * SLAB + r] = pq[u];
                cs[j * SLAB + r] = pc[u];
            }
            __syncthreads();
            if (js + 1 < nd) fetch(ct, js + 1);
            else if (ct + 1 < n_ct) fetch(ct + 1, 0);
            const int jn = min(DK, d - j0);
            auto feature = [&](int j) {
                double qa[4], cb[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { qa[a] = qs[j * SLAB + 4 * ty + a]; cb[a] = cs[j * SLAB + tx + 16 * a]; }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const double diff = qa[a] - cb[b];
                        acc[a][b] = acc[a][b] + diff * diff;         // not contracted: -ffp-contract=off
                    }
            };
            if (jn == DK) {
                // whole slab: straight-line, so the LDS reads of later features are in flight under the arithmetic of
                // earlier ones (feature order of every sum is unchanged)
#pragma unroll
                for (int j = 0; j < DK; ++j) feature(j);
            } else {
                for (int j = 0; j < jn; ++j) feature(j);
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) dist[(4 * ty + a) * (TC + 1) + tx + 16 * b] = acc[a][b];
        __syncthreads();
        // Selection.  After the first few tiles almost nothing beats a query's current k-th distance, so all 256 threads
        // first screen the tile (4 threads per query, 16 candidates each, loads batched) against that bound and leave a
        // bit mask; only then does the query's owner insert the few survivors, in candidate order, re-checking each
        // against the bound as it tightens (strict <: on a tie the earlier row stays ahead).
        {
            const int qi = t >> 2, part = t & 3;
            const double bound = topd[qi * k + k - 1];
            const double* dp = dist + qi * (TC + 1) + part * 16;
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = dp[u];
            unsigned m = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) m |= (v[u] < bound ? 1u : 0u) << u;
            flags[t] = m;
        }
        __syncthreads();
        if (t < TQ && q0 + t < nq) {
            double* td = topd + t * k;
            long long* ti = topi + t * k;
            const int cn = (int)min((long)TC, nc - c0);
            double worst = td[k - 1];
            for (int part = 0; part < 4; ++part) {
                unsigned m = flags[4 * t + part];
                while (m) {
                    const int c = part * 16 + __builtin_ctz(m);
                    m &= m - 1;
                    const double v = dist[t * (TC + 1) + c];
                    if (c < cn && v < worst) {
                        int p = k - 1;
                        while (p > 0 && td[p - 1] > v) { td[p] = td[p - 1]; ti[p] = ti[p - 1]; --p; }
                        td[p] = v; ti[p] = c0 + c;
                        worst = td[k - 1];
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = t; i < TQ * k; i += KNN_THREADS) {
        const long qr = q0 + i / k;
        if (qr < nq) {
            idx_out[qr * k + i % k] = topi[i];
            if (dist_out) dist_out[qr * k + i % k] = topd[i];
        }
    }
}

// out[i][:] = (x_b - x_n) * r_i + x_b  (mode 0, smote.py:22-23)   or   (x_b + x_n) / 2  (mode 1, smote.py:26-29)
__global__ void __launch_bounds__(256)
smote_rows_kernel(const double* __restrict__ X, long ldx, int d, const long long* __restrict__ base,
                  const long long* __restrict__ nbr, const double* __restrict__ r, long n_rows, int mode,
                  double* __restrict__ out, long ldo, long n_x) {
    const long total = n_rows * d;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long i = e / d;
        const int j = (int)(e % d);
        const long long bi = base[i], ni = nbr[i];
        if (bi < 0 || bi >= n_x || ni < 0 || ni >= n_x) { out[i * ldo + j] = __builtin_nan(""); continue; }   // never read outside X
        const double xb = X[bi * ldx + j], xn = X[ni * ldx + j];
        double v;
        if (mode == 0) { const double diff = xb - xn; const double prod = diff * r[i]; v = prod + xb; }
        else { const double s = xb + xn; v = s / 2.0; }
        out[i * ldo + j] = v;
    }
}

}  // namespace

#define KNNC(call)                                                                                      \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) return smrt_set_error(-2, std::string(#call ": ") + hipGetErrorString(e__)); \
    } while (0)

extern "C" int smrt_knn_f64(const double* Q_dev, int64_t nq, int64_t ldq, const double* C_dev, int64_t nc, int64_t ldc,
                            int64_t d, int32_t k, int64_t* idx_out_dev, double* rdist_out_dev, void* hip_stream) {
    if (!Q_dev || !C_dev || !idx_out_dev) return smrt_set_error(-1, "smrt_knn_f64: null pointer");
    if (nq < 1 || nc < 1 || d < 1 || d > (1 << 20) || ldq < d || ldc < d)
        return smrt_set_error(-1, "smrt_knn_f64: bad sizes");
    if (k < 1 || k > KNN_KMAX || k > nc) return smrt_set_error(-1, "smrt_knn_f64: k must be in [1, min(64, candidates)]");
    if ((nq + TQ - 1) / TQ > 0x7fffffffL) return smrt_set_error(-1, "smrt_knn_f64: too many query rows");
    const size_t lds_bytes = sizeof(double) * (2 * (size_t)DK * SLAB + (size_t)TQ * (TC + 1) + 2 * (size_t)TQ * k);
    // always the same value (the size at k = 64), so concurrent callers with different k cannot undercut each other
    constexpr size_t lds_max = sizeof(double) * (2 * (size_t)DK * SLAB + (size_t)TQ * (TC + 1) + 2 * (size_t)TQ * KNN_KMAX);
    KNNC(hipFuncSetAttribute((const void*)knn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    hipLaunchKernelGGL(knn_kernel, dim3((unsigned)((nq + TQ - 1) / TQ)), dim3(KNN_THREADS), lds_bytes, (hipStream_t)hip_stream,
                       Q_dev, (long)nq, (long)ldq, C_dev, (long)nc, (long)ldc, (int)d, (int)k,
                       (long long*)idx_out_dev, rdist_out_dev);
    KNNC(hipGetLastError());
    return 0;
}

extern "C" int smrt_smote_rows_f64(const double* X_dev, int64_t n_x, int64_t ldx, int64_t d, const int64_t* base_idx_dev,
                                   const int64_t* nbr_idx_dev, const double* r_dev, int64_t n_rows, int32_t mode,
                                   double* out_dev, int64_t ldo, void* hip_stream) {
    if (n_rows == 0) return 0;
    if (!X_dev || !base_idx_dev || !nbr_idx_dev || !out_dev || (mode == 0 && !r_dev))
        return smrt_set_error(-1, "smrt_smote_rows_f64: null pointer");
    if (n_rows < 0 || n_x < 1 || d < 1 || ldx < d || ldo < d || (mode != 0 && mode != 1))
        return smrt_set_error(-1, "smrt_smote_rows_f64: bad sizes");
    const long total = (long)n_rows * d;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 256L * 32);
    hipLaunchKernelGGL(smote_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)hip_stream, X_dev, (long)ldx, (int)d,
                       (const long long*)base_idx_dev, (const long long*)nbr_idx_dev, r_dev, (long)n_rows, (int)mode,
                       out_dev, (long)ldo, (long)n_x);
    KNNC(hipGetLastError());
    return 0;
}
