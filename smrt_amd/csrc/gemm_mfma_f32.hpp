// LDS-staged fp32 MFMA tile GEMM for gfx950 (v_mfma_f32_32x32x2_f32, exact fp32 fma chain).
//
//   C[M,N] = epilogue( A · B ),   256 threads = 4 waves in a 2x2 arrangement,
//   block tile (64*WM) x (64*WN), K-tile 32, each wave owns WM x WN sub-tiles of 32x32.
//
// Operand layouts (no transposes are ever materialised):
//   INNER : global [mn][k]  (k contiguous)  -> LDS [mn][36]      read as one ds_read_b128 per 4 k-steps
//   OUTER : global [k][mn]  (mn contiguous) -> LDS [32][BMN+4]   read as one ds_read_b32 per k-step
// The k order inside an 8-wide k-block is permuted identically for A and B (lane half h takes
// k = 8*kb + 4*h + s at step s), which is legal because the MFMA sums over k.
//
// MFMA 32x32x2 fragment maps (cdna guide section 3): lane l: r = l & 31, h = l >> 5
//   A operand: A[m = r][k = h]     B operand: B[k = h][n = r]
//   C/D reg i: row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace smrt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Layout : int { INNER = 0, OUTER = 1 };

constexpr int GEMM_THREADS = 256;
constexpr int BK = 32;
constexpr int INNER_STRIDE = 36;     // floats; 144 B = 16 B * odd -> conflict-free ds_read_b128

template <int BMN> constexpr int outer_stride() { return BMN + 4; }
template <int LAY, int BMN> constexpr int lds_floats() {
    return LAY == INNER ? BMN * INNER_STRIDE : BK * outer_stride<BMN>();
}

// ------------------------------------------------------------------ global -> register staging
// ONES: the operand has one virtual extra mn-index (== MN_data) that reads 1.0 for k < K
// (bias-gradient row of the augmented weight matrix).
template <int LAY, int BMN, bool ONES>
struct Stage {
    static constexpr int PASSES = (LAY == INNER) ? BMN / 32 : (BK * BMN / 4) / GEMM_THREADS;
    float4 r[PASSES];

    // g: operand base; ld: leading dim (floats, % 4 == 0); mn0: tile origin along mn;
    // MN_data: number of real mn entries; k0..: tile origin along k; K: k extent.
    // full: the whole BMN x BK tile lies inside the matrix (wave-uniform) -> unconditional 16-byte loads.
    __device__ __forceinline__ void load(const float* __restrict__ g, long ld, int mn0, int MN_data,
                                         int k0, int K, int tid, bool full) {
        if (full) {
            if constexpr (LAY == INNER) {
                const float* ptr = g + (long)(mn0 + (tid >> 3)) * ld + k0 + (tid & 7) * 4;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) r[p] = *reinterpret_cast<const float4*>(ptr + (long)(32 * p) * ld);
            } else {
                constexpr int F4 = BMN / 4, RPP = GEMM_THREADS / F4;
                const float* ptr = g + (long)(k0 + tid / F4) * ld + mn0 + (tid % F4) * 4;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) r[p] = *reinterpret_cast<const float4*>(ptr + (long)(RPP * p) * ld);
            }
            return;
        }
        // Edge tiles: branch-free.  Every lane loads 16 bytes from a CLAMPED in-bounds address (row clamped to the last
        // real row, column quad clamped to the last quad of the leading dimension); mask() later replaces the
        // out-of-range components by 0 (or by 1 for the virtual ones index) with selects.  The selects must NOT sit
        // here: they would make the wave wait for the loads before the K-tile's MFMAs instead of after them.
        const int ldq = (int)ld - 4;
        if constexpr (LAY == INNER) {
            const int row = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                r[p] = *reinterpret_cast<const float4*>(g + (long)min(mn0 + row + 32 * p, MN_data - 1) * ld + min(k0 + kq, ldq));
        } else {
            constexpr int F4 = BMN / 4, RPP = GEMM_THREADS / F4;
            const int krow = tid / F4, mq = (tid % F4) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                r[p] = *reinterpret_cast<const float4*>(g + (long)min(k0 + krow + RPP * p, K - 1) * ld + min(mn0 + mq, ldq));
        }
    }

    // Edge tiles only: zero (or one) what the clamped loads fetched from outside the matrix.  Same arguments as load().
    __device__ __forceinline__ void mask(int mn0, int MN_data, int k0, int K, int tid, bool full) {
        if (full) return;
        if constexpr (LAY == INNER) {
            const int row = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int mn = mn0 + row + 32 * p, k = k0 + kq;
                const bool in = mn < MN_data;
                const float one = (ONES && mn == MN_data) ? 1.f : 0.f;
                const float4 t = r[p];
                float4 v;
                v.x = (k     < K) ? (in ? t.x : one) : 0.f;
                v.y = (k + 1 < K) ? (in ? t.y : one) : 0.f;
                v.z = (k + 2 < K) ? (in ? t.z : one) : 0.f;
                v.w = (k + 3 < K) ? (in ? t.w : one) : 0.f;
                r[p] = v;
            }
        } else {
            constexpr int F4 = BMN / 4, RPP = GEMM_THREADS / F4;
            const int krow = tid / F4, mq = (tid % F4) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int k = k0 + krow + RPP * p, mn = mn0 + mq;
                const bool kin = k < K;
                const float4 t = r[p];
                float4 v;
                v.x = kin ? (mn     < MN_data ? t.x : ((ONES && mn     == MN_data) ? 1.f : 0.f)) : 0.f;
                v.y = kin ? (mn + 1 < MN_data ? t.y : ((ONES && mn + 1 == MN_data) ? 1.f : 0.f)) : 0.f;
                v.z = kin ? (mn + 2 < MN_data ? t.z : ((ONES && mn + 2 == MN_data) ? 1.f : 0.f)) : 0.f;
                v.w = kin ? (mn + 3 < MN_data ? t.w : ((ONES && mn + 3 == MN_data) ? 1.f : 0.f)) : 0.f;
                r[p] = v;
            }
        }
    }

    // Add this thread's staged values into its running per-column sums (OUTER layout: each float4 is 4 consecutive
    // mn entries of one k row, the same 4 columns in every pass).  Out-of-range entries were staged as zeros.
    __device__ __forceinline__ void add_to(float4& cs) const {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { cs.x += r[p].x; cs.y += r[p].y; cs.z += r[p].z; cs.w += r[p].w; }
    }

    __device__ __forceinline__ void store(float* lds, int tid) const {
        if constexpr (LAY == INNER) {
            const int row = tid >> 3, kq = (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                *reinterpret_cast<float4*>(lds + (row + 32 * p) * INNER_STRIDE + kq) = r[p];
        } else {
            constexpr int F4 = BMN / 4, RPP = GEMM_THREADS / F4;
            const int krow = tid / F4, mq = (tid % F4) * 4;
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                *reinterpret_cast<float4*>(lds + (krow + RPP * p) * outer_stride<BMN>() + mq) = r[p];
        }
    }
};

// Fragment for one 8-wide k-block: f[s] feeds MFMA step s.
template <int LAY, int BMN>
__device__ __forceinline__ void read_frag(const float* lds, int base_mn, int kb, int r, int h, float (&f)[4]) {
    if constexpr (LAY == INNER) {
        const float4 v = *reinterpret_cast<const float4*>(lds + (base_mn + r) * INNER_STRIDE + kb * 8 + 4 * h);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        const float* p = lds + (kb * 8 + 4 * h) * outer_stride<BMN>() + base_mn + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = p[s * outer_stride<BMN>()];
    }
}

// ------------------------------------------------------------------ wave -> 32x32 block arrangements
// A wave owns up to four 32x32 blocks of the tile (accumulator slot s = WN * i + j).  Which ones is the tile's ARRANGEMENT,
// chosen once per workgroup (uniform):
//   0  2x2 : wave (wr, wc) owns rows 32 (WM wr + i), columns 32 (WN wc + j)            -- every interior tile
//   1  row owner (128x128 tiles only): wave w owns row block w, column blocks s = 0..   -- tiles with few real COLUMN blocks
//   2  column owner (128x128 only):    wave w owns column block w, row blocks s = 0..   -- tiles with few real ROW blocks
// In an edge tile of a 784 / 450 / 20 / 10-wide layer only MB x NB of the 4 x 4 blocks hold a real element.  Under 2x2 the
// tile lasts as long as its busiest wave, min(MB,2) min(NB,2) blocks; a row-owner tile lasts NB, a column-owner tile MB
// blocks: 66 real columns (3 blocks) take 3 block-times instead of 4, 16 real columns 1 instead of 2.
template <int WM, int WN>
__device__ __forceinline__ int blk_row(int arr, int wave, int i, int j) {
    return arr == 0 ? (wave >> 1) * 32 * WM + i * 32 : (arr == 1 ? wave * 32 : (WN * i + j) * 32);
}
template <int WM, int WN>
__device__ __forceinline__ int blk_col(int arr, int wave, int i, int j) {
    return arr == 0 ? (wave & 1) * 32 * WN + j * 32 : (arr == 1 ? (WN * i + j) * 32 : wave * 32);
}

// ------------------------------------------------------------------ block-tile main loop
// Accumulates A[m0.., kbeg..kend) * B[kbeg..kend), n0..) into acc[WM][WN].
// bsum (optional): per-thread running sums over k of the B values this thread stages (Stage::add_to) -- the
// weight-gradient kernel turns them into the bias gradient (column sums of dPre) without an extra row of MFMA tiles.
// LI x LJ: how many of this wave's 32-row / 32-column blocks are multiplied (the rest lie wholly in the padding of an edge
// tile; their accumulators stay zero).  Compile-time, so the main loop has no branch in it -- see gemm_tile below.
// ARR: the arrangement above (1: LI == 1 and LJ column blocks from the tile's left edge; 2: LJ == 1 and LI row blocks from
// the top); the product of row fragment i and column fragment j goes to accumulator slot i + j there.
template <int ALAY, int BLAY, int WM, int WN, bool A_ONES, int LI, int LJ, int ARR = 0>
__device__ __forceinline__ void gemm_tile_impl(const float* __restrict__ A, long lda, int M_data,
                                               const float* __restrict__ B, long ldb, int N_data,
                                               int m0, int n0, int kbeg, int kend,
                                               float* ldsA, float* ldsB, f32x16 (&acc)[WM][WN], float4* bsum) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Stage<ALAY, BM, A_ONES> sa;
    Stage<BLAY, BN, false> sb;
    // Interior tiles (the common case) take unconditional 16-byte loads; the bounds-checked element-wise form costs
    // ~250 scalar/branch instructions per K-tile, a third of the K-tile's MFMA time.
    const bool fullA = (m0 + BM <= M_data), fullB = (n0 + BN <= N_data);
    {
        const bool fk = (kbeg + BK <= kend);
        sa.load(A, lda, m0, M_data, kbeg, kend, tid, fullA && fk);
        sb.load(B, ldb, n0, N_data, kbeg, kend, tid, fullB && fk);
        sa.mask(m0, M_data, kbeg, kend, tid, fullA && fk);
        sb.mask(n0, N_data, kbeg, kend, tid, fullB && fk);
    }
    sa.store(ldsA, tid);
    sb.store(ldsB, tid);
    if (bsum) sb.add_to(*bsum);
    __syncthreads();

    // ldsA / ldsB each hold TWO tile images: K-tile t is computed from image t & 1 while the registers staged for
    // tile t+1 go into the other image, so one barrier per K-tile suffices (the image being overwritten was last
    // read before the previous barrier).
    constexpr int SZA = lds_floats<ALAY, BM>(), SZB = lds_floats<BLAY, BN>();
    constexpr int FI = ARR == 0 ? WM : (LI > 0 ? LI : 1), FJ = ARR == 0 ? WN : (LJ > 0 ? LJ : 1);
    const int abase = ARR == 0 ? wr * 32 * WM : (ARR == 1 ? wave * 32 : 0);
    const int bbase = ARR == 0 ? wc * 32 * WN : (ARR == 2 ? wave * 32 : 0);
    // The MFMAs of one K-tile image, NKB of its four 8-wide k-blocks (compile-time: the last K-tile of K = 450 holds two real
    // k, of K = 784 sixteen -- multiplying its zero padding was 5 % / 2 % of those layers' matrix time).
    // Fragments of k-block kb+1 are requested before the MFMAs of k-block kb; the asm (which the tile addresses
    // escape into, so its memory clobber covers them) and the scheduling barrier pin that order.  Left alone,
    // hipcc sinks each LDS read to just before its MFMA and every group of MFMAs waits out a full LDS latency.
    auto multiply = [&](auto nkb_c, const float* la, const float* lb) {
        constexpr int NKB = decltype(nkb_c)::value;
        float fa[2][FI][4], fb[2][FJ][4];
#pragma unroll
        for (int i = 0; i < FI; ++i) read_frag<ALAY, BM>(la, abase + i * 32, 0, r, h, fa[0][i]);
#pragma unroll
        for (int j = 0; j < FJ; ++j) read_frag<BLAY, BN>(lb, bbase + j * 32, 0, r, h, fb[0][j]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int cur = kb & 1, nxt = cur ^ 1;
            if (kb + 1 < NKB) {
#pragma unroll
                for (int i = 0; i < FI; ++i) read_frag<ALAY, BM>(la, abase + i * 32, kb + 1, r, h, fa[nxt][i]);
#pragma unroll
                for (int j = 0; j < FJ; ++j) read_frag<BLAY, BN>(lb, bbase + j * 32, kb + 1, r, h, fb[nxt][j]);
            }
            asm volatile("" :: "v"(ldsA), "v"(ldsB) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < LI; ++i)
#pragma unroll
                    for (int j = 0; j < LJ; ++j) {
                        f32x16& d = ARR == 0 ? acc[i][j] : acc[(i + j) / WN][(i + j) % WN];
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][s], fb[cur][j][s], d, 0, 0, 0);
                    }
        }
    };
    int buf = 0, k0 = kbeg;
    for (; k0 + BK < kend; k0 += BK, buf ^= 1) {         // every K-tile but the last: stage the next one around the MFMAs
        const bool fk = (k0 + 2 * BK <= kend);
        sa.load(A, lda, m0, M_data, k0 + BK, kend, tid, fullA && fk);
        sb.load(B, ldb, n0, N_data, k0 + BK, kend, tid, fullB && fk);
        multiply(std::integral_constant<int, BK / 8>{}, ldsA + buf * SZA, ldsB + buf * SZB);
        sa.mask(m0, M_data, k0 + BK, kend, tid, fullA && fk);
        sb.mask(n0, N_data, k0 + BK, kend, tid, fullB && fk);
        sa.store(ldsA + (buf ^ 1) * SZA, tid);
        sb.store(ldsB + (buf ^ 1) * SZB, tid);
        if (bsum) sb.add_to(*bsum);
        __syncthreads();
    }
    {                                                    // the last K-tile: only its k-blocks that hold a real k (uniform)
        const int nkb = (kend - k0 + 7) >> 3;
        const float* la = ldsA + buf * SZA;
        const float* lb = ldsB + buf * SZB;
        if (nkb == 1)      multiply(std::integral_constant<int, 1>{}, la, lb);
        else if (nkb == 2) multiply(std::integral_constant<int, 2>{}, la, lb);
        else               multiply(std::integral_constant<int, BK / 8>{}, la, lb);
    }
}

// Edge tiles of 784 / 450 / 20 / 10-wide layers are mostly padding (16 real columns in the last 128 of 784): a wave
// multiplies only the blocks of its region that hold a real element, and a 128x128 edge tile takes the arrangement that
// spreads its real blocks evenly over the four waves (returned: the epilogue maps accumulators through blk_row / blk_col).
// The choice is made ONCE per tile, between complete copies of the main loop (a branch around the MFMA group inside one loop makes hipcc spill the accumulators); it is
// wave-uniform, and every copy passes the same barriers, so the waves of a workgroup may take different copies.
template <int ALAY, int BLAY, int WM, int WN, bool A_ONES>
__device__ __forceinline__ int gemm_tile(const float* __restrict__ A, long lda, int M_data,
                                         const float* __restrict__ B, long ldb, int N_data,
                                         int m0, int n0, int kbeg, int kend,
                                         float* ldsA, float* ldsB, f32x16 (&acc)[WM][WN], float4* bsum = nullptr) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wr = wave >> 1, wc = wave & 1;
#define SMRT_TILE(LI_, LJ_, ARR_) gemm_tile_impl<ALAY, BLAY, WM, WN, A_ONES, LI_, LJ_, ARR_>(A, lda, M_data, B, ldb, N_data, m0, n0, kbeg, kend, ldsA, ldsB, acc, bsum)
    if constexpr (WM == 2 && WN == 2) {
        // real 32-row / 32-column blocks of this tile, and the arrangement under which its busiest wave has the fewest
        const int MB = min(4, (M_data + (A_ONES ? 1 : 0) - m0 + 31) >> 5), NB = min(4, (N_data - n0 + 31) >> 5);
        const int c0 = min(MB, 2) * min(NB, 2);
        if (NB < c0 && NB <= MB) {                  // row owners: wave w multiplies row block w by column blocks 0 .. NB-1
            if (wave >= MB) SMRT_TILE(0, 0, 0);
            else if (NB == 1) SMRT_TILE(1, 1, 1);
            else if (NB == 2) SMRT_TILE(1, 2, 1);
            else SMRT_TILE(1, 3, 1);
            return 1;
        }
        if (MB < c0) {                              // column owners: wave w multiplies row blocks 0 .. MB-1 by column block w
            if (wave >= NB) SMRT_TILE(0, 0, 0);
            else if (MB == 1) SMRT_TILE(1, 1, 2);
            else if (MB == 2) SMRT_TILE(2, 1, 2);
            else SMRT_TILE(3, 1, 2);
            return 2;
        }
    }
    const int rem_m = M_data + (A_ONES ? 1 : 0) - (m0 + wr * 32 * WM), rem_n = N_data - (n0 + wc * 32 * WN);
    const int li = __builtin_amdgcn_readfirstlane(rem_m <= 0 ? 0 : min(WM, (rem_m + 31) >> 5));
    const int lj = __builtin_amdgcn_readfirstlane(rem_n <= 0 ? 0 : min(WN, (rem_n + 31) >> 5));
    if (li == WM && lj == WN) { SMRT_TILE(WM, WN, 0); return 0; }
    if constexpr (WM == 2) { if (li == 2 && lj == 1) { SMRT_TILE(2, 1, 0); return 0; } }
    if constexpr (WN == 2) { if (li == 1 && lj == 2) { SMRT_TILE(1, 2, 0); return 0; } }
    if constexpr (WM == 2 || WN == 2) { if (li == 1 && lj == 1) { SMRT_TILE(1, 1, 0); return 0; } }
    SMRT_TILE(0, 0, 0);         // nothing of this wave's region is real: it still stages its share of both operands
#undef SMRT_TILE
    return 0;
}

// Visit every accumulator element of this thread: f(row, col, value) with global row/col (`arr`: what gemm_tile returned).
template <int WM, int WN, class F>
__device__ __forceinline__ void for_each_c(const f32x16 (&acc)[WM][WN], int m0, int n0, int arr, F&& f) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int rb = m0 + blk_row<WM, WN>(arr, wave, i, j), cb = n0 + blk_col<WM, WN>(arr, wave, i, j);
#pragma unroll
            for (int e = 0; e < 16; ++e) f(rb + (e & 3) + 8 * (e >> 2) + 4 * h, cb + r, acc[i][j][e]);
        }
}

}  // namespace smrt
