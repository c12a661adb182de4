// Dense-layer GEMMs for SMALL mini-batches (M = rows <= SMALL_M_MAX), e.g. the reference's own configuration:
// 784-450-(10,10)-450-784 at batch 128 / 256 (smrt/balance/tests/test_smrt.py:17-21).
//
// At these sizes the LDS-tiled kernel (gemm_mfma_f32.hpp) is latency-bound twice over: 256 x 450 outputs are 32 of
// its 64x64 tiles (32 of 256 CUs busy), and each workgroup walks K = 784 as 25 dependent
// global-load -> LDS -> barrier -> MFMA rounds of ~1 us.  Here instead
//   * a workgroup owns ONE small output tile and its four waves split K between them: wave w takes the 16-wide k-blocks
//     with index = w (mod 4), so the dependent MFMA chain is 4x shorter;
//   * each wave loads its operand fragments straight from global memory into registers in MFMA layout -- no LDS
//     staging, no barrier in the main loop -- a batch of k-blocks at a time, the next batch requested before the
//     current one is multiplied;
//   * the four partial tiles meet in LDS once, and the threads run the fused epilogue on 4 outputs each with 16-byte
//     stores.
// What bounds such a launch is the rate at which ONE CU can pull its tile's operand panels through its L1 (25-70 GB/s
// per CU: the panels were rewritten by the previous launch on other XCDs), so the lever is the NUMBER of CUs that pull
// at once: the tile is (16 MT) x (16 NT), MT, NT in {1, 2}, on v_mfma_f32_16x16x4_f32, picked per launch so that the grid
// comes near the chip's 256 CUs -- 128 x 450 outputs are 60 tiles of 32x32 (200 KB of panels each), 120 of 16x32, 232 of
// 16x16 (100 KB each).  Several classes trained side by side (grouped launches) fill the chip with 32x32 tiles, which
// move the fewest bytes.  Every shape sums a wave's k-blocks in the same order and the four partial tiles the same way,
// so a result does not depend on the shape that produced it (bit-identical; tests/test_gpu_api.py).
//
// MFMA v_mfma_f32_16x16x4_f32 maps: lane l: i = l & 15, q = l >> 4; A operand A[m = i][k = q], B operand B[k = q][n = i];
// C/D reg r: row = 4 q + r, col = i.  Inside a 16-wide k-block lane group q takes k = 4 q + s at step s for BOTH operands
// (the MFMA sums over k), so a k-contiguous operand is one 16-byte load per k-block.  An n-contiguous B operand
// (weights [k][n]) is loaded NT floats per lane (lane i: columns n0 + NT i ..), and accumulator nt holds the columns
// n0 + NT i + nt: 16 lanes x NT x 4 bytes contiguous per k row.
// (The weight-gradient launch keeps its 32x32 tiles on v_mfma_f32_32x32x2_f32: both its operands are mn-contiguous,
// K is only the mini-batch, and it has 780 tiles already.)
//
// Reference arithmetic: as vae_kernels.hip (layer.py:365-368, _ae_utils.py:15-49, base.py:127-149).
#include "vae_kernels.hpp"
#include "device_utils.hpp"
#include "gemm_mfma_f32.hpp"

#include <algorithm>
#include <cstdlib>

namespace smrt {

// diagnostic builds (make stamps): phase time stamps (100 MHz wall clock) of the first workgroup and of one from the middle
#ifdef SMRT_STAMPS
#define SST(i) do { if (a.dbg && threadIdx.x == 0) { const unsigned b_ = blockIdx.y * gridDim.x + blockIdx.x, n_ = gridDim.x * gridDim.y; \
    if (b_ == 0) a.dbg[i] = wall_clock64(); else if (b_ == n_ / 2) a.dbg[16 + (i)] = wall_clock64(); else if (b_ == n_ - 1) a.dbg[32 + (i)] = wall_clock64(); } } while (0)
#else
#define SST(i) do { } while (0)
#endif

constexpr int ST = 32;                 // output tile edge
constexpr int SRED = 33;               // LDS row stride of a partial tile

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ 32x32 weight-gradient tiles on v_mfma_f32_32x32x2_f32
// Both operands are mn-contiguous ([k][m] layer inputs, [k][n] gradients; k = row of the mini-batch): per 8-wide k-block a
// lane takes four 4-byte loads per operand (k = block + 4 h + s at step s), each 128 B contiguous per half-wave.  The loads
// are BUFFER loads: per-lane offset (4 h rows + the lane's column: one register per operand for the whole tile) + a uniform
// offset per (block, step) in an SGPR, and rows past the mini-batch read as zero (the descriptor ends with the last row), so
// a ragged K needs neither clamps nor masks.  What still needs a mask is an edge tile: columns past the matrix, and the
// virtual ones index of the bias gradient.
template <bool ONES>
__device__ __forceinline__ f32x4v mask_kblock(f32x4v t, int mn, int MN_data, int kb4, int K) {
    const bool in = mn < MN_data;
    const float fill = (ONES && mn == MN_data) ? 1.f : 0.f;
    f32x4v v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (kb4 + s < K) ? (in ? t[s] : fill) : 0.f;
    return v;
}

// One 32x32 tile of A^T.B over k in [0, K), A = [K][lda] (+ a virtual all-ones column at index M_data when A_ONES), B = [K][ldb]:
// returns this thread's four reduced outputs (row = tid >> 3, columns 4 * (tid & 7) .. + 3 of the tile).
// `red` = 4 * 32 * SRED floats of LDS.  SG 8-wide k-blocks per wave are in flight at a time (ONE batch: K up to 32 SG rows is a
// single round trip), without a second buffer: the weight-gradient launch wants four workgroups a CU more than overlap.
template <bool A_ONES, int SG>
__device__ __forceinline__ f32x4v small_tile(const float* __restrict__ A, long lda, int M_data,
                                             const float* __restrict__ B, long ldb, int N_data,
                                             int m0, int n0, int K, float* red) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nkb = (K + 7) >> 3;                       // 8-wide k-blocks in all; this wave takes wave, wave + 4, ...
    const int mine = (nkb - wave + 3) >> 2;             // how many of them are this wave's (may be 0)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const unsigned lda4 = (unsigned)lda * 4u, ldb4 = (unsigned)ldb * 4u;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (unsigned)K * lda4), rb = make_rsrc(B, (unsigned)K * ldb4);
    const unsigned va = 4u * (unsigned)h * lda4 + 4u * (unsigned)min(m0 + r, M_data - 1);
    const unsigned vb = 4u * (unsigned)h * ldb4 + 4u * (unsigned)min(n0 + r, N_data - 1);

    f32x4v fa[SG], fb[SG];
    // interior tiles need no masking at all (wave-uniform); rows past K arrive as zeros
    const bool clean = (m0 + ST <= M_data) && (n0 + ST <= N_data);
#pragma unroll 1
    for (int i0 = 0; i0 < mine; i0 += SG) {
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            const unsigned row0 = (unsigned)(((i0 + j) * 4 + wave) * 8);      // (uniform) blocks past the end read zeros
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, va, (row0 + s) * lda4, 0));
                fb[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vb, (row0 + s) * ldb4, 0));
            }
        }
        if (!clean) {
#pragma unroll
            for (int j = 0; j < SG; ++j) {
                const int kb4 = ((i0 + j) * 4 + wave) * 8 + 4 * h;
                fa[j] = mask_kblock<A_ONES>(fa[j], m0 + r, M_data, kb4, K);
                fb[j] = mask_kblock<false>(fb[j], n0 + r, N_data, kb4, K);
            }
        }
        const int cnt = min(SG, mine - i0);
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            if (j < cnt) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][s], fb[j][s], acc, 0, 0, 0);
            }
        }
    }
    // the four waves' partial tiles meet in LDS
    float* mine_red = red + wave * (ST * SRED);
#pragma unroll
    for (int e = 0; e < 16; ++e) mine_red[((e & 3) + 8 * (e >> 2) + 4 * h) * SRED + r] = acc[e];
    __syncthreads();
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    f32x4v out;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* p = red + row * SRED + c4 + c;
        out[c] = (p[0] + p[ST * SRED]) + (p[2 * ST * SRED] + p[3 * ST * SRED]);
    }
    return out;
}

// ------------------------------------------------------------------ (16 MT) x (16 NT) tiles on v_mfma_f32_16x16x4_f32
template <int MT, int NT> struct TS {
    static constexpr int TM = 16 * MT, TN = 16 * NT;
    static constexpr int TPR = TN / 4;                  // epilogue threads per tile row (4 outputs each)
    static constexpr int ACTIVE = TM * TPR;             // epilogue threads of a tile: 64 / 128 / 256
    static constexpr int RS = TN + 1;                   // LDS row stride of a partial tile
    static constexpr int SG = (MT + NT <= 2) ? 8 : 4;   // 16-wide k-blocks per wave per batch
    static constexpr int RED = 4 * TM * RS;             // floats of LDS for the four waves' partial tiles
};
typedef float f32x2v __attribute__((ext_vector_type(2)));

// One 16-wide k-block of this lane's operand fragments: a[mt][s] = A[row 16 mt + i][k4 + s], b[nt][s] = B[k4 + s][column of
// accumulator nt], k4 = block start + 4 q.
template <int MT, int NT> struct KFrag { f32x4v a[MT]; float b[NT][4]; };

// Loads from clamped, always-valid addresses; mask16() zeroes what lies outside the matrices LATER, when the values are
// about to be used (a select right behind the load would make the wave wait for its prefetch before multiplying the batch
// it already has).  A is always k-contiguous here ([m][k]); B is k-contiguous (INNER, [n][k]) or n-contiguous (OUTER, [k][n]).
template <int BLAY, int MT, int NT>
__device__ __forceinline__ void fetch16(KFrag<MT, NT>& f, const float* __restrict__ A, long lda, int M_data,
                                        const float* __restrict__ B, long ldb, int N_data, int m0, int n0, int k4, int K, int i) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        f.a[mt] = *reinterpret_cast<const f32x4v*>(A + (long)min(m0 + 16 * mt + i, M_data - 1) * lda + min(k4, (int)lda - 4));
    if constexpr (BLAY == INNER) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4v t = *reinterpret_cast<const f32x4v*>(B + (long)min(n0 + 16 * nt + i, N_data - 1) * ldb + min(k4, (int)ldb - 4));
#pragma unroll
            for (int s = 0; s < 4; ++s) f.b[nt][s] = t[s];
        }
    } else {
        const float* p = B + min(n0 + NT * i, (int)ldb - NT);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long r = (long)min(k4 + s, K - 1) * ldb;
            if constexpr (NT == 1) f.b[0][s] = p[r];
            else {
                const f32x2v t = *reinterpret_cast<const f32x2v*>(p + r);
                f.b[0][s] = t[0]; f.b[1][s] = t[1];
            }
        }
    }
}
template <int BLAY, int MT, int NT>
__device__ __forceinline__ void mask16(KFrag<MT, NT>& f, int M_data, int N_data, int m0, int n0, int k4, int K, int i) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const bool in = (m0 + 16 * mt + i) < M_data;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.a[mt][s] = (in && k4 + s < K) ? f.a[mt][s] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (BLAY == INNER) ? n0 + 16 * nt + i : n0 + NT * i + nt;
        const bool in = col < N_data;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.b[nt][s] = (in && k4 + s < K) ? f.b[nt][s] : 0.f;
    }
}

// One (16 MT) x (16 NT) tile of A.B over k in [0, K): returns this thread's four reduced outputs -- row = tid / TPR, columns
// 4 (tid % TPR) .. + 3 of the tile; meaningful for tid < ACTIVE.  `red` = TS<MT, NT>::RED floats of LDS.
struct NoOverlap { __device__ __forceinline__ void operator()() const {} };
// `overlap` runs between the request for the first batch of fragments and its first use: independent arithmetic put there
// (the reparameterisation's Philox draw) hides under the fetch, which is most of a small tile's time.
template <int BLAY, int MT, int NT, class Overlap = NoOverlap>
__device__ __forceinline__ f32x4v small_tile16(const float* __restrict__ A, long lda, int M_data,
                                               const float* __restrict__ B, long ldb, int N_data,
                                               int m0, int n0, int K, float* red, Overlap overlap = Overlap()) {
    using T = TS<MT, NT>;
    constexpr int SG16 = T::SG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int nkb = (K + 15) >> 4;                      // 16-wide k-blocks in all; this wave takes wave, wave + 4, ...
    const int mine = (nkb - wave + 3) >> 2;             // how many of them are this wave's (may be 0)
    f32x4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};

    KFrag<MT, NT> cur[SG16], nxt[SG16];
    auto fetch = [&](int first, KFrag<MT, NT> (&x)[SG16]) {
#pragma unroll
        for (int j = 0; j < SG16; ++j)                  // (blocks past the end read a clamped address)
            fetch16<BLAY, MT, NT>(x[j], A, lda, M_data, B, ldb, N_data, m0, n0, ((first + j) * 4 + wave) * 16 + 4 * q, K, i);
    };
    // Masking (all wave-uniform): interior tiles need none for rows / columns, and a ragged K only shows in the LAST k-block,
    // i.e. in one batch of one wave.
    const bool edge_mn = (m0 + T::TM > M_data) || (n0 + T::TN > N_data);
    const int last_kb = ((K & 15) != 0 && ((nkb - 1) & 3) == wave) ? (nkb - 1) >> 2 : -1;     // index among this wave's blocks
    auto mask = [&](int first, KFrag<MT, NT> (&x)[SG16]) {
        if (!edge_mn && !(last_kb >= first && last_kb < first + SG16)) return;
#pragma unroll
        for (int j = 0; j < SG16; ++j)
            mask16<BLAY, MT, NT>(x[j], M_data, N_data, m0, n0, ((first + j) * 4 + wave) * 16 + 4 * q, K, i);
    };
    if (mine > 0) fetch(0, cur);
    __builtin_amdgcn_sched_barrier(0);
    overlap();
    __builtin_amdgcn_sched_barrier(0);
    if (mine > 0) mask(0, cur);
#pragma unroll 1
    for (int b0 = 0; b0 < mine; b0 += SG16) {
        const bool more = b0 + SG16 < mine;
        if (more) fetch(b0 + SG16, nxt);
        asm volatile("" ::: "memory");                   // next batch is requested before this one is multiplied
        __builtin_amdgcn_sched_barrier(0);
        const int cnt = min(SG16, mine - b0);
#pragma unroll
        for (int j = 0; j < SG16; ++j) {
            if (j < cnt) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j].a[mt][s], cur[j].b[nt][s], acc[mt][nt], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < SG16; ++j) cur[j] = nxt[j];
            mask(b0 + SG16, cur);
        }
    }
    // the four waves' partial tiles meet in LDS
    float* mine_red = red + wave * (T::TM * T::RS);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = (BLAY == INNER) ? 16 * nt + i : NT * i + nt;
#pragma unroll
            for (int r = 0; r < 4; ++r) mine_red[(16 * mt + 4 * q + r) * T::RS + col] = acc[mt][nt][r];
        }
    __syncthreads();
    const int row = tid / T::TPR, c4 = (tid % T::TPR) * 4;
    f32x4v out = {0.f, 0.f, 0.f, 0.f};
    if (tid < T::ACTIVE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* p = red + row * T::RS + c4 + c;
            out[c] = (p[0] + p[T::TM * T::RS]) + (p[2 * T::TM * T::RS] + p[3 * T::TM * T::RS]);
        }
    }
    return out;
}

// Tile shape of a small-batch launch over an M x N output: the largest of 32x32 / 16x32 / 16x16 whose grid reaches
// SMALL_TILE_FILL workgroups, counting the other classes' grids when several are recorded side by side (grouped launches);
// else 16x16.  SMRT_HIP_SMALL_TILE = 22 | 12 | 11 forces one (measurements, and the test that all shapes agree bit for bit).
constexpr int SMALL_TILE_FILL = 200;
static int forced_small_tile() {
    const char* e = std::getenv("SMRT_HIP_SMALL_TILE");               // (read per call: tests vary it)
    return (e && *e) ? std::atoi(e) : 0;
}
static int forced_mid_tile() {
    const char* e = std::getenv("SMRT_HIP_MID_TILE");
    return (e && *e) ? std::atoi(e) : 0;
}
int small_tile_shape(int M, int N) {
    const int f = forced_small_tile();
    if (f == 22 || f == 12 || f == 11) return f;
    const int group = g_record.out ? std::max(1, g_record.group) : 1;
    const char* fe = std::getenv("SMRT_HIP_TILE_FILL");
    const int fill = (fe && *fe) ? std::atoi(fe) : SMALL_TILE_FILL;
    auto tiles = [&](int tm, int tn) { return ((M + tm - 1) / tm) * ((N + tn - 1) / tn) * group; };
    if (tiles(32, 32) >= fill) return 22;
    if (tiles(16, 32) >= fill) return 12;
    return 11;
}
static inline int shape_tm(int shape) { return shape == 22 ? 32 : 16; }
static inline int shape_tn(int shape) { return shape == 11 ? 16 : 32; }

// Workgroup -> tile, XCD-aware (speed only, never correctness): the dispatcher deals workgroups round-robin over the 8 XCDs,
// each with an L2 of its own that starts every launch cold (the operands were rewritten by the previous launch), so
// workgroup b shares its L2 with b + 8, b + 16, ...  XCD x takes the x-th contiguous chunk of the tile list (bijective for
// any grid); the list runs along the dimension whose operand is SMALLER, so that a chunk re-reads a few panels of the
// larger operand and all of the smaller one: 128 x 450 outputs over K = 784 pull 0.65 MB into each L2 instead of 1.8 MB.
__device__ __forceinline__ void xcd_tile(int gx, int gy, int mode, int& tx, int& ty) {
    if (mode == 0) { tx = blockIdx.x; ty = blockIdx.y; return; }
    const int T = gx * gy, b = blockIdx.y * gx + blockIdx.x;
    const int x = b & 7, slot = b >> 3, base = T >> 3, rem = T & 7;
    const int t = x * base + min(x, rem) + slot;
    if (mode == 1) { tx = t / gy; ty = t - tx * gy; }      // the list runs down the columns of tiles
    else           { ty = t / gx; tx = t - ty * gx; }      // ... along the rows
}
// 0: tiles as dispatched; 1 / 2: XCD chunks of the column- / row-major tile list.  SMRT_HIP_XCD_MAP overrides (measurements).
static int pick_xmap(int M, int N) {
    const char* e = std::getenv("SMRT_HIP_XCD_MAP");
    if (e && *e) return std::atoi(e);
    return N >= M ? 1 : 2;
}

// ------------------------------------------------------------------ dense layer forward / backward
// (body shared by the plain kernel and its grouped twin; gx = columns of tiles of THIS problem's grid)
template <int BLAY, int EPI, int MT, int NT>
__device__ __forceinline__ void small_gemm_body(const GemmArgs& a, int tx, int ty) {
    using T = TS<MT, NT>;
    __shared__ float red[T::RED];
    __shared__ float lsub[(EPI == EPI_BCE || EPI == EPI_MSE) ? T::ACTIVE : 1];
    const int tid = threadIdx.x;
    const int m0 = ty * T::TM, n0 = tx * T::TN;
    const bool active = tid < T::ACTIVE;
    const int row = m0 + tid / T::TPR, col0 = n0 + (tid % T::TPR) * 4;
    // ---- epilogue operands (bias, targets, producing layer's output, heads + epsilon): requested NOW, unconditionally and
    // from clamped in-bounds addresses, so that they arrive under the main loop.  Loaded where they are used -- each under
    // its own bounds check -- the compiler waits for every one of them separately: four to twelve dependent memory round
    // trips behind the GEMM of a kernel that is latency-bound to begin with.
    const int rowc = min(row, a.M - 1);
    float pb[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {0.f, 0.f, 0.f, 0.f}, px2[4] = {0.f, 0.f, 0.f, 0.f}, pe[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int colc = min(col0 + c, a.N - 1);
        if constexpr (EPI == EPI_BIAS_ACT || EPI == EPI_HEADS_SPLIT || EPI == EPI_BCE || EPI == EPI_MSE) pb[c] = a.bias[colc];
        if constexpr (EPI == EPI_BCE || EPI == EPI_MSE || EPI == EPI_ACTGRAD || EPI == EPI_LATENT)
            px[c] = a.aux[(long)rowc * a.ldaux + colc];
        if constexpr (EPI == EPI_LATENT) {
            px2[c] = a.aux[(long)rowc * a.ldaux + a.Z + colc];
            pe[c] = a.aux2[(long)rowc * a.ldaux2 + colc];
        }
    }
    SST(0);
    const f32x4v v = small_tile16<BLAY, MT, NT>(a.A, a.lda, a.M, a.B, a.ldb, a.N, m0, n0, a.K, red);
    SST(1);
    const bool rin = active && row < a.M;
    float lsum = 0.f;
    float o[4], o2[4];
    bool cin[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { cin[c] = rin && (col0 + c) < a.N; o[c] = 0.f; o2[c] = 0.f; }

    if constexpr (EPI == EPI_BIAS_ACT || EPI == EPI_HEADS_SPLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) o[c] = act_fwd(a.act, v[c] + pb[c]);      // layer.py:368
    } else if constexpr (EPI == EPI_BCE) {
        // sigmoid output + clipped Bernoulli NLL + dL/dpre (layer.py:333-338, _ae_utils.py:31-32)
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) {
            const float p = fast_sigmoid(v[c] + pb[c]);
            const float x = px[c];
            const float pl = tf_maximum(p, a.lo), pc = tf_minimum(pl, a.hi), om = 1.0f - pc;
            lsum -= x * __logf(pc) + (1.0f - x) * __logf(om);
            const bool pass = (p >= a.lo) && (pl <= a.hi);
            const float dp = pass ? -(x * __builtin_amdgcn_rcpf(pc) - (1.0f - x) * __builtin_amdgcn_rcpf(om)) * a.inv_batch : 0.0f;
            o[c] = dp * p * (1.0f - p);
            o2[c] = p;
        }
    } else if constexpr (EPI == EPI_MSE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) {
            const float y = act_fwd(a.act, v[c] + pb[c]);
            const float d = y - px[c];
            lsum += d * d;
            o[c] = act_bwd(a.act, y, 2.0f * d * a.inv_batch);
        }
    } else if constexpr (EPI == EPI_ACTGRAD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) o[c] = act_bwd(a.act, px[c], v[c]);
    } else if constexpr (EPI == EPI_LATENT) {
        // v = dL/dz -> both heads: dmu = dz + mu/B ; dls = dz*eps*exp(ls) + (exp(2 ls) - 1)/B   (layer.py:279, _ae_utils.py:49)
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) {
            const float mu = px[c];
            const float ls = px2[c];
            const float e = pe[c];
            const float el = expf(ls);
            o[c] = act_bwd(a.act, mu, v[c] + mu * a.inv_batch);
            o2[c] = act_bwd(a.act, ls, v[c] * e * el + (expf(2.0f * ls) - 1.0f) * a.inv_batch);
        }
    }

    // ---- stores: 16 bytes per lane where the four columns are real and the destination quad is aligned
    if constexpr (EPI == EPI_HEADS_SPLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (cin[c]) {
            const int col = col0 + c;
            if (col < a.Z) a.C[(long)row * a.Z + col] = o[c];
            else           a.out2[(long)row * a.Z + (col - a.Z)] = o[c];
        }
    } else {
        float* dst = a.C + (long)row * a.ldc + col0;
        if (cin[3] && ((a.ldc & 3) == 0)) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (cin[c]) dst[c] = o[c];
        }
        if constexpr (EPI == EPI_LATENT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (cin[c]) a.C[(long)row * a.ldc + a.Z + col0 + c] = o2[c];
        }
        if constexpr (EPI == EPI_BCE) {
            if (a.out2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (cin[c]) a.out2[(long)row * a.ldout2 + col0 + c] = o2[c];
            }
        }
    }
    if constexpr (EPI == EPI_BCE || EPI == EPI_MSE) {
        // Loss partials are per 16x16 SUB-tile whatever the tile shape, each summed the same way (a thread's four columns,
        // then wave `w` = sub-tile w: lane l takes row l / 4, column quad l % 4, shuffle tree), and finalize adds them in
        // sub-tile order: the cost, too, is the same bits for every shape.
        if (active) lsub[tid] = lsum;                   // == [row in tile][column quad]
        __syncthreads();
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
        if (w < MT * NT) {
            const int mt = w / NT, nt = w % NT;
            const float tot = wave_sum(lsub[(16 * mt + (l >> 2)) * T::TPR + 4 * nt + (l & 3)]);
            const int sr = ty * MT + mt, sc = tx * NT + nt, gx16 = (a.N + 15) >> 4;
            if (l == 0 && sr * 16 < a.M && sc < gx16) a.partials[sr * gx16 + sc] = tot;
        }
    }
}

template <int BLAY, int EPI, int MT, int NT>
__global__ __launch_bounds__(256, 2) void small_gemm_kernel(const GemmArgs a) {
    int tx, ty;
    xcd_tile((int)gridDim.x, (int)gridDim.y, a.xmap, tx, ty);
    small_gemm_body<BLAY, EPI, MT, NT>(a, tx, ty);
}

// grouped twin: gridDim.z problems (one per handle), each with its own arguments and grid extent in a device table
template <int BLAY, int EPI, int MT, int NT>
__global__ __launch_bounds__(256, 2) void small_gemm_multi(const MultiSlot<GemmArgs>* __restrict__ t) {
    const MultiSlot<GemmArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    small_gemm_body<BLAY, EPI, MT, NT>(s.a, (int)blockIdx.x, (int)blockIdx.y);
}
#define SMRT_REG_SMALL1(BL, EP, MT, NT) static MultiReg<GemmArgs> reg_small_##BL##EP##MT##NT(small_gemm_kernel<BL, EP, MT, NT>, small_gemm_multi<BL, EP, MT, NT>)
#define SMRT_REG_SMALL(BL, EP) SMRT_REG_SMALL1(BL, EP, 1, 1); SMRT_REG_SMALL1(BL, EP, 1, 2); SMRT_REG_SMALL1(BL, EP, 2, 2)
SMRT_REG_SMALL(OUTER, EPI_BIAS_ACT); SMRT_REG_SMALL(OUTER, EPI_BCE); SMRT_REG_SMALL(OUTER, EPI_MSE);
SMRT_REG_SMALL(OUTER, EPI_HEADS_SPLIT); SMRT_REG_SMALL(INNER, EPI_ACTGRAD); SMRT_REG_SMALL(INNER, EPI_LATENT);
#undef SMRT_REG_SMALL
#undef SMRT_REG_SMALL1

// number of loss partials an output-layer launch writes: one per 16x16 sub-tile, whatever the tile shape
int small_gemm_blocks(const GemmArgs& a) { return ((a.M + 15) / 16) * ((a.N + 15) / 16); }

template <int BLAY, int EPI>
static void launch_small_t(hipStream_t s, const GemmArgs& a0) {
    GemmArgs a = a0;
    a.xmap = pick_xmap(a.M, a.N);
    const int sh = small_tile_shape(a.M, a.N);
    const dim3 grid((a.N + shape_tn(sh) - 1) / shape_tn(sh), (a.M + shape_tm(sh) - 1) / shape_tm(sh));
    if (sh == 22)      SMRT_LAUNCH((small_gemm_kernel<BLAY, EPI, 2, 2>), grid, dim3(256), 0, s, a);
    else if (sh == 12) SMRT_LAUNCH((small_gemm_kernel<BLAY, EPI, 1, 2>), grid, dim3(256), 0, s, a);
    else               SMRT_LAUNCH((small_gemm_kernel<BLAY, EPI, 1, 1>), grid, dim3(256), 0, s, a);
}

void launch_small_gemm(hipStream_t s, int kind, int epi, const GemmArgs& a) {
    if (kind == GEMM_NN) {
        if (epi == EPI_BIAS_ACT)         launch_small_t<OUTER, EPI_BIAS_ACT>(s, a);
        else if (epi == EPI_BCE)         launch_small_t<OUTER, EPI_BCE>(s, a);
        else if (epi == EPI_MSE)         launch_small_t<OUTER, EPI_MSE>(s, a);
        else                             launch_small_t<OUTER, EPI_HEADS_SPLIT>(s, a);
    } else {
        if (epi == EPI_ACTGRAD)          launch_small_t<INNER, EPI_ACTGRAD>(s, a);
        else                             launch_small_t<INNER, EPI_LATENT>(s, a);
    }
}

// ------------------------------------------------------------------ the middle of the VAE, fused
// At these sizes every launch costs ~4 us before it computes anything (its arguments, its operands and the step scalars
// are three dependent memory round trips: the bare reparameterisation, 1 280 elements, takes 4.5 us), and the three
// launches around the latent layer -- heads (2 Z <= 32 columns: four workgroups), reparameterisation, first decoder layer
// (K = Z) -- are nothing but that floor, three times: 20 us of a 73 us step of the reference's own net at batch 128.
// Here every workgroup of the first decoder layer's grid recomputes the heads tile of its rows (one (16 MT) x 32 tile over
// K = h_L: what any workgroup of the neighbouring layers does), reparameterises into LDS, and multiplies its own 16 NT
// output columns from there; the workgroups of the first column also write the heads, z, epsilon and the KL partial.
// Backward the same way: dz tile (K = h_L) -> latent gradients in LDS -> own columns of the encoder's dPre (K = 2 Z).
// Same arithmetic in the same order as the separate kernels (small_tile16, the same epilogues): results are bit-identical.
constexpr int MID_LD = 36;             // LDS row stride of the latent tiles (2 Z <= 32 columns + slack for clamped reads)

template <int MT, int NT>
__device__ __forceinline__ void mid_fwd_body(const MidArgs& a, int tx, int ty) {
    using TH = TS<MT, 2>;              // the heads tile: (16 MT) x 32
    using T = TS<MT, NT>;              // this workgroup's tile of the first decoder layer
    __shared__ float red[TH::RED];
    __shared__ __align__(16) float ml[TH::TM * MID_LD];
    __shared__ __align__(16) float zt[TH::TM * MID_LD];
    __shared__ float klt[TH::TM * 16];                 // KL terms [row][k], Z <= 16
    const int tid = threadIdx.x;
    const int m0 = ty * T::TM, n0 = tx * T::TN;
    const int lr = tid >> 3, c4 = (tid & 7) * 4, row = m0 + lr;                    // heads tile: 8 threads per row
    const int lr2 = tid / T::TPR, c42 = (tid % T::TPR) * 4, row2 = m0 + lr2;      // decoder tile
    const bool act1 = tid < TH::ACTIVE, act2 = tid < T::ACTIVE;
    const int Z = a.Z, N2 = 2 * a.Z;
    const bool first = tx == 0;
    float pb[4], pd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        pb[c] = a.bh[min(c4 + c, N2 - 1)];
        pd[c] = a.bd[min(n0 + c42 + c, a.hL - 1)];
    }
    // epsilon of this thread's reparameterisation elements -- up to 2 per thread (32 Z <= 512): injected values are requested
    // now; the Philox draw (indexed by step and row, ~150 instructions per element) runs while the heads tile's operands are
    // on their way
    float pe[2] = {0.f, 0.f};
    if (a.eps_in) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            const int r = e / Z, k = e - r * Z;
            if (e < T::TM * Z) pe[u] = a.eps_in[(long)min(m0 + r, a.M - 1) * Z + k];
        }
    }
    const uint32_t step = a.st->step;
    SST(0);
    auto draw = [&]() {
        if (a.eps_in) return;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            const int r = e / Z, k = e - r * Z;
            if (e < T::TM * Z) pe[u] = philox_normal1(a.seed, STREAM_EPS, step, (uint64_t)(a.row0_global + m0 + r) * Z + k);
        }
    };
    // ---- heads: act(h . [W_mu | W_ls] + b)                                        layer.py:312-317
    const f32x4v v = small_tile16<OUTER, MT, 2>(a.H, a.ldh, a.M, a.Wh, a.ldwh, N2, m0, 0, a.hL, red, draw);
    SST(1);
    if (act1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = c4 + c;
            const bool in = row < a.M && col < N2;
            const float y = in ? act_fwd(a.act, v[c] + pb[c]) : 0.f;
            ml[lr * MID_LD + col] = y;
            if (first && in) a.ML[(long)row * a.ldml + col] = y;
        }
    }
    __syncthreads();
    // ---- z = mu + eps * exp(log_sigma), KL row terms                              layer.py:279, _ae_utils.py:49
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + u * 256;
        if (e < T::TM * Z) {
            const int r = e / Z, k = e - r * Z, grow = m0 + r;
            float z = 0.f, klv = 0.f;
            if (grow < a.M) {
                const float mu = ml[r * MID_LD + k], ls = ml[r * MID_LD + Z + k];
                const float ee = pe[u];
                z = mu + ee * expf(ls);
                if (first) {
                    a.Zb[(long)grow * a.ldz + k] = z;
                    a.Eb[(long)grow * a.ldz + k] = ee;
                    klv = -0.5f * (1.0f + 2.0f * ls - mu * mu - expf(2.0f * ls));
                }
            }
            zt[r * MID_LD + k] = z;
            klt[r * 16 + k] = klv;
        }
    }
    __syncthreads();
    if (first) {                                        // (block-uniform)
        // KL partials are per 16-ROW block whatever the tile shape, each summed the same way (wave w = block w: lane l
        // adds elements l, l + 64, .. of the block's 16 Z terms in order, then the shuffle tree)
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
        if (w < MT) {
            float kl = 0.f;
            for (int e = l; e < 16 * Z; e += 64) { const int r = e / Z; kl += klt[(16 * w + r) * 16 + (e - r * Z)]; }
            kl = wave_sum(kl);
            const int sr = ty * MT + w;
            if (l == 0 && sr * 16 < a.M) a.kl_partials[sr] = kl;
        }
    }
    SST(2);
    // ---- first decoder layer, this workgroup's columns: act(z . W + b), z from LDS (K = Z)        layer.py:325-331
    const f32x4v v2 = small_tile16<OUTER, MT, NT>(zt, MID_LD, T::TM, a.Wd, a.ldwd, a.hL, 0, n0, Z, red);
    SST(3);
    if (act2 && row2 < a.M) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = act_fwd(a.act, v2[c] + pd[c]);
        float* dst = a.OUT + (long)row2 * a.ldo + n0 + c42;
        if (n0 + c42 + 3 < a.hL && (a.ldo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n0 + c42 + c < a.hL) dst[c] = o[c];
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void mid_bwd_body(const MidArgs& a, int tx, int ty) {
    using TZ = TS<MT, 1>;              // the dz tile: (16 MT) x 16 (Z <= 16)
    using T = TS<MT, NT>;              // this workgroup's tile of the last encoder layer's dPre
    __shared__ float red[(TZ::RED > T::RED) ? TZ::RED : T::RED];
    __shared__ __align__(16) float dh[T::TM * MID_LD];
    const int tid = threadIdx.x;
    const int m0 = ty * T::TM, n0 = tx * T::TN;
    const int lr = tid >> 2, c4 = (tid & 3) * 4, row = m0 + lr;                    // dz tile: 4 threads per row
    const int lr2 = tid / T::TPR, c42 = (tid % T::TPR) * 4, row2 = m0 + lr2;
    const bool act1 = tid < TZ::ACTIVE, act2 = tid < T::ACTIVE;
    const int Z = a.Z, N2 = 2 * a.Z;
    const bool first = tx == 0;
    const int rowc = min(row, a.M - 1), rowc2 = min(row2, a.M - 1);
    float pmu[4], pls[4], pe[4], py[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kc = min(c4 + c, Z - 1);
        pmu[c] = a.ML[(long)rowc * a.ldml + kc];
        pls[c] = a.ML[(long)rowc * a.ldml + Z + kc];
        pe[c] = a.Eb[(long)rowc * a.ldz + kc];
        py[c] = a.H[(long)rowc2 * a.ldh + min(n0 + c42 + c, a.hL - 1)];
    }
    // the latent tile's pad columns [2 Z, 32) are read (and masked) by the second product: keep them finite
    for (int e = tid; e < T::TM * MID_LD; e += 256) dh[e] = 0.f;
    // ---- dz = dPre_dec1 . W_dec1^T (K = h_L), then through the reparameterisation and the KL term into both heads
    //      dmu = dz + mu/B ; dls = dz*eps*exp(ls) + (exp(2 ls) - 1)/B              layer.py:279, _ae_utils.py:49
    const f32x4v v = small_tile16<INNER, MT, 1>(a.DP, a.lddp, a.M, a.Wd, a.ldwd, Z, m0, 0, a.hL, red);
    if (act1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = c4 + c;
            if (col < Z) {
                float g_mu = 0.f, g_ls = 0.f;
                if (row < a.M) {
                    const float mu = pmu[c], ls = pls[c], e = pe[c];
                    const float el = expf(ls);
                    g_mu = act_bwd(a.act, mu, v[c] + mu * a.inv_batch);
                    g_ls = act_bwd(a.act, ls, v[c] * e * el + (expf(2.0f * ls) - 1.0f) * a.inv_batch);
                    if (first) {
                        a.DML[(long)row * a.lddml + col] = g_mu;
                        a.DML[(long)row * a.lddml + Z + col] = g_ls;
                    }
                }
                dh[lr * MID_LD + col] = g_mu;
                dh[lr * MID_LD + Z + col] = g_ls;
            }
        }
    }
    __syncthreads();
    // ---- dPre of the last encoder layer, this workgroup's columns: (dPre_heads . W_heads^T) * act'(h)   (K = 2 Z)
    const f32x4v v2 = small_tile16<INNER, MT, NT>(dh, MID_LD, T::TM, a.Wh, a.ldwh, a.hL, 0, n0, N2, red);
    if (act2 && row2 < a.M) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = act_bwd(a.act, py[c], v2[c]);
        float* dst = a.DOUT + (long)row2 * a.lddo + n0 + c42;
        if (n0 + c42 + 3 < a.hL && (a.lddo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n0 + c42 + c < a.hL) dst[c] = o[c];
        }
    }
}

template <int MT, int NT> __global__ __launch_bounds__(256, 2) void mid_fwd_kernel(const MidArgs a) {
    int tx, ty;
    xcd_tile((int)gridDim.x, (int)gridDim.y, a.xmap, tx, ty);
    mid_fwd_body<MT, NT>(a, tx, ty);
}
template <int MT, int NT> __global__ __launch_bounds__(256, 2) void mid_bwd_kernel(const MidArgs a) {
    int tx, ty;
    xcd_tile((int)gridDim.x, (int)gridDim.y, a.xmap, tx, ty);
    mid_bwd_body<MT, NT>(a, tx, ty);
}
template <int MT, int NT> __global__ __launch_bounds__(256, 2) void mid_fwd_multi(const MultiSlot<MidArgs>* __restrict__ t) {
    const MultiSlot<MidArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    mid_fwd_body<MT, NT>(s.a, (int)blockIdx.x, (int)blockIdx.y);
}
template <int MT, int NT> __global__ __launch_bounds__(256, 2) void mid_bwd_multi(const MultiSlot<MidArgs>* __restrict__ t) {
    const MultiSlot<MidArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    mid_bwd_body<MT, NT>(s.a, (int)blockIdx.x, (int)blockIdx.y);
}
#define SMRT_REG_MID(MT, NT) static MultiReg<MidArgs> reg_mid_fwd##MT##NT(mid_fwd_kernel<MT, NT>, mid_fwd_multi<MT, NT>); \
                             static MultiReg<MidArgs> reg_mid_bwd##MT##NT(mid_bwd_kernel<MT, NT>, mid_bwd_multi<MT, NT>)
SMRT_REG_MID(1, 1); SMRT_REG_MID(1, 2); SMRT_REG_MID(2, 2);
#undef SMRT_REG_MID

// number of KL partials a mid_fwd launch writes: one per 16-row block, whatever the tile shape
int mid_row_blocks(int rows, int hL) { (void)hL; return (rows + 15) / 16; }
void launch_mid_fwd(hipStream_t s, const MidArgs& a0) {
    MidArgs a = a0;
    a.xmap = pick_xmap(a.M, a.hL);
    // (a mid workgroup's time is its heads / dz tile, whatever the width of its own columns: 16x32 tiles halve the
    // workgroups that each recompute that tile, measured 46.6 -> 45.5 us per N1 mini-batch of 128 rows against 16x16)
    const int fm = forced_mid_tile(), sa = small_tile_shape(a.M, a.hL);
    const int sh = (fm == 22 || fm == 12 || fm == 11) ? fm : (sa == 11 ? 12 : sa);
    const dim3 grid((a.hL + shape_tn(sh) - 1) / shape_tn(sh), (a.M + shape_tm(sh) - 1) / shape_tm(sh));
    if (sh == 22)      SMRT_LAUNCH((mid_fwd_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else if (sh == 12) SMRT_LAUNCH((mid_fwd_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else               SMRT_LAUNCH((mid_fwd_kernel<1, 1>), grid, dim3(256), 0, s, a);
}
void launch_mid_bwd(hipStream_t s, const MidArgs& a0) {
    MidArgs a = a0;
    a.xmap = pick_xmap(a.M, a.hL);
    // (a mid workgroup's time is its heads / dz tile, whatever the width of its own columns: 16x32 tiles halve the
    // workgroups that each recompute that tile, measured 46.6 -> 45.5 us per N1 mini-batch of 128 rows against 16x16)
    const int fm = forced_mid_tile(), sa = small_tile_shape(a.M, a.hL);
    const int sh = (fm == 22 || fm == 12 || fm == 11) ? fm : (sa == 11 ? 12 : sa);
    const dim3 grid((a.hL + shape_tn(sh) - 1) / shape_tn(sh), (a.M + shape_tm(sh) - 1) / shape_tm(sh));
    if (sh == 22)      SMRT_LAUNCH((mid_bwd_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else if (sh == 12) SMRT_LAUNCH((mid_bwd_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else               SMRT_LAUNCH((mid_bwd_kernel<1, 1>), grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ weight + bias gradients of every layer, one launch
// [X | 1]^T . dPre with K = rows of the mini-batch (no split-K: one slab); the virtual ones index gives the bias row.
__device__ __forceinline__ void small_dw_body(const DwArgs& a) {
    __shared__ float red[4 * ST * SRED];
    const int t = blockIdx.x;
    int li = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_layers; ++i) if (t >= a.L[i].tile_begin) li = i;
    const DwLayer& L = a.L[li];
    const int lt = t - L.tile_begin;
    const int m0 = (lt / L.tiles_n) * ST, n0 = (lt % L.tiles_n) * ST;
    const f32x4v v = small_tile<true, 8>(L.A, L.lda, L.fi, L.B, L.ldb, L.fo, m0, n0, a.rows, red);
    const int row = m0 + (threadIdx.x >> 3), col0 = n0 + (threadIdx.x & 7) * 4;
    if (row <= L.fi) {                                   // row == fi is the bias gradient
        float* dst = L.C + (long)row * L.ldc + col0;     // ldc % 4 == 0 and col0 % 4 == 0: aligned
        if (col0 + 3 < L.fo) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (col0 + c < L.fo) dst[c] = v[c];
        }
    }
}

__global__ __launch_bounds__(256, 2) void small_dw_kernel(const DwArgs a) { small_dw_body(a); }
__global__ __launch_bounds__(256, 2) void small_dw_multi(const MultiSlot<DwArgs>* __restrict__ t) {
    const MultiSlot<DwArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx) return;
    small_dw_body(s.a);
}
static MultiReg<DwArgs> reg_small_dw(small_dw_kernel, small_dw_multi);

void launch_small_dw(hipStream_t s, const DwArgs& a) {
    SMRT_LAUNCH(small_dw_kernel, dim3(a.total_tiles), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ weight gradients + L2 + clip + optimizer, one launch
// The gradient of a small mini-batch is ONE slab (no split-K), so the workgroup that has reduced a 32x32 tile of
// [X | 1]^T . dPre holds final values: it applies L2, the value clip and the optimizer to those elements itself
// (base.py:127-149) -- the 2.9 MB gradient never goes to memory and comes back, and the 7 us optimizer sweep (a launch of
// its own, mostly its floor) is gone from the step.  The tile's weights and both slots are REQUESTED BEFORE the main loop,
// 16 bytes per thread from clamped addresses, so they arrive under it.  (Round 2 tried this with the slot loads behind the
// reduction: 19 us against 10 + 7.)  Same arithmetic as the sweep (opt_update_ieee), same bits.
template <int SGB>
__device__ __forceinline__ void small_dw_opt_body(const DwFusedArgs& a) {
    __shared__ float red[4 * ST * SRED];
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    SST(0);
    const DwRec L = a.tiles[t];                                          // (uniform index: one scalar load)
    const float* A = L.A ? L.A : a.X;
    const long lda = L.A ? (long)L.lda : a.ldx;
    const int m0 = L.m0, n0 = L.n0;
    const int row = m0 + (tid >> 3), col0 = n0 + (tid & 7) * 4;
    const bool inq = row <= L.fi && col0 < L.ldc;                         // this thread's quad exists (row == fi: the bias)
    const long ix = L.off + (long)min(row, L.fi) * L.ldc + min(col0, L.ldc - 4);
    const float4 w4 = *reinterpret_cast<const float4*>(a.W + ix);
    const float4 s04 = *reinterpret_cast<const float4*>(a.s0 + ix);
    const float4 s14 = *reinterpret_cast<const float4*>(a.s1 + ix);
    float alpha = 0.f;
    if (a.opt == 2) alpha = a.lr * sqrtf(1.0f - a.st->b2pow) / (1.0f - a.st->b1pow);
    if (a.advance && t == 0 && tid == 0) {
        // every earlier kernel of the step has read the step scalars (stream order) and no block of this one does
        // (advance is never set under Adam, whose blocks all read the beta powers above)
        a.st->step += 1;
        a.st->b1pow *= 0.9f;
        a.st->b2pow *= 0.999f;
    }
    SST(1);
    // (ONE batch of eight 8-wide k-blocks in flight: mini-batches up to 256 rows are a single round trip per wave, and without
    // the second buffer the kernel stays under 128 VGPRs -- four workgroups a CU, so that N1's 780 tiles are all resident at
    // once; with 163 VGPRs three were, the last 12 tiles started when the first ones had finished, and the launch took two
    // tile times instead of one)
    const f32x4v v = small_tile<true, SGB>(A, lda, L.fi, L.B, L.ldb, L.fo, m0, n0, a.rows, red);
    SST(2);
    float l2acc = 0.f;
    if (inq) {
        float gv[4] = {v[0], v[1], v[2], v[3]};
        float wv[4] = {w4.x, w4.y, w4.z, w4.w};
        float s0v[4] = {s04.x, s04.y, s04.z, s04.w}, s1v[4] = {s14.x, s14.y, s14.z, s14.w};
        if (a.has_l2 && L.l2 && row < L.fi) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gv[j] += a.l2 * wv[j];                               // tf.nn.l2_loss on the layer's weights, base.py:129
                l2acc += 0.5f * wv[j] * wv[j];
            }
        }
        if (a.grad_out) *reinterpret_cast<float4*>(a.grad_out + ix) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        const bool has_s0 = a.opt != 0 && a.opt != 5;
        const bool has_s1 = a.opt == 1 || a.opt == 2 || a.opt == 4;
        float gc[4], wn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gc[j] = a.clip ? tf_clip(gv[j], a.gmin, a.gmax) : gv[j];      // tf.clip_by_value, base.py:144
        // (opt_apply4: v_rsq_f32 / v_rcp-free forms, 1 ulp, as the LDS-resident kernels' epilogues -- four workgroups a CU
        // run this phase in lockstep, so its instruction count is paid four times over)
        opt_apply4(a.opt, a.lr, alpha, wv, gc, s0v, s1v, wn);
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = wn[j];
        *reinterpret_cast<float4*>(a.W + ix) = make_float4(wv[0], wv[1], wv[2], wv[3]);
        if (has_s0) *reinterpret_cast<float4*>(a.s0 + ix) = make_float4(s0v[0], s0v[1], s0v[2], s0v[3]);
        if (has_s1) *reinterpret_cast<float4*>(a.s1 + ix) = make_float4(s1v[0], s1v[1], s1v[2], s1v[3]);
    }
    SST(3);
#ifdef SMRT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SST(4);
#endif
    if (a.want_l2) {                                                 // (uniform) only the steps whose cost is read
        const float tot = block_sum_256(l2acc, sred);
        if (tid == 0) a.l2_partials[t] = tot;
    }
    SST(5);
}

template <int SGB> __global__ __launch_bounds__(256, 4) void small_dw_opt_kernel(const DwFusedArgs a) { small_dw_opt_body<SGB>(a); }
template <int SGB> __global__ __launch_bounds__(256, 4) void small_dw_opt_multi(const MultiSlot<DwFusedArgs>* __restrict__ t) {
    const MultiSlot<DwFusedArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx) return;
    small_dw_opt_body<SGB>(s.a);
}
static MultiReg<DwFusedArgs> reg_small_dw_opt4(small_dw_opt_kernel<4>, small_dw_opt_multi<4>);
static MultiReg<DwFusedArgs> reg_small_dw_opt8(small_dw_opt_kernel<8>, small_dw_opt_multi<8>);

// (the batch is sized to the mini-batch: up to 128 rows are four k-blocks per wave, up to 256 eight -- loads past the end
// would return zeros, but they would still be issued)
void launch_small_dw_opt(hipStream_t s, const DwFusedArgs& a) {
    if (a.rows <= 128) SMRT_LAUNCH(small_dw_opt_kernel<4>, dim3(a.total_tiles), dim3(256), 0, s, a);
    else               SMRT_LAUNCH(small_dw_opt_kernel<8>, dim3(a.total_tiles), dim3(256), 0, s, a);
}

// (The same launch on 64x64 tiles -- 16x16x4 MFMAs, both operands mn-contiguous, panels shared through LDS: half the operand
// traffic -- was built and measured in round 3: N1 alone 8.0 us per launch either way, the grouped 8-class step of C3 5.75 -> 6.2 ms
// per epoch.  Removed.)

}  // namespace smrt
