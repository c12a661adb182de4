// Dense-layer GEMMs for SMALL mini-batches (M = rows <= SMALL_M_MAX), e.g. the reference's own configuration:
// 784-450-(10,10)-450-784 at batch 128 / 256 (smrt/balance/tests/test_smrt.py:17-21).
//
// At these sizes the LDS-tiled kernel (gemm_mfma_f32.hpp) is latency-bound twice over: 256 x 450 outputs are 32 of
// its 64x64 tiles (32 of 256 CUs busy), and each workgroup walks K = 784 as 25 dependent
// global-load -> LDS -> barrier -> MFMA rounds of ~1 us.  Here instead
//   * a workgroup owns ONE 32x32 output tile (120 workgroups for 256 x 450) and its four waves split K between them:
//     wave w takes the 8-wide k-blocks with index = w (mod 4), so the dependent MFMA chain is 4x shorter;
//   * each wave loads its operand fragments straight from global memory into registers in MFMA layout -- no LDS
//     staging, no barrier in the main loop -- eight k-blocks (64 registers) per batch, the next batch requested
//     before the current one is multiplied;
//   * the four partial tiles meet in LDS once, and all 256 threads run the fused epilogue on 4 outputs each with
//     16-byte stores.
//
// MFMA v_mfma_f32_32x32x2_f32 maps: lane l: r = l & 31, h = l >> 5; A operand A[m = r][k = h], B operand B[k = h][n = r];
// C/D reg i: row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r.  Inside an 8-wide k-block lane half h takes k = 4 h + s at step s
// for BOTH operands (the MFMA sums over k), so a k-contiguous operand is one 16-byte load per k-block.
//
// Reference arithmetic: as vae_kernels.hip (layer.py:365-368, _ae_utils.py:15-49, base.py:127-149).
#include "vae_kernels.hpp"
#include "device_utils.hpp"
#include "gemm_mfma_f32.hpp"

namespace smrt {

constexpr int ST = 32;                 // output tile edge
constexpr int SG = 8;                  // k-blocks per wave per batch (8 x 8 = 64 k)
constexpr int SRED = 33;               // LDS row stride of a partial tile

typedef float f32x4v __attribute__((ext_vector_type(4)));

// The four k-step values of one 8-wide k-block for this lane (k = kb4 + s, s = 0..3; kb4 = block start + 4 h), loaded from a
// clamped in-bounds address.  mask_kblock() then zeroes (or sets to one, for the virtual ones index of the bias gradient)
// what lies outside the matrix -- LATER, when the values are about to be used: a select right behind the load would make the
// wave wait for its prefetch before multiplying the batch it already has.
template <int LAY>
__device__ __forceinline__ f32x4v load_kblock(const float* __restrict__ g, long ld, int mn, int MN_data, int kb4, int K) {
    if constexpr (LAY == INNER) {       // [mn][k], k contiguous: one 16-byte load
        return *reinterpret_cast<const f32x4v*>(g + (long)min(mn, MN_data - 1) * ld + min(kb4, (int)ld - 4));
    } else {                            // [k][mn], mn contiguous: four 4-byte loads, each 128 B contiguous per half-wave
        const float* p = g + min(mn, MN_data - 1);
        f32x4v v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = p[(long)min(kb4 + s, K - 1) * ld];
        return v;
    }
}
template <bool ONES>
__device__ __forceinline__ f32x4v mask_kblock(f32x4v t, int mn, int MN_data, int kb4, int K) {
    const bool in = mn < MN_data;
    const float fill = (ONES && mn == MN_data) ? 1.f : 0.f;
    f32x4v v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (kb4 + s < K) ? (in ? t[s] : fill) : 0.f;
    return v;
}

// One 32x32 tile of A.B over k in [0, K): returns this thread's four reduced outputs (row = tid >> 3, columns
// 4 * (tid & 7) .. + 3 of the tile).  `red` = 4 * 32 * SRED floats of LDS.
template <int ALAY, int BLAY, bool A_ONES>
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

    f32x4v fa[SG], fb[SG], na[SG], nb[SG];
    auto fetch = [&](int first, f32x4v (&xa)[SG], f32x4v (&xb)[SG]) {
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            const int kb4 = ((first + j) * 4 + wave) * 8 + 4 * h;      // blocks past the end read a clamped address
            xa[j] = load_kblock<ALAY>(A, lda, m0 + r, M_data, kb4, K);
            xb[j] = load_kblock<BLAY>(B, ldb, n0 + r, N_data, kb4, K);
        }
    };
    // interior tiles with K a multiple of 8 need no masking at all (wave-uniform)
    const bool clean = (m0 + ST <= M_data) && (n0 + ST <= N_data) && ((K & 7) == 0);
    auto mask = [&](int first, f32x4v (&xa)[SG], f32x4v (&xb)[SG]) {
        if (clean) return;
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            const int kb4 = ((first + j) * 4 + wave) * 8 + 4 * h;
            xa[j] = mask_kblock<A_ONES>(xa[j], m0 + r, M_data, kb4, K);
            xb[j] = mask_kblock<false>(xb[j], n0 + r, N_data, kb4, K);
        }
    };
    if (mine > 0) { fetch(0, fa, fb); mask(0, fa, fb); }
#pragma unroll 1
    for (int i = 0; i < mine; i += SG) {
        const bool more = i + SG < mine;
        if (more) fetch(i + SG, na, nb);
        asm volatile("" ::: "memory");                   // next batch is requested before this one is multiplied
        __builtin_amdgcn_sched_barrier(0);
        const int cnt = min(SG, mine - i);
#pragma unroll
        for (int j = 0; j < SG; ++j) {
            if (j < cnt) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][s], fb[j][s], acc, 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < SG; ++j) { fa[j] = na[j]; fb[j] = nb[j]; }
            mask(i + SG, fa, fb);
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

// ------------------------------------------------------------------ dense layer forward / backward
// (body shared by the plain kernel and its grouped twin; gx = columns of tiles of THIS problem's grid)
template <int ALAY, int BLAY, int EPI>
__device__ __forceinline__ void small_gemm_body(const GemmArgs& a, unsigned gx) {
    __shared__ float red[4 * ST * SRED];
    __shared__ float sred[4];
    const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
    const int row = m0 + (threadIdx.x >> 3), col0 = n0 + (threadIdx.x & 7) * 4;
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
    const f32x4v v = small_tile<ALAY, BLAY, false>(a.A, a.lda, a.M, a.B, a.ldb, a.N, m0, n0, a.K, red);
    const bool rin = row < a.M;
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
        const float tot = block_sum_256(lsum, sred);
        if (threadIdx.x == 0) a.partials[blockIdx.y * gx + blockIdx.x] = tot;
    }
}

template <int ALAY, int BLAY, int EPI>
__global__ __launch_bounds__(256, 2) void small_gemm_kernel(const GemmArgs a) { small_gemm_body<ALAY, BLAY, EPI>(a, gridDim.x); }

// grouped twin: gridDim.z problems (one per handle), each with its own arguments and grid extent in a device table
template <int ALAY, int BLAY, int EPI>
__global__ __launch_bounds__(256, 2) void small_gemm_multi(const MultiSlot<GemmArgs>* __restrict__ t) {
    const MultiSlot<GemmArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    small_gemm_body<ALAY, BLAY, EPI>(s.a, s.gx);
}
#define SMRT_REG_SMALL(AL, BL, EP) static MultiReg<GemmArgs> reg_small_##AL##BL##EP(small_gemm_kernel<AL, BL, EP>, small_gemm_multi<AL, BL, EP>)
SMRT_REG_SMALL(INNER, OUTER, EPI_BIAS_ACT); SMRT_REG_SMALL(INNER, OUTER, EPI_BCE); SMRT_REG_SMALL(INNER, OUTER, EPI_MSE);
SMRT_REG_SMALL(INNER, OUTER, EPI_HEADS_SPLIT); SMRT_REG_SMALL(INNER, INNER, EPI_ACTGRAD); SMRT_REG_SMALL(INNER, INNER, EPI_LATENT);
#undef SMRT_REG_SMALL

int small_gemm_blocks(const GemmArgs& a) { return ((a.M + ST - 1) / ST) * ((a.N + ST - 1) / ST); }

template <int ALAY, int BLAY, int EPI>
static void launch_small_t(hipStream_t s, const GemmArgs& a) {
    dim3 grid((a.N + ST - 1) / ST, (a.M + ST - 1) / ST);
    SMRT_LAUNCH((small_gemm_kernel<ALAY, BLAY, EPI>), grid, dim3(256), 0, s, a);
}

void launch_small_gemm(hipStream_t s, int kind, int epi, const GemmArgs& a) {
    if (kind == GEMM_NN) {
        if (epi == EPI_BIAS_ACT)         launch_small_t<INNER, OUTER, EPI_BIAS_ACT>(s, a);
        else if (epi == EPI_BCE)         launch_small_t<INNER, OUTER, EPI_BCE>(s, a);
        else if (epi == EPI_MSE)         launch_small_t<INNER, OUTER, EPI_MSE>(s, a);
        else                             launch_small_t<INNER, OUTER, EPI_HEADS_SPLIT>(s, a);
    } else {
        if (epi == EPI_ACTGRAD)          launch_small_t<INNER, INNER, EPI_ACTGRAD>(s, a);
        else                             launch_small_t<INNER, INNER, EPI_LATENT>(s, a);
    }
}

// ------------------------------------------------------------------ the middle of the VAE, fused
// At these sizes every launch costs ~4.5 us before it computes anything (its arguments, its operands and the step scalars
// are three dependent memory round trips: the bare reparameterisation, 1 280 elements, takes 4.5 us), and the three
// launches around the latent layer -- heads (2 Z <= 32 columns: four workgroups), reparameterisation, first decoder layer
// (K = Z) -- are nothing but that floor, three times: 20 us of a 73 us step of the reference's own net at batch 128.
// Here every workgroup of the first decoder layer's grid recomputes the heads tile of its 32 rows (one 32x32 tile over
// K = h_L: what any workgroup of the neighbouring layers does), reparameterises into LDS, and multiplies its own 32
// output columns from there; the workgroups of the first column also write the heads, z, epsilon and the KL partial.
// Backward the same way: dz tile (K = h_L) -> latent gradients in LDS -> own 32 columns of the encoder's dPre (K = 2 Z).
// Same arithmetic in the same order as the separate kernels (small_tile, the same epilogues): results are bit-identical.
constexpr int MID_LD = 36;             // LDS row stride of the 32-row latent tiles (2 Z <= 32 columns + slack for clamped reads)

__device__ __forceinline__ void mid_fwd_body(const MidArgs& a) {
    __shared__ float red[4 * ST * SRED];
    __shared__ __align__(16) float ml[ST * MID_LD];
    __shared__ __align__(16) float zt[ST * MID_LD];
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
    const int lr = tid >> 3, c4 = (tid & 7) * 4, row = m0 + lr;
    const int Z = a.Z, N2 = 2 * a.Z;
    const bool first = blockIdx.x == 0;
    float pb[4], pd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        pb[c] = a.bh[min(c4 + c, N2 - 1)];
        pd[c] = a.bd[min(n0 + c4 + c, a.hL - 1)];
    }
    // epsilon of this thread's reparameterisation elements, requested now (injected) -- up to 2 per thread (32 Z <= 512)
    float pe[2] = {0.f, 0.f};
    if (a.eps_in) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            const int r = i / Z, k = i - r * Z;
            if (i < ST * Z) pe[u] = a.eps_in[(long)min(m0 + r, a.M - 1) * Z + k];
        }
    }
    const uint32_t step = a.st->step;
    // ---- heads: act(h . [W_mu | W_ls] + b)                                        layer.py:312-317
    const f32x4v v = small_tile<INNER, OUTER, false>(a.H, a.ldh, a.M, a.Wh, a.ldwh, N2, m0, 0, a.hL, red);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = c4 + c;
        const bool in = row < a.M && col < N2;
        const float y = in ? act_fwd(a.act, v[c] + pb[c]) : 0.f;
        ml[lr * MID_LD + col] = y;
        if (first && in) a.ML[(long)row * a.ldml + col] = y;
    }
    __syncthreads();
    // ---- z = mu + eps * exp(log_sigma), KL row terms                              layer.py:279, _ae_utils.py:49
    float kl = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        if (i < ST * Z) {
            const int r = i / Z, k = i - r * Z, grow = m0 + r;
            float z = 0.f;
            if (grow < a.M) {
                const float mu = ml[r * MID_LD + k], ls = ml[r * MID_LD + Z + k];
                const float e = a.eps_in ? pe[u] : philox_normal1(a.seed, STREAM_EPS, step, (uint64_t)(a.row0_global + grow) * Z + k);
                z = mu + e * expf(ls);
                if (first) {
                    a.Zb[(long)grow * a.ldz + k] = z;
                    a.Eb[(long)grow * a.ldz + k] = e;
                    kl += -0.5f * (1.0f + 2.0f * ls - mu * mu - expf(2.0f * ls));
                }
            }
            zt[r * MID_LD + k] = z;
        }
    }
    if (first) {                                        // (block-uniform)
        const float tot = block_sum_256(kl, sred);
        if (tid == 0) a.kl_partials[blockIdx.y] = tot;
    }
    __syncthreads();
    // ---- first decoder layer, this workgroup's 32 columns: act(z . W + b), z from LDS (K = Z)        layer.py:325-331
    const f32x4v v2 = small_tile<INNER, OUTER, false>(zt, MID_LD, ST, a.Wd, a.ldwd, a.hL, 0, n0, Z, red);
    if (row < a.M) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = act_fwd(a.act, v2[c] + pd[c]);
        float* dst = a.OUT + (long)row * a.ldo + n0 + c4;
        if (n0 + c4 + 3 < a.hL && (a.ldo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n0 + c4 + c < a.hL) dst[c] = o[c];
        }
    }
}

__device__ __forceinline__ void mid_bwd_body(const MidArgs& a) {
    __shared__ float red[4 * ST * SRED];
    __shared__ __align__(16) float dh[ST * MID_LD];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * ST, n0 = blockIdx.x * ST;
    const int lr = tid >> 3, c4 = (tid & 7) * 4, row = m0 + lr;
    const int Z = a.Z, N2 = 2 * a.Z;
    const bool first = blockIdx.x == 0;
    const int rowc = min(row, a.M - 1);
    float pmu[4], pls[4], pe[4], py[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kc = min(c4 + c, Z - 1);
        pmu[c] = a.ML[(long)rowc * a.ldml + kc];
        pls[c] = a.ML[(long)rowc * a.ldml + Z + kc];
        pe[c] = a.Eb[(long)rowc * a.ldz + kc];
        py[c] = a.H[(long)rowc * a.ldh + min(n0 + c4 + c, a.hL - 1)];
    }
    // ---- dz = dPre_dec1 . W_dec1^T (K = h_L), then through the reparameterisation and the KL term into both heads
    //      dmu = dz + mu/B ; dls = dz*eps*exp(ls) + (exp(2 ls) - 1)/B              layer.py:279, _ae_utils.py:49
    const f32x4v v = small_tile<INNER, INNER, false>(a.DP, a.lddp, a.M, a.Wd, a.ldwd, Z, m0, 0, a.hL, red);
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
    __syncthreads();
    // ---- dPre of the last encoder layer, this workgroup's 32 columns: (dPre_heads . W_heads^T) * act'(h)   (K = 2 Z)
    const f32x4v v2 = small_tile<INNER, INNER, false>(dh, MID_LD, ST, a.Wh, a.ldwh, a.hL, 0, n0, N2, red);
    if (row < a.M) {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = act_bwd(a.act, py[c], v2[c]);
        float* dst = a.DOUT + (long)row * a.lddo + n0 + c4;
        if (n0 + c4 + 3 < a.hL && (a.lddo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n0 + c4 + c < a.hL) dst[c] = o[c];
        }
    }
}

__global__ __launch_bounds__(256, 2) void mid_fwd_kernel(const MidArgs a) { mid_fwd_body(a); }
__global__ __launch_bounds__(256, 2) void mid_bwd_kernel(const MidArgs a) { mid_bwd_body(a); }
__global__ __launch_bounds__(256, 2) void mid_fwd_multi(const MultiSlot<MidArgs>* __restrict__ t) {
    const MultiSlot<MidArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    mid_fwd_body(s.a);
}
__global__ __launch_bounds__(256, 2) void mid_bwd_multi(const MultiSlot<MidArgs>* __restrict__ t) {
    const MultiSlot<MidArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx || blockIdx.y >= s.gy) return;
    mid_bwd_body(s.a);
}
static MultiReg<MidArgs> reg_mid_fwd(mid_fwd_kernel, mid_fwd_multi);
static MultiReg<MidArgs> reg_mid_bwd(mid_bwd_kernel, mid_bwd_multi);

int mid_row_blocks(int rows) { return (rows + ST - 1) / ST; }
void launch_mid_fwd(hipStream_t s, const MidArgs& a) {
    SMRT_LAUNCH(mid_fwd_kernel, dim3((a.hL + ST - 1) / ST, (a.M + ST - 1) / ST), dim3(256), 0, s, a);
}
void launch_mid_bwd(hipStream_t s, const MidArgs& a) {
    SMRT_LAUNCH(mid_bwd_kernel, dim3((a.hL + ST - 1) / ST, (a.M + ST - 1) / ST), dim3(256), 0, s, a);
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
    const f32x4v v = small_tile<OUTER, OUTER, true>(L.A, L.lda, L.fi, L.B, L.ldb, L.fo, m0, n0, a.rows, red);
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

}  // namespace smrt
