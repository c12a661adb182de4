// Two-kernel training step for nets that fit in LDS (tabular-sized VAEs, e.g. 64-128-(16,16)-128-64:
// 96 KB of weights + 43 KB of activations per workgroup).
//
//   fused_fwd_bwd_kernel : one workgroup (4 waves) owns 16 consecutive rows of the mini-batch and carries
//       them through EVERY layer, forward, loss and backward-through-activations.  All weights are staged
//       into LDS by ONE batch of wide loads; after that nothing on the critical path touches global
//       memory: activations stay in LDS between layers, both MFMA operands are LDS reads, stage
//       barriers wait for LDS only.  Layer outputs and pre-activation gradients are streamed to global
//       memory fire-and-forget for the second kernel.
//   fused_dw_opt_kernel  : one wave per 16x16 tile of [X | 1]^T . dPre (all layers in one grid); the
//       epilogue adds L2, undoes dropout, clips and applies the optimizer in place.  The last workgroup
//       to finish advances the step counter / Adam powers (ticket counter, no spinning).
//
// On this machine a dependent global-memory round trip inside these tiny launches costs ~2.5 us, more than
// all the arithmetic of a layer, so both kernels are built around "one batch of loads, then compute".
//
// Reference arithmetic: smrt/autoencode/layer.py:272-279,299-340,365-368; _ae_utils.py:15-49;
// base.py:127-149.
//
// MFMA v_mfma_f32_16x16x4_f32 (exact fp32) maps (cdna guide section 3): lane l: c = l & 15, q = l >> 4
//   A operand A[m = c][k = q], B operand B[k = q][n = c], C/D reg r: row = 4 q + r, col = c.
// Inside a 16-wide k-block the k order is permuted identically for both operands (lane group q takes
// k = 16 kb + 4 q + s at step s) so an LDS A fragment is one ds_read_b128.  Four k-blocks feed four
// independent accumulator chains (40-cycle dependent latency at a 32-cycle issue).
#include "vae_kernels.hpp"
#include "device_utils.hpp"

namespace smrt {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FTM = 16;                 // rows per workgroup
constexpr int FTHREADS = 512;          // 8 waves: two per SIMD, so one wave's epilogue/LDS latency hides under the other's MFMAs
constexpr int FWAVES = FTHREADS / 64;
constexpr int BTHREADS = 256;          // dW + optimizer kernel: 4 waves, one 16x16 tile each

#define MFMA16(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, C_, 0, 0, 0)

// Workgroup barrier that orders LDS traffic only: outstanding global STORES keep draining in the
// background (a __syncthreads() would wait for them: one memory round trip per layer).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0) only (vmcnt / expcnt fields at their maxima)
    __builtin_amdgcn_s_barrier();
}

template <bool TRANS>
__device__ __forceinline__ void load_frags(const float* A_lds, int sa, const float* W_lds, int sw, int c, int q, int n,
                                           int k0, float4& a, float4& b) {
    const int k = k0 + 4 * q;
    a = *reinterpret_cast<const float4*>(A_lds + c * sa + k);
    if constexpr (TRANS) {
        b = *reinterpret_cast<const float4*>(W_lds + n * sw + k);          // Wop[k][n] = W[n][k]
    } else {
        const float* p = W_lds + k * sw + n;                              // Wop[k][n] = W[k][n]
        b = make_float4(p[0], p[sw], p[2 * sw], p[3 * sw]);
    }
}

// C[16 x N] = A_lds[16 x K16] . Wop, all operands in LDS, zero padded to multiples of 16.
// epi(row, col, value) is called for col < N only.
template <bool TRANS, class Epi>
__device__ __forceinline__ void tile_gemm16(const float* A_lds, int sa, int K16, const float* W_lds, int sw,
                                            int N, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    for (int n0 = wave * 16; n0 < N; n0 += 16 * FWAVES) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        const int n = n0 + c;
        int k0 = 0;
        for (; k0 + 64 <= K16; k0 += 64) {
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            load_frags<TRANS>(A_lds, sa, W_lds, sw, c, q, n, k0, a0, b0);
            load_frags<TRANS>(A_lds, sa, W_lds, sw, c, q, n, k0 + 16, a1, b1);
            load_frags<TRANS>(A_lds, sa, W_lds, sw, c, q, n, k0 + 32, a2, b2);
            load_frags<TRANS>(A_lds, sa, W_lds, sw, c, q, n, k0 + 48, a3, b3);
            MFMA16(a0.x, b0.x, acc0); MFMA16(a1.x, b1.x, acc1); MFMA16(a2.x, b2.x, acc2); MFMA16(a3.x, b3.x, acc3);
            MFMA16(a0.y, b0.y, acc0); MFMA16(a1.y, b1.y, acc1); MFMA16(a2.y, b2.y, acc2); MFMA16(a3.y, b3.y, acc3);
            MFMA16(a0.z, b0.z, acc0); MFMA16(a1.z, b1.z, acc1); MFMA16(a2.z, b2.z, acc2); MFMA16(a3.z, b3.z, acc3);
            MFMA16(a0.w, b0.w, acc0); MFMA16(a1.w, b1.w, acc1); MFMA16(a2.w, b2.w, acc2); MFMA16(a3.w, b3.w, acc3);
        }
        for (; k0 < K16; k0 += 16) {
            float4 a0, b0;
            load_frags<TRANS>(A_lds, sa, W_lds, sw, c, q, n, k0, a0, b0);
            MFMA16(a0.x, b0.x, acc0); MFMA16(a0.y, b0.y, acc1); MFMA16(a0.z, b0.z, acc2); MFMA16(a0.w, b0.w, acc3);
        }
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(4 * q + r, n, (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]));
        }
    }
}

#ifdef SMRT_STAMPS
#define STAMP() do { __syncthreads(); if (a.dbg && blockIdx.x == 0 && tid == 0) { a.dbg[stamp_i] = wall_clock64(); a.dbg[32 + stamp_i] = clock64(); } ++stamp_i; } while (0)
#define STAMP_NS() do { if (a.dbg && blockIdx.x == 0 && tid == 0) { a.dbg[stamp_i] = wall_clock64(); a.dbg[32 + stamp_i] = clock64(); } ++stamp_i; } while (0)
#else
#define STAMP() do { } while (0)
#define STAMP_NS() do { } while (0)
#endif

constexpr int STAGE_U = 12;             // 16-byte weight loads in flight per lane (one batch = 96 KB per workgroup)

__global__ __launch_bounds__(FTHREADS) void fused_fwd_bwd_kernel(const FusedArgs a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float red[FWAVES];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * FTM;
    const int rows_here = min(FTM, a.rows - r0);
    const int L = a.L, NL = a.NL, Z = a.Z;
#ifdef SMRT_STAMPS
    int stamp_i = 0;
#endif
    STAMP();

    // ---- issue every global load this kernel depends on, in one batch: weights, X tile, step counter, epsilon,
    // and the per-layer descriptors (kernel arguments live in memory: the first scalar load of each cache
    // line is a full memory round trip, so touch them all now instead of one per stage)
    {
        int warm = 0;
#pragma unroll 1
        for (int j = 0; j < NL; ++j) warm += a.lay[j].fi + a.lay[j].w_off;
        asm volatile("" ::"s"(warm));
    }
    const uint32_t step = a.st->step;
    float* xs = lds + a.lds_x;
    const int xf4_per_row = (a.D + 3) >> 2;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool x_small = FTM * xf4_per_row <= FTHREADS;         // one float4 per lane covers the tile
    if (x_small && tid < rows_here * xf4_per_row && ((a.D & 3) == 0)) {
        const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
        xv = *reinterpret_cast<const float4*>(a.X + (long)(r0 + r) * a.ldx + c4);
    }
    float eps_pre[2] = {0.f, 0.f};
    if (a.eps_in) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * FTHREADS;
            if (i < rows_here * Z) eps_pre[u] = a.eps_in[(long)r0 * Z + i];
        }
    }
    const int total_f4 = a.w_total >> 2;
    for (int base = 0; base < total_f4; base += FTHREADS * STAGE_U) {
        f32x4 v[STAGE_U];                                  // native vectors: an array of HIP float4 structs lands in scratch
        int dst[STAGE_U];
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
            // clamped + unconditional: a branch around a load makes hipcc wait for each one separately;
            // a clamped duplicate later rewrites the last element with the same value.
            const int i = min(base + tid + u * FTHREADS, total_f4 - 1);
            v[u] = *reinterpret_cast<const f32x4*>(a.Wflat + 4L * i);
            dst[u] = a.stage_tab[i];                       // LDS float offset of this 16-byte piece (host-built)
        }
        if (base == 0) {
            // zero the LDS (pads must read as 0) while the loads are in flight
            float4* l4 = reinterpret_cast<float4*>(lds);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = tid; i < a.lds_total / 4; i += FTHREADS) l4[i] = z4;
            lds_barrier();
        }
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) *reinterpret_cast<f32x4*>(lds + dst[u]) = v[u];
    }
    if (x_small && ((a.D & 3) == 0)) {
        if (tid < rows_here * xf4_per_row) {
            const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
            *reinterpret_cast<float4*>(xs + r * a.s_x + c4) = xv;
        }
    } else {
        for (int i = tid; i < rows_here * a.D; i += FTHREADS) {
            const int r = i / a.D, cc = i - r * a.D;
            xs[r * a.s_x + cc] = a.X[(long)(r0 + r) * a.ldx + cc];
        }
    }
    lds_barrier();
    STAMP();

    float* zs = lds + a.lds_z;
    float* es = lds + a.lds_e;
    float* dcur = lds + a.lds_d0;
    float* dnext = lds + a.lds_d1;
    float bce = 0.f, kl = 0.f;

    // ---- forward
    for (int l = 0; l < NL; ++l) {
        const FusedLayer& y = a.lay[l];
        const float* in = (l == 0) ? xs : (l == L + 1 ? zs : lds + a.lay[l - 1].lds_out);
        const int sin = (l == 0) ? a.s_x : (l == L + 1 ? a.s_z : a.lay[l - 1].s_out);
        const float* wimg = lds + y.lds_w;
        const float* bias = lds + y.lds_b;
        const int K16 = (y.fi + 15) & ~15;
        if (l < NL - 1) {
            float* o = lds + y.lds_out;
            tile_gemm16<false>(in, sin, K16, wimg, y.s_w, y.fo, [&](int row, int col, float v) {
                const float yv = act_fwd(y.act, v + bias[col]);                        // layer.py:368
                o[row * y.s_out + col] = yv;
                if (row < rows_here) y.out_g[(long)(r0 + row) * y.ld + col] = yv;
            });
            STAMP_NS();
            lds_barrier();
            STAMP_NS();
            if (l == L) {
                // z = mu + eps * exp(log_sigma)  (layer.py:279) and the KL row terms (_ae_utils.py:49)
                for (int i = tid, u = 0; i < FTM * Z; i += FTHREADS, ++u) {
                    const int r = i / Z, k = i - r * Z;
                    if (r < rows_here) {
                        const float mu = o[r * y.s_out + k], ls = o[r * y.s_out + Z + k];
                        float e;
                        if (a.eps_in) e = (u == 0) ? eps_pre[0] : (u == 1 ? eps_pre[1] : a.eps_in[(long)r0 * Z + i]);
                        else e = philox_normal1(a.seed, STREAM_EPS, step, (uint64_t)(a.row0_global + r0 + r) * Z + k);
                        const float z = mu + e * expf(ls);
                        zs[r * a.s_z + k] = z;
                        es[r * a.s_z + k] = e;
                        a.Zb_g[(long)(r0 + r) * a.ldz + k] = z;
                        kl += -0.5f * (1.0f + 2.0f * ls - mu * mu - expf(2.0f * ls));
                    }
                }
                lds_barrier();
            }
        } else {
            // sigmoid output + clipped Bernoulli NLL + dL/dpre in one epilogue (layer.py:333-338, _ae_utils.py:31-32)
            tile_gemm16<false>(in, sin, K16, wimg, y.s_w, y.fo, [&](int row, int col, float v) {
                float d = 0.f;
                if (row < rows_here) {
                    const float p = 1.0f / (1.0f + expf(-(v + bias[col])));
                    const float x = xs[row * a.s_x + col];
                    const float pl = fmaxf(p, a.lo), pc = fminf(pl, a.hi), om = 1.0f - pc;
                    bce -= x * logf(pc) + (1.0f - x) * logf(om);
                    const bool pass = (p >= a.lo) && (pl <= a.hi);
                    const float dp = pass ? -(x / pc - (1.0f - x) / om) * a.inv_batch : 0.0f;
                    d = dp * p * (1.0f - p);
                    y.dpre_g[(long)(r0 + row) * y.ld + col] = d;
                }
                dcur[row * a.s_d + col] = d;
            });
            lds_barrier();
        }
    }
    STAMP();

    // ---- backward through the activations: dcur holds dPre of layer l, [16 x fo_l] zero padded
    for (int l = NL - 1; l >= 1; --l) {
        const FusedLayer& y = a.lay[l];
        const float* wimg = lds + y.lds_w;
        const int K16 = (y.fo + 15) & ~15;
        if (l == L + 1) {
            const FusedLayer& hd = a.lay[L];
            const float* ml = lds + hd.lds_out;
            tile_gemm16<true>(dcur, a.s_d, K16, wimg, y.s_w, y.fi, [&](int row, int col, float v) {
                float g_mu = 0.f, g_ls = 0.f;
                if (row < rows_here) {
                    const float mu = ml[row * hd.s_out + col], ls = ml[row * hd.s_out + Z + col];
                    const float e = es[row * a.s_z + col];
                    const float dmu = v + mu * a.inv_batch;
                    const float dls = v * e * expf(ls) + (expf(2.0f * ls) - 1.0f) * a.inv_batch;
                    g_mu = act_bwd(hd.act, mu, dmu);
                    g_ls = act_bwd(hd.act, ls, dls);
                    hd.dpre_g[(long)(r0 + row) * hd.ld + col] = g_mu;
                    hd.dpre_g[(long)(r0 + row) * hd.ld + Z + col] = g_ls;
                }
                dnext[row * a.s_d + col] = g_mu;
                dnext[row * a.s_d + Z + col] = g_ls;
            });
        } else {
            const FusedLayer& pv = a.lay[l - 1];
            const float* po = lds + pv.lds_out;
            tile_gemm16<true>(dcur, a.s_d, K16, wimg, y.s_w, y.fi, [&](int row, int col, float v) {
                float g = 0.f;
                if (row < rows_here) {
                    g = act_bwd(pv.act, po[row * pv.s_out + col], v);
                    pv.dpre_g[(long)(r0 + row) * pv.ld + col] = g;
                }
                dnext[row * a.s_d + col] = g;
            });
        }
        // the next stage reads k up to rup16(width just written): clear the tail this stage did not write
        {
            const int wrote = (l == L + 1) ? 2 * Z : y.fi;
            const int tail = ((wrote + 15) & ~15) - wrote;
            for (int i = tid; i < FTM * tail; i += FTHREADS) {
                const int r = i / tail;
                dnext[r * a.s_d + wrote + (i - r * tail)] = 0.f;
            }
        }
        lds_barrier();
        float* t = dcur; dcur = dnext; dnext = t;
    }
    STAMP();

    const float bsum = block_sum<FWAVES>(bce, red);
    const float ksum = block_sum<FWAVES>(kl, red);
    if (tid == 0) { a.bce_p[blockIdx.x] = bsum; a.kl_p[blockIdx.x] = ksum; }
    STAMP();
}

void launch_fused_fwd_bwd(hipStream_t s, const FusedArgs& a) {
    const int grid = (a.rows + FTM - 1) / FTM;
    hipLaunchKernelGGL(fused_fwd_bwd_kernel, dim3(grid), dim3(FTHREADS), (size_t)a.lds_total * sizeof(float), s, a);
}

int fused_set_max_lds(int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_fwd_bwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// =====================================================================================
// Weight + bias gradients of every layer with the optimizer fused into the epilogue.
// One wave per 16x16 tile of  [X | 1]^T . dPre ; K = rows of the mini-batch (no split).
// =====================================================================================
__device__ __forceinline__ float opt_apply(int opt, float lr, float alpha, float w, float g, float& s0, float& s1) {
    switch (opt) {
        case 1:     // RMSProp: decay .9, momentum 0, eps 1e-10 (rms slot starts at 1)
            s0 = s0 + (g * g - s0) * (1.0f - 0.9f);
            s1 = s1 * 0.0f + lr * g / sqrtf(s0 + 1e-10f);
            return w - s1;
        case 2:     // Adam
            s0 = s0 + (g - s0) * (1.0f - 0.9f);
            s1 = s1 + (g * g - s1) * (1.0f - 0.999f);
            return w - alpha * s0 / (sqrtf(s1) + 1e-8f);
        case 3: case 6:   // Adagrad / ProximalAdagrad(l1 = l2 = 0)
            s0 = s0 + g * g;
            return w - lr * g / sqrtf(s0);
        case 4: {   // Adadelta: rho .95, eps 1e-8
            s0 = s0 * 0.95f + g * g * (1.0f - 0.95f);
            const float upd = sqrtf(s1 + 1e-8f) / sqrtf(s0 + 1e-8f) * g;
            s1 = s1 * 0.95f + upd * upd * (1.0f - 0.95f);
            return w - lr * upd;
        }
        default:    // SGD / ProximalGradientDescent(l1 = l2 = 0)
            return w - lr * g;
    }
}

__global__ __launch_bounds__(BTHREADS) void fused_dw_opt_kernel(const DwOptArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int t = blockIdx.x * 4 + wave;
    const bool live = t < a.total_tiles;
    const int tt = live ? t : a.total_tiles - 1;           // idle waves shadow the last tile and store nothing

    int li = 0;
#pragma unroll 1
    for (int i = 1; i < a.n_layers; ++i) if (tt >= a.L[i].tile_begin) li = i;
    const DwOptLayer& Ly = a.L[li];
    const int lt = tt - Ly.tile_begin;
    const int m0 = (lt / Ly.tiles_n) * 16, n0 = (lt % Ly.tiles_n) * 16;
    const int fi = Ly.fi, fo = Ly.fo, rows = a.rows;

    // ---- batch 1 of loads: the optimizer operands of this lane's four output elements
    const bool has_s0 = a.opt != 0 && a.opt != 5;
    const bool has_s1 = a.opt == 1 || a.opt == 2 || a.opt == 4;
    const int col = n0 + c;
    long idx[4]; bool ok[4];
    float w[4], s0[4], s1[4], wd[4], mk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * q + r;
        ok[r] = live && row <= fi && col < fo;               // row == fi is the bias
        idx[r] = Ly.off + (long)min(row, fi) * Ly.ld + min(col, fo - 1);
        w[r] = a.W[idx[r]];
        s0[r] = has_s0 ? a.s0[idx[r]] : 0.f;
        s1[r] = has_s1 ? a.s1[idx[r]] : 0.f;
        wd[r] = a.has_drop ? a.Wd[idx[r]] : w[r];
        mk[r] = a.has_drop ? a.mask[idx[r]] : 1.f;
    }
    float alpha = 0.f;
    if (a.opt == 2) alpha = a.lr * sqrtf(1.0f - a.st->b2pow) / (1.0f - a.st->b1pow);

    // ---- [X | 1]^T . dPre over all rows: 64 rows (4 k-blocks) per batch of loads
    const int m = m0 + c, mm = min(m, fi - 1);               // A operand: output row m (input feature)
    const int n = n0 + c, nn = min(n, fo - 1);               // B operand: output col n
    const float* Ap = Ly.A + mm;
    const float* Bp = Ly.B + nn;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int k0 = 0; k0 < rows; k0 += 128) {                 // 128 rows (8 k-blocks, 64 loads per lane) per batch
        float av[8][4], bv[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + 16 * j + 4 * q + s, kk = min(k, rows - 1);
                const float x = Ap[(long)kk * Ly.lda];
                const float d = Bp[(long)kk * Ly.ldb];
                const bool kin = k < rows;
                av[j][s] = (kin && m < fi) ? x : ((kin && m == fi) ? 1.0f : 0.0f);     // virtual ones column -> db
                bv[j][s] = (kin && n < fo) ? d : 0.0f;
            }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                MFMA16(av[4 * jj + 0][s], bv[4 * jj + 0][s], acc0); MFMA16(av[4 * jj + 1][s], bv[4 * jj + 1][s], acc1);
                MFMA16(av[4 * jj + 2][s], bv[4 * jj + 2][s], acc2); MFMA16(av[4 * jj + 3][s], bv[4 * jj + 3][s], acc3);
            }
    }

    // ---- epilogue: L2, dropout backward, value clip, optimizer (base.py:127-149)
    float l2acc = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float g = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
        const bool is_w = (m0 + 4 * q + r) < fi;
        if (ok[r]) {
            if (is_w) {
                if (a.has_l2 && Ly.l2) { g += a.l2 * wd[r]; l2acc += 0.5f * wd[r] * wd[r]; }
                if (a.has_drop) g = (g / a.keep) * mk[r];
            }
            if (a.grad_out) a.grad_out[idx[r]] = g;
            if (a.clip) g = fminf(fmaxf(g, a.gmin), a.gmax);
            const float wn = opt_apply(a.opt, a.lr, alpha, w[r], g, s0[r], s1[r]);
            a.W[idx[r]] = wn;
            if (has_s0) a.s0[idx[r]] = s0[r];
            if (has_s1) a.s1[idx[r]] = s1[r];
        }
    }
    const float tot = block_sum_256(l2acc, red);            // __syncthreads inside: every lane's st reads are done
    if (tid == 0) {
        a.l2_partials[blockIdx.x] = tot;
        // last workgroup out advances the step-dependent scalars: every other workgroup has already read them
        const unsigned prev = atomicAdd(a.ticket, 1u);
        if (prev == gridDim.x - 1) {
            a.st->step += 1;
            a.st->b1pow *= 0.9f;
            a.st->b2pow *= 0.999f;
            *a.ticket = 0u;
        }
    }
}

int dw_opt_blocks(int total_tiles) { return (total_tiles + 3) / 4; }

void launch_fused_dw_opt(hipStream_t s, const DwOptArgs& a) {
    hipLaunchKernelGGL(fused_dw_opt_kernel, dim3(dw_opt_blocks(a.total_tiles)), dim3(BTHREADS), 0, s, a);
}

}  // namespace smrt
