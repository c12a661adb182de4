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

// Weight image in LDS ("k-quad interleaved"): element W[kf][nf] (kf = fan-in index, nf = fan-out index) lives at
//   ((kf >> 2) * nn + nf) * 4 + (kf & 3)      floats, nn = rup16(fan_out) + 1 (odd -> conflict-free strided reads)
// Forward  (sum over kf): a lane's four k values are contiguous       -> ONE ds_read_b128.
// Backward (sum over nf): a lane's four k values are 16 bytes apart   -> ds_read with immediate offsets.
template <bool TRANS>
__device__ __forceinline__ void load_frags(const float* A_lds, int sa, const float* W_lds, int nn, int c, int q, int n,
                                           int k0, float4& a, float4& b) {
    const int k = k0 + 4 * q;
    a = *reinterpret_cast<const float4*>(A_lds + c * sa + k);
    if constexpr (TRANS) {
        const float* p = W_lds + (((n >> 2) * nn + k) << 2) + (n & 3);    // Wop[k][n] = W[kf = n][nf = k]
        b = make_float4(p[0], p[4], p[8], p[12]);
    } else {
        b = *reinterpret_cast<const float4*>(W_lds + ((((k0 >> 2) + q) * nn + n) << 2));   // Wop[k][n] = W[kf = k][nf = n]
    }
}

// C[16 x N] = A_lds[16 x K16] . Wop, all operands in LDS, zero padded to multiples of 16.
// epi(row0, col, v[4]) is called once per lane for col < N with the values of rows row0 .. row0+3.
template <bool TRANS, class Epi>
__device__ __forceinline__ void tile_gemm16(const float* A_lds, int sa, int K16, const float* W_lds, int nn,
                                            int N, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    for (int n0 = wave * 16; n0 < N; n0 += 16 * FWAVES) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        const int n = n0 + c;
        int k0 = 0;
        for (; k0 + 64 <= K16; k0 += 64) {
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            load_frags<TRANS>(A_lds, sa, W_lds, nn, c, q, n, k0, a0, b0);
            load_frags<TRANS>(A_lds, sa, W_lds, nn, c, q, n, k0 + 16, a1, b1);
            load_frags<TRANS>(A_lds, sa, W_lds, nn, c, q, n, k0 + 32, a2, b2);
            load_frags<TRANS>(A_lds, sa, W_lds, nn, c, q, n, k0 + 48, a3, b3);
            MFMA16(a0.x, b0.x, acc0); MFMA16(a1.x, b1.x, acc1); MFMA16(a2.x, b2.x, acc2); MFMA16(a3.x, b3.x, acc3);
            MFMA16(a0.y, b0.y, acc0); MFMA16(a1.y, b1.y, acc1); MFMA16(a2.y, b2.y, acc2); MFMA16(a3.y, b3.y, acc3);
            MFMA16(a0.z, b0.z, acc0); MFMA16(a1.z, b1.z, acc1); MFMA16(a2.z, b2.z, acc2); MFMA16(a3.z, b3.z, acc3);
            MFMA16(a0.w, b0.w, acc0); MFMA16(a1.w, b1.w, acc1); MFMA16(a2.w, b2.w, acc2); MFMA16(a3.w, b3.w, acc3);
        }
        for (; k0 < K16; k0 += 16) {
            float4 a0, b0;
            load_frags<TRANS>(A_lds, sa, W_lds, nn, c, q, n, k0, a0, b0);
            MFMA16(a0.x, b0.x, acc0); MFMA16(a0.y, b0.y, acc1); MFMA16(a0.z, b0.z, acc2); MFMA16(a0.w, b0.w, acc3);
        }
        if (n < N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
            epi(4 * q, n, v);                       // rows 4q .. 4q+3 of column n
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
    __shared__ FusedLayer slay[MAX_LAYERS];
    {
        constexpr int NW = (int)(sizeof(FusedLayer) * MAX_LAYERS / sizeof(int));
        const int* src = reinterpret_cast<const int*>(&a.lay[0]);
        int* dstl = reinterpret_cast<int*>(&slay[0]);
        for (int i = tid; i < NW; i += FTHREADS) dstl[i] = src[i];
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
        for (int u = 0; u < STAGE_U; ++u) {
            float* d = lds + dst[u];                       // consecutive fan-out columns are 4 floats apart in the image
            d[0] = v[u][0]; d[4] = v[u][1]; d[8] = v[u][2]; d[12] = v[u][3];
        }
    }
    // X tile into LDS, and its transposed copy XT[col][ldT] for the weight-gradient kernel (zeros on padded rows)
    if (x_small && ((a.D & 3) == 0)) {
        if (tid < FTM * xf4_per_row) {
            const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
            *reinterpret_cast<float4*>(xs + r * a.s_x + c4) = xv;
            float* xt = a.XT + (long)c4 * a.ldT + r0 + r;
            xt[0] = xv.x; xt[a.ldT] = xv.y; xt[2 * a.ldT] = xv.z; xt[3 * a.ldT] = xv.w;
        }
    } else {
        for (int i = tid; i < FTM * a.D; i += FTHREADS) {
            const int r = i / a.D, cc = i - r * a.D;
            const float xval = (r < rows_here) ? a.X[(long)(r0 + r) * a.ldx + cc] : 0.0f;
            xs[r * a.s_x + cc] = xval;
            a.XT[(long)cc * a.ldT + r0 + r] = xval;
        }
    }
    lds_barrier();
    STAMP();

    float* zs = lds + a.lds_z;
    float* es = lds + a.lds_e;
    float* dcur = lds + a.lds_d0;
    float* dnext = lds + a.lds_d1;
    float bce = 0.f, kl = 0.f;

    // Descriptor fields are copied into locals at the top of each stage: read through `slay[l]` inside the
    // epilogue they are re-fetched from the kernel-argument segment for every element.
    const float lo = a.lo, hi = a.hi, inv_batch = a.inv_batch;
    const int s_x = a.s_x, s_z = a.s_z, s_d = a.s_d;
    const long ldT = a.ldT;

    // Epilogues are branch-free over the lane's four rows: workspace rows are padded to a multiple of 16 so
    // rows past the batch are computed and stored like the others (their dPre is forced to 0, nothing reads them).
    // ---- forward
    for (int l = 0; l < NL; ++l) {
        const float* in = (l == 0) ? xs : (l == L + 1 ? zs : lds + slay[l - 1].lds_out);
        const int sin = (l == 0) ? s_x : (l == L + 1 ? s_z : slay[l - 1].s_out);
        const float* wimg = lds + slay[l].lds_w;
        const float* bias = lds + slay[l].lds_b;                       // stride 4 (same interleave as the weights)
        const int fi = slay[l].fi, fo = slay[l].fo, nn = slay[l].nn, act = slay[l].act;
        const int K16 = (fi + 15) & ~15;
        if (l < NL - 1) {
            float* o = lds + slay[l].lds_out;
            const int s_out = slay[l].s_out;
            float* og = slay[l].out_g + r0;                                   // transposed: [col][ldT], rows contiguous
            tile_gemm16<false>(in, sin, K16, wimg, nn, fo, [&](int row0, int col, float (&v)[4]) {
                const float bcol = bias[4 * col];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bcol;
                act_fwd4(act, v);                                                     // layer.py:368
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(row0 + r) * s_out + col] = v[r];
                *reinterpret_cast<float4*>(og + (long)col * ldT + row0) = make_float4(v[0], v[1], v[2], v[3]);
            });
            STAMP_NS();
            lds_barrier();
            STAMP_NS();
            if (l == L) {
                // z = mu + eps * exp(log_sigma)  (layer.py:279) and the KL row terms (_ae_utils.py:49)
                float* zg = a.Zb_g + r0;                                      // transposed [Z][ldT]
                for (int i = tid, u = 0; i < FTM * Z; i += FTHREADS, ++u) {
                    const int r = i / Z, k = i - r * Z;
                    if (r >= rows_here) zg[(long)k * ldT + r] = 0.0f;          // finite filler for the padded rows
                    if (r < rows_here) {
                        const float mu = o[r * s_out + k], ls = o[r * s_out + Z + k];
                        float e;
                        if (a.eps_in) e = (u == 0) ? eps_pre[0] : (u == 1 ? eps_pre[1] : a.eps_in[(long)r0 * Z + i]);
                        else e = philox_normal1(a.seed, STREAM_EPS, step, (uint64_t)(a.row0_global + r0 + r) * Z + k);
                        const float z = mu + e * expf(ls);
                        zs[r * s_z + k] = z;
                        es[r * s_z + k] = e;
                        zg[(long)k * ldT + r] = z;
                        kl += -0.5f * (1.0f + 2.0f * ls - mu * mu - expf(2.0f * ls));
                    }
                }
                lds_barrier();
            }
        } else {
            // sigmoid output + clipped Bernoulli NLL + dL/dpre in one epilogue (layer.py:333-338, _ae_utils.py:31-32)
            float* dg = slay[l].dpre_g + r0;
            tile_gemm16<false>(in, sin, K16, wimg, nn, fo, [&](int row0, int col, float (&v)[4]) {
                const float bcol = bias[4 * col];
                float dv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (row0 + r) < rows_here;
                    const float p = 1.0f / (1.0f + expf(-(v[r] + bcol)));
                    const float x = xs[(row0 + r) * s_x + col];
                    const float pl = fmaxf(p, lo), pc = fminf(pl, hi), om = 1.0f - pc;
                    const float term = x * logf(pc) + (1.0f - x) * logf(om);
                    bce -= valid ? term : 0.0f;
                    const bool pass = valid && (p >= lo) && (pl <= hi);
                    const float dp = pass ? -(x / pc - (1.0f - x) / om) * inv_batch : 0.0f;
                    const float d = valid ? dp * p * (1.0f - p) : 0.0f;
                    dv[r] = d;
                    dcur[(row0 + r) * s_d + col] = d;
                }
                *reinterpret_cast<float4*>(dg + (long)col * ldT + row0) = make_float4(dv[0], dv[1], dv[2], dv[3]);
            });
            lds_barrier();
        }
    }
    STAMP();

    // ---- backward through the activations: dcur holds dPre of layer l, [16 x fo_l] zero padded
    for (int l = NL - 1; l >= 1; --l) {
        const float* wimg = lds + slay[l].lds_w;
        const int fi = slay[l].fi, nn = slay[l].nn;
        const int K16 = (slay[l].fo + 15) & ~15;
        const float* po = lds + slay[l - 1].lds_out;
        const int p_sout = slay[l - 1].s_out, p_act = slay[l - 1].act;
        float* pg = slay[l - 1].dpre_g + r0;
        if (l == L + 1) {
            // l - 1 == L: the fused heads.  v = dL/dz through the reparameterisation and the KL term.
            tile_gemm16<true>(dcur, s_d, K16, wimg, nn, fi, [&](int row0, int col, float (&v)[4]) {
                float mu[4], ls[4], g_mu[4], g_ls[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mu[r] = po[(row0 + r) * p_sout + col];
                    ls[r] = po[(row0 + r) * p_sout + Z + col];
                    const float e = es[(row0 + r) * s_z + col];
                    g_mu[r] = v[r] + mu[r] * inv_batch;
                    g_ls[r] = v[r] * e * expf(ls[r]) + (expf(2.0f * ls[r]) - 1.0f) * inv_batch;
                }
                act_bwd4(p_act, mu, g_mu);
                act_bwd4(p_act, ls, g_ls);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (row0 + r) < rows_here;
                    g_mu[r] = valid ? g_mu[r] : 0.0f;
                    g_ls[r] = valid ? g_ls[r] : 0.0f;
                    dnext[(row0 + r) * s_d + col] = g_mu[r];
                    dnext[(row0 + r) * s_d + Z + col] = g_ls[r];
                }
                *reinterpret_cast<float4*>(pg + (long)col * ldT + row0) = make_float4(g_mu[0], g_mu[1], g_mu[2], g_mu[3]);
                *reinterpret_cast<float4*>(pg + (long)(Z + col) * ldT + row0) = make_float4(g_ls[0], g_ls[1], g_ls[2], g_ls[3]);
            });
        } else {
            tile_gemm16<true>(dcur, s_d, K16, wimg, nn, fi, [&](int row0, int col, float (&v)[4]) {
                float yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = po[(row0 + r) * p_sout + col];
                act_bwd4(p_act, yv, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) dnext[(row0 + r) * s_d + col] = v[r];
                *reinterpret_cast<float4*>(pg + (long)col * ldT + row0) = make_float4(v[0], v[1], v[2], v[3]);
            });
        }
        // the next stage reads k up to rup16(width just written): clear the tail this stage did not write
        {
            const int wrote = (l == L + 1) ? 2 * Z : fi;
            const int tail = ((wrote + 15) & ~15) - wrote;
            for (int i = tid; i < FTM * tail; i += FTHREADS) {
                const int r = i / tail;
                dnext[r * s_d + wrote + (i - r * tail)] = 0.f;
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
#ifdef SMRT_STAMPS
#define BST(i) do { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[i] = wall_clock64(); } while (0)
#else
#define BST(i) do { } while (0)
#endif
    BST(20);
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

    BST(21);
    // ---- [X | 1]^T . dPre over all rows.  Both operands are stored TRANSPOSED ([feature][row], rows contiguous)
    // by the forward/backward kernel, so a lane's four k values are one 16-byte load; rows past the batch hold
    // dPre = 0 (and finite activations), so no k masking is needed up to rup16(rows).
    const int m = m0 + c, mm = min(m, fi - 1);                // A operand: output row m (input feature); m == fi -> ones
    const int n = n0 + c, nn = min(n, fo - 1);                // B operand: output col n
    const float* Ap = Ly.A + (long)mm * a.ldT + 4 * q;
    const float* Bp = Ly.B + (long)nn * a.ldT + 4 * q;
    const bool ones = (m == fi);
    const int rows16 = (rows + 15) & ~15;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int k0 = 0; k0 < rows16; k0 += 128) {               // 128 rows (8 k-blocks, 16 loads per lane) per batch
        f32x4 av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = min(k0 + 16 * j, rows16 - 16);      // clamped + unconditional
            av[j] = *reinterpret_cast<const f32x4*>(Ap + k);
            bv[j] = *reinterpret_cast<const f32x4*>(Bp + k);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool kin = (k0 + 16 * j) < rows16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[j][e] = ones ? 1.0f : av[j][e];            // virtual ones column -> bias gradient
                bv[j][e] = kin ? bv[j][e] : 0.0f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                MFMA16(av[4 * jj + 0][e], bv[4 * jj + 0][e], acc0); MFMA16(av[4 * jj + 1][e], bv[4 * jj + 1][e], acc1);
                MFMA16(av[4 * jj + 2][e], bv[4 * jj + 2][e], acc2); MFMA16(av[4 * jj + 3][e], bv[4 * jj + 3][e], acc3);
            }
    }

    BST(22);
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
    BST(23);
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
    BST(24);
}

int dw_opt_blocks(int total_tiles) { return (total_tiles + 3) / 4; }

void launch_fused_dw_opt(hipStream_t s, const DwOptArgs& a) {
    hipLaunchKernelGGL(fused_dw_opt_kernel, dim3(dw_opt_blocks(a.total_tiles)), dim3(BTHREADS), 0, s, a);
}

}  // namespace smrt
