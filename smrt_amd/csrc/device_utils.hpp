// Device helpers shared by the VAE kernels: activations, Philox4x32-10, wave/block reductions.
// gfx950 only (64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smrt {

enum Act : int { ACT_ELU = 0, ACT_IDENTITY = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

// 1 / (1 + e^-v) on the transcendental unit: v_exp_f32 + v_rcp_f32 (1 ulp each; saturates to exactly 0 / 1
// like the IEEE expression).  The library expf + correctly rounded divide cost ~25 VALU instructions per
// value, which dominates the epilogue of the latency-bound small-net kernels.
__device__ __forceinline__ float fast_sigmoid(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// tf.maximum(t, lo) / tf.minimum(t, hi) as TF-1.x's CPU kernels evaluate them (Eigen scalar_max_op / scalar_min_op are
// std::max / std::min: "(a < b) ? b : a" and "(b < a) ? b : a" with a = t): a NaN t STAYS NaN, where fmaxf / fminf would
// return the bound.  tf.clip_by_value(t, lo, hi) = minimum(maximum(t, lo), hi) (_ae_utils.py:31, base.py:144) therefore
// hands a NaN gradient on to the weights -- once p == 1.0f has produced inf * 0 in the output layer the
// reference's parameters, costs and generated rows are NaN from then on, and so are ours (the oracle's np.clip agrees).
__device__ __forceinline__ float tf_maximum(float t, float lo) { return (t < lo) ? lo : t; }
__device__ __forceinline__ float tf_minimum(float t, float hi) { return (hi < t) ? hi : t; }
__device__ __forceinline__ float tf_clip(float t, float lo, float hi) { return tf_minimum(tf_maximum(t, lo), hi); }

// Forward activations (reference: smrt/autoencode/autoencoder.py:28-34 -> tf.nn.*).
__device__ __forceinline__ float act_fwd(int act, float v) {
    switch (act) {
        case ACT_SIGMOID:  return fast_sigmoid(v);
        case ACT_TANH:     return tanhf(v);
        case ACT_RELU:     return fmaxf(v, 0.0f);
        case ACT_ELU:      return v > 0.0f ? v : expm1f(v);
        default:           return v;
    }
}

// Activation gradients as functions of the layer OUTPUT y (what TF-1.x's *Grad kernels use).
__device__ __forceinline__ float act_bwd(int act, float y, float dy) {
    switch (act) {
        case ACT_SIGMOID:  return dy * y * (1.0f - y);
        case ACT_TANH:     return dy * (1.0f - y * y);
        case ACT_RELU:     return y > 0.0f ? dy : 0.0f;
        case ACT_ELU:      return y < 0.0f ? dy * (y + 1.0f) : dy;
        default:           return dy;
    }
}

// Four values at once with ONE (uniform) switch: lets the compiler interleave the four dependency chains.
__device__ __forceinline__ void act_fwd4(int act, float (&v)[4]) {
    switch (act) {
        case ACT_SIGMOID:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fast_sigmoid(v[r]);
            break;
        case ACT_TANH:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            break;
        case ACT_RELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            break;
        case ACT_ELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : expm1f(v[r]);
            break;
        default: break;
    }
}

__device__ __forceinline__ void act_bwd4(int act, const float (&y)[4], float (&dy)[4]) {
    switch (act) {
        case ACT_SIGMOID:
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[r] = dy[r] * y[r] * (1.0f - y[r]);
            break;
        case ACT_TANH:
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[r] = dy[r] * (1.0f - y[r] * y[r]);
            break;
        case ACT_RELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[r] = y[r] > 0.0f ? dy[r] : 0.0f;
            break;
        case ACT_ELU:
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[r] = y[r] < 0.0f ? dy[r] * (y[r] + 1.0f) : dy[r];
            break;
        default: break;
    }
}

// ---------------------------------------------------------------- optimizer arithmetic (per-layer path)
// One element of tf.train.*Optimizer.apply_gradients as the reference constructs them (autoencoder.py:38-48, base.py:138-149;
// [TF-sem] slot initial values and epsilon placement are in api.hip: smrt_vae_reset_opt_state and SURVEY 8 a-9), with IEEE
// sqrt and divide.  Shared by the optimizer sweep (vae_kernels.hip) and the fused weight-gradient + update launch
// (small_gemm.hip), so both give the same bits for the same gradient.
__device__ __forceinline__ float opt_update_ieee(int opt, float lr, float alpha, float w, float g, float& s0, float& s1) {
    switch (opt) {
        case 1: {   // RMSProp: decay .9, momentum 0, eps 1e-10, rms slot starts at 1
            s0 = s0 + (g * g - s0) * (1.0f - 0.9f);
            s1 = s1 * 0.0f + lr * g / sqrtf(s0 + 1e-10f);
            return w - s1;
        }
        case 2: {   // Adam: alpha = lr * sqrt(1 - b2^t) / (1 - b1^t), eps 1e-8 outside the sqrt
            s0 = s0 + (g - s0) * (1.0f - 0.9f);
            s1 = s1 + (g * g - s1) * (1.0f - 0.999f);
            return w - alpha * s0 / (sqrtf(s1) + 1e-8f);
        }
        case 3: case 6: {   // Adagrad / ProximalAdagrad(l1 = l2 = 0): accumulator starts at 0.1
            s0 = s0 + g * g;
            return w - lr * g / sqrtf(s0);
        }
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

// L2-free optimizer arithmetic of one element / of four (base.py:135-149, autoencoder.py:38-48); shared by every kernel that
// applies an update in its epilogue.
__device__ __forceinline__ float opt_apply(int opt, float lr, float alpha, float w, float g, float& s0, float& s1) {
    switch (opt) {
        case 1:     // RMSProp: decay .9, momentum 0, eps 1e-10 (rms slot starts at 1); v_rsq_f32 is 1 ulp
            s0 = s0 + (g * g - s0) * (1.0f - 0.9f);
            s1 = s1 * 0.0f + lr * g * __builtin_amdgcn_rsqf(s0 + 1e-10f);
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

// Four elements at once with ONE (uniform) switch around them.
__device__ __forceinline__ void opt_apply4(int opt, float lr, float alpha, const float (&w)[4], const float (&g)[4],
                                           float (&s0)[4], float (&s1)[4], float (&wn)[4]) {
    switch (opt) {
#define SMRT_OPT_CASE(K) _Pragma("unroll") for (int r = 0; r < 4; ++r) wn[r] = opt_apply(K, lr, alpha, w[r], g[r], s0[r], s1[r]); break
        case 1: SMRT_OPT_CASE(1);
        case 2: SMRT_OPT_CASE(2);
        case 3: case 6: SMRT_OPT_CASE(3);
        case 4: SMRT_OPT_CASE(4);
        default: SMRT_OPT_CASE(0);
#undef SMRT_OPT_CASE
    }
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG (Salmon et al., SC'11).  Streams: 1 = training epsilon, 2 = dropout, 3 = generate.
// Mirrored bit-for-bit by oracle/vae_oracle.py: philox4x32_10.
enum : uint32_t { STREAM_EPS = 1, STREAM_DROPOUT = 2, STREAM_GENERATE = 3 };

struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (written as 32 x 32 -> 64 products, hipcc emits v_mad_u64_u32 pairs that run SLOWER than mul_hi + mul_lo here:
        // the persistent training kernel's epsilon draw went from hidden to exposed, 2.68 -> 2.81 ms per C2 epoch)
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// sin and cos of 2 pi u for u = k / 2^24 in [0, 1).  In revolutions the quadrant reduction is EXACT (4 u - rint(4 u) has no
// rounding), which leaves one rounding in the angle and the two cephes minimax polynomials on [-pi/4, pi/4]: ~1 ulp, about
// 25 VALU instructions for the pair against ~150 for libm's sinf + cosf with their general range reduction.
// Every operation is spelled out (fmaf / fmul_rn): the same bits wherever this is inlined, whatever the compiler would have
// contracted there.
__device__ __forceinline__ void sincos_rev(float u, float& s, float& c) {
    const float x = 4.0f * u;
    const float qf = rintf(x);                                        // 0 .. 4
    const float a = __fmul_rn(x - qf, 1.5707963267948966f);           // [-pi/4, pi/4]
    const float a2 = __fmul_rn(a, a);
    float sp = -1.9515295891e-4f;
    sp = __fmaf_rn(sp, a2, 8.3321608736e-3f);
    sp = __fmaf_rn(sp, a2, -1.6666654611e-1f);
    const float sn = __fmaf_rn(__fmul_rn(a, a2), sp, a);
    float cp = 2.443315711809948e-5f;
    cp = __fmaf_rn(cp, a2, -1.388731625493765e-3f);
    cp = __fmaf_rn(cp, a2, 4.166664568298827e-2f);
    const float cs = __fmaf_rn(__fmul_rn(a2, a2), cp, __fmaf_rn(-0.5f, a2, 1.0f));
    const int qi = (int)qf & 3;                                       // angle = qi * pi/2 + a
    s = (qi & 1) ? cs : sn;
    c = (qi & 1) ? sn : cs;
    if (qi == 1 || qi == 2) c = -c;
    if (qi >= 2) s = -s;
}

__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& n0, float& n1) {
    const float u1 = ((float)(xa >> 8) + 1.0f) * 5.9604644775390625e-08f;   // (0,1], exact in fp32
    const float u2 = (float)(xb >> 8) * 5.9604644775390625e-08f;            // [0,1)
    const float r = __builtin_amdgcn_sqrtf(__fmul_rn(-2.0f, logf(u1)));     // v_sqrt_f32: 1 ulp
    float sn, cs;
    sincos_rev(u2, sn, cs);
    n0 = __fmul_rn(r, cs);
    n1 = __fmul_rn(r, sn);
}

// loc + scale * n of a drawn normal, the same bits in every kernel that applies it
__device__ __forceinline__ float shift_scale(float loc, float scale, float n) { return __fmaf_rn(scale, n, loc); }

// Standard normal for flat element index e of (stream, step).
__device__ __forceinline__ float philox_normal1(uint64_t seed, uint32_t stream, uint32_t step, uint64_t e) {
    const uint64_t q = e >> 2;
    const Philox4 p = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, stream,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    float a, b;
    const int lane = (int)(e & 3);
    if (lane < 2) box_muller(p.x, p.y, a, b); else box_muller(p.z, p.w, a, b);
    return (lane & 1) ? b : a;
}

__device__ __forceinline__ float philox_uniform1(uint64_t seed, uint32_t stream, uint32_t step, uint64_t e) {
    const uint64_t q = e >> 2;
    const Philox4 p = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, stream,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    const int lane = (int)(e & 3);
    const uint32_t x = lane == 0 ? p.x : lane == 1 ? p.y : lane == 2 ? p.z : p.w;
    return (float)(x >> 8) * 5.9604644775390625e-08f;
}

// ---------------------------------------------------------------- raw buffer descriptors
// A buffer load takes a per-lane byte offset (VGPR) plus a wave-uniform one (SGPR) from the descriptor's base, and returns 0
// for offsets at or beyond `bytes` instead of faulting: row clamps and 64-bit per-lane addresses both go away.
constexpr unsigned RSRC_WORD3 = 0x00020000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, RSRC_WORD3);
}

// ---------------------------------------------------------------- reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // lane 0 holds the sum
}

// Sum over a 256-thread block in a fixed order; result valid in thread 0.  `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Same for a block of NW waves (fixed summation order); `red` = NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

}  // namespace smrt
