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

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based RNG (Salmon et al., SC'11).  Streams: 1 = training epsilon, 2 = dropout, 3 = generate.
// Mirrored bit-for-bit by oracle/vae_oracle.py: philox4x32_10.
enum : uint32_t { STREAM_EPS = 1, STREAM_DROPOUT = 2, STREAM_GENERATE = 3 };

struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& n0, float& n1) {
    const float u1 = ((float)(xa >> 8) + 1.0f) * 5.9604644775390625e-08f;   // (0,1], exact in fp32
    const float u2 = (float)(xb >> 8) * 5.9604644775390625e-08f;            // [0,1)
    const float r = sqrtf(-2.0f * logf(u1));
    const float th = 6.283185307179586f * u2;
    n0 = r * cosf(th);
    n1 = r * sinf(th);
}

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
