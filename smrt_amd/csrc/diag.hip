// Achievable-peak and latency-floor measurements for the roofline figures ("measure achievable peaks
// with a microbenchmark and quote both"; "report the latency floor at reference-default batch sizes").
// No counterpart in the reference; nothing on the product path calls these.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/smrt_hip.h"

extern int smrt_set_error(int code, const std::string& m);

#define DIAGC(call)                                                                                   \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) return smrt_set_error(-2, std::string(#call ": ") + hipGetErrorString(e__)); \
    } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));

// Every wave keeps four independent 32x32 accumulators in flight: 4 x v_mfma_f32_32x32x2_f32 per iteration
// (4 x 4096 FLOP), operands in registers, nothing but the matrix pipe is exercised.
__global__ void __launch_bounds__(256) diag_mfma_kernel(float* sink, int iters) {
    v16f acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    float a = 1.0f + 0.001f * threadIdx.x, b = 0.5f - 0.001f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (s == 12345.678f) sink[0] = s;                       // keeps the loop alive; practically never taken
}

// The same loop on v_mfma_f32_16x16x4_f32 (the shape the LDS-resident small-net kernels use): 4 x 2048 FLOP per iteration.
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) diag_mfma16_kernel(float* sink, int iters) {
    v4f acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    float a = 1.0f + 0.001f * threadIdx.x, b = 0.5f - 0.001f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 4; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (s == 12345.678f) sink[0] = s;
}

__global__ void __launch_bounds__(256) diag_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) diag_read_kernel(const float4* __restrict__ src, float* sink, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = src[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}

// The smallest kernel that still depends on its predecessor: one word per workgroup, read-modify-write.
__global__ void diag_touch_kernel(unsigned* p) {
    if (threadIdx.x == 0) p[blockIdx.x] += 1u;
}

static float best_ms(hipStream_t s, int reps, const std::function<hipError_t()>& body, hipError_t* err) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(a, s);
        hipError_t e = body();
        (void)hipEventRecord(b, s);
        hipError_t e2 = hipEventSynchronize(b);
        if (e != hipSuccess || e2 != hipSuccess) { *err = e != hipSuccess ? e : e2; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
}

static int mfma_peak(int32_t device, bool small_shape, double* tflops) {
    if (!tflops) return smrt_set_error(-1, "smrt_diag_mfma_peak: null output");
    DIAGC(hipSetDevice(device));
    hipDeviceProp_t prop;
    DIAGC(hipGetDeviceProperties(&prop, device));
    hipStream_t s;
    DIAGC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* sink = nullptr;
    DIAGC(hipMalloc((void**)&sink, 256));
    const int iters = 1 << 15;
    const int blocks = prop.multiProcessorCount * 2;         // 2 x 4 waves per CU: two waves per SIMD
    hipError_t err = hipSuccess;
    const float ms = best_ms(s, 4, [&]() {
        if (small_shape) hipLaunchKernelGGL(diag_mfma16_kernel, dim3(blocks), dim3(256), 0, s, sink, iters);
        else hipLaunchKernelGGL(diag_mfma_kernel, dim3(blocks), dim3(256), 0, s, sink, iters);
        return hipGetLastError();
    }, &err);
    (void)hipFree(sink); (void)hipStreamDestroy(s);
    if (err != hipSuccess) return smrt_set_error(-2, std::string("smrt_diag_mfma_peak: ") + hipGetErrorString(err));
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4.0 * (small_shape ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2);
    *tflops = flop / (ms * 1e-3) / 1e12;
    return 0;
}

extern "C" int smrt_diag_mfma_peak(int32_t device, double* tflops) { return mfma_peak(device, false, tflops); }
extern "C" int smrt_diag_mfma_peak_16x16x4(int32_t device, double* tflops) { return mfma_peak(device, true, tflops); }

extern "C" int smrt_diag_hbm_bandwidth(int32_t device, int64_t bytes, double* copy_gbs, double* read_gbs) {
    if (bytes < (1 << 20) || !copy_gbs || !read_gbs) return smrt_set_error(-1, "smrt_diag_hbm_bandwidth: bad arguments");
    DIAGC(hipSetDevice(device));
    hipStream_t s;
    DIAGC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long n4 = bytes / 16;
    float4 *src = nullptr, *dst = nullptr;
    DIAGC(hipMalloc((void**)&src, n4 * 16));
    DIAGC(hipMalloc((void**)&dst, n4 * 16));
    DIAGC(hipMemsetAsync(src, 0, n4 * 16, s));
    DIAGC(hipMemsetAsync(dst, 0, n4 * 16, s));
    hipError_t err = hipSuccess;
    const int blocks = 256 * 16;
    const float ms_c = best_ms(s, 5, [&]() {
        hipLaunchKernelGGL(diag_copy_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)src, dst, n4);
        return hipGetLastError();
    }, &err);
    const float ms_r = best_ms(s, 5, [&]() {
        hipLaunchKernelGGL(diag_read_kernel, dim3(blocks), dim3(256), 0, s, (const float4*)src, (float*)dst, n4);
        return hipGetLastError();
    }, &err);
    (void)hipFree(src); (void)hipFree(dst); (void)hipStreamDestroy(s);
    if (err != hipSuccess) return smrt_set_error(-2, std::string("smrt_diag_hbm_bandwidth: ") + hipGetErrorString(err));
    *copy_gbs = 2.0 * n4 * 16 / (ms_c * 1e-3) / 1e9;           // read + write
    *read_gbs = 1.0 * n4 * 16 / (ms_r * 1e-3) / 1e9;
    return 0;
}

extern "C" int smrt_diag_step_floor(int32_t device, int32_t kernels_per_step, int32_t steps, int32_t workgroups,
                                    int32_t threads, double* us_per_step) {
    if (kernels_per_step < 1 || steps < 1 || workgroups < 1 || workgroups > 65536 || threads < 64 || threads > 1024 ||
        (long)kernels_per_step * steps > 200000 || !us_per_step)
        return smrt_set_error(-1, "smrt_diag_step_floor: bad arguments");
    DIAGC(hipSetDevice(device));
    hipStream_t s;
    DIAGC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned* p = nullptr;
    DIAGC(hipMalloc((void**)&p, (size_t)workgroups * 4));
    DIAGC(hipMemsetAsync(p, 0, (size_t)workgroups * 4, s));
    DIAGC(hipStreamSynchronize(s));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    DIAGC(hipGraphCreate(&graph, 0));
    hipGraphNode_t prev = nullptr;
    for (long i = 0; i < (long)kernels_per_step * steps; ++i) {
        void* args[] = {(void*)&p};
        hipKernelNodeParams kp{};
        kp.func = (void*)diag_touch_kernel;
        kp.gridDim = dim3(workgroups); kp.blockDim = dim3(threads); kp.sharedMemBytes = 0;
        kp.kernelParams = args; kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        DIAGC(hipGraphAddKernelNode(&node, graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
        prev = node;
    }
    DIAGC(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipError_t err = hipSuccess;
    (void)hipGraphLaunch(exec, s);                            // warm-up (uploads the graph)
    const float ms = best_ms(s, 5, [&]() { return hipGraphLaunch(exec, s); }, &err);
    unsigned first = 0;
    (void)hipMemcpy(&first, p, 4, hipMemcpyDeviceToHost);
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
    (void)hipFree(p); (void)hipStreamDestroy(s);
    if (err != hipSuccess) return smrt_set_error(-2, std::string("smrt_diag_step_floor: ") + hipGetErrorString(err));
    if (first != 6u * (unsigned)kernels_per_step * (unsigned)steps)
        return smrt_set_error(-4, "smrt_diag_step_floor: the chain did not run every launch");
    *us_per_step = 1e3 * ms / steps;
    return 0;
}
