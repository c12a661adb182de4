// The forward / loss / backward launch of the two-kernel step of LDS-resident nets (see fused_small.hip), as a body that is
// instantiated per stage-sequence form (fused_stage.hpp: SeqGeneric / SeqVae1) -- the generic kernel in fused_small.hip, the ones
// written out per net shape in persist_seq.hip.
#pragma once
#include "fused_stage.hpp"

namespace smrt {

template <class Seq>
__device__ __forceinline__ void fused_fwd_bwd_body(const FusedArgs& a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float red[FWAVES], red2[FWAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * FTM;
    const int rows_here = min(FTM, a.rows - r0);
    const int Z = a.Z;
#ifdef SMRT_STAMPS
    int stamp_i = 0;
#endif
    STAMP();

    // ---- issue every global load this kernel depends on, in one batch.  The weights arrive by LDS-DMA: the
    // parameter buffer has a twin laid out exactly like this kernel's LDS weight image (k-quad interleaved, zero
    // padded; kept current by the optimizer epilogue), so staging is a straight 16-byte-per-lane copy with no
    // VGPRs and no ds_write pass.  Kernel arguments live in memory too (the first scalar load of each cache
    // line is a full round trip): the stage descriptors are fetched now, not one per stage.
    const uint32_t step = a.st->step;                      // scalar load: not queued behind the vector loads below
    {
        const int n4 = a.img_total >> 2;                   // multiple of 64: every wave-instruction is full
        for (int i = tid; i < n4; i += FTHREADS)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(a.Wimg + 4L * i), (lds_ptr_t)(lds + 4 * (i - lane)), 16, 0, 0);
    }
    float* xs = lds + a.lds_x;
    const int xf4_per_row = (a.D + 3) >> 2;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool x_fast = (FTM * xf4_per_row <= FTHREADS) && ((a.D & 3) == 0);   // one float4 per lane covers the tile
    if (x_fast && tid < rows_here * xf4_per_row) {
        const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
        xv = *reinterpret_cast<const float4*>(a.X + (long)(r0 + r) * a.ldx + c4);
    }
    float eps_pre[2] = {0.f, 0.f};                         // injected epsilon: fetched with the other loads
    if (a.eps_in) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * FTHREADS;
            if (i < rows_here * Z) eps_pre[u] = a.eps_in[(long)r0 * Z + i];
        }
    }
    {
        // zero the activation part of the LDS (pads must read as 0) while the loads are in flight
        float4* l4 = reinterpret_cast<float4*>(lds);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = (a.img_total >> 2) + tid; i < a.lds_total / 4; i += FTHREADS) l4[i] = z4;
        lds_barrier();
    }
    // Epsilon into its LDS tile.  Injected values were fetched with the other loads (the reparameterisation below must not
    // contain a global load, or the compiler puts an s_waitcnt vmcnt(0) in front of it that also waits for every
    // activation store still draining).  The kernel's own epsilon (Philox, indexed by step and global row) depends on nothing that is still in flight: it is
    // drawn HERE, while the weight image is on its way, instead of on the critical path between the heads and the decoder
    // (ten Philox rounds + Box-Muller are ~300 dependent VALU instructions).
    {
        float* es0 = lds + a.lds_e;
        for (int i = tid, u = 0; i < rows_here * Z; i += FTHREADS, ++u) {
            const int r = i / Z, k = i - r * Z;
            float e;
            if (a.eps_in) e = (u == 0) ? eps_pre[0] : (u == 1 ? eps_pre[1] : a.eps_in[(long)r0 * Z + i]);
            else e = philox_normal1(a.seed, STREAM_EPS, step, (uint64_t)(a.row0_global + r0) * Z + i);
            es0[r * a.s_z + k] = e;
        }
    }
    // X tile into LDS (zeros on padded rows)
    if (x_fast) {
        if (tid < FTM * xf4_per_row) {
            const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
            *reinterpret_cast<float4*>(xs + r * a.s_x + c4) = xv;
        }
    } else {
        for (int i = tid; i < FTM * a.D; i += FTHREADS) {
            const int r = i / a.D, cc = i - r * a.D;
            xs[r * a.s_x + cc] = (r < rows_here) ? a.X[(long)(r0 + r) * a.ldx + cc] : 0.0f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // every load above has landed (LDS-DMA included)
    __builtin_amdgcn_s_barrier();
    // blocked transposed copy of the X tile for the weight-gradient kernel: stores issued after the wait, so
    // nothing on the critical path waits for them
    {
        float* xt = a.XT + (long)blockIdx.x * ((a.D + 15) & ~15) * 16;
        for (int i = tid; i < FTM * a.D; i += FTHREADS) {
            const int cc = i >> 4, r = i & 15;
            xt[i] = xs[r * a.s_x + cc];                     // i == cc * 16 + r: fully coalesced
        }
    }
    STAMP();

    float* zs = lds + a.lds_z;
    float* es = lds + a.lds_e;
    float bce = 0.f, kl = 0.f;
    const float lo = a.lo, hi = a.hi, inv_batch = a.inv_batch;
    const int s_z = a.s_z;

    // ---- every GEMM stage (forward layers, then backward through the activations) runs through this one body
    StageCtx sx;
#ifdef SMRT_STAMPS
    sx.sdbg = nullptr;
#endif
    sx.lds = lds; sx.wave = wave; sx.c = c; sx.q = q; sx.tid = tid; sx.rows_here = rows_here; sx.Z = Z; sx.s_z = s_z;
    sx.lo = lo; sx.hi = hi; sx.inv_batch = inv_batch; sx.es = es; sx.zs = zs; sx.Zb_g = a.Zb_g; sx.blk = blockIdx.x;
    stage_rec_t* const stage_tab = (stage_rec_t*)(unsigned long long)a.stage_u;
    Seq::run(a.n_stages, stage_tab, sx, bce, kl, [&](int) { STAMP_NS(); });
    STAMP();

    // both partial sums in one pass: wave shuffles, one LDS exchange, fixed summation order
    bce = wave_sum(bce); kl = wave_sum(kl);
    if (lane == 0) { red[wave] = bce; red2[wave] = kl; }
    lds_barrier();                              // LDS only: the last stage's activation stores keep draining
    if (tid == 0) {
        float bsum = 0.f, ksum = 0.f;
#pragma unroll
        for (int i = 0; i < FWAVES; ++i) { bsum += red[i]; ksum += red2[i]; }
        a.bce_p[blockIdx.x] = bsum; a.kl_p[blockIdx.x] = ksum;
    }
    STAMP();
}


}  // namespace smrt
