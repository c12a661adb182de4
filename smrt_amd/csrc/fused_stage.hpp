// Shared pieces of the LDS-resident small-net training kernels: the per-launch two-kernel step (fused_small.hip) and the
// persistent whole-epoch kernel (persist_train.hip) run their forward / backward GEMM stages through the SAME stage body.
#pragma once
#include "vae_kernels.hpp"
#include "device_utils.hpp"

#include <type_traits>

namespace smrt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
// The stage records carry their output block as two 32-bit halves; a pointer rebuilt from integers is a GENERIC pointer
// to the compiler, and stores through it are flat_store: they count on lgkmcnt as well (every later LDS wait becomes
// lgkmcnt(0)) and issue through both address paths.  These types keep the stores global_store.
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f32x4 gf32x4;

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
//
// Global activations / gradients for the weight-gradient kernel are "blocked transposed": [row block of 16][feature][16
// rows].  A lane's four rows of one column are one 16-byte store, a wave's store covers 1 KiB contiguous, and the
// weight-gradient kernel's MFMA fragments are 1 KiB contiguous loads of the same buffer.

#ifdef SMRT_STAMPS
#define STAMP() do { __syncthreads(); if (a.dbg && blockIdx.x == 0 && tid == 0) { a.dbg[stamp_i] = wall_clock64(); a.dbg[32 + stamp_i] = clock64(); } ++stamp_i; } while (0)
#define STAMP_NS() do { if (a.dbg && blockIdx.x == 0 && tid == 0 && stamp_i < 31) { a.dbg[stamp_i] = wall_clock64(); a.dbg[32 + stamp_i] = clock64(); } ++stamp_i; } while (0)
#else
#define STAMP() do { } while (0)
#define STAMP_NS() do { } while (0)
#endif

// NB 16-wide k-blocks of one 16x16 output tile: every LDS fragment read first, then the MFMAs (4 independent chains).
// ap: this lane's A fragment (activations [row c][k = 4 q ..], k contiguous: one ds_read_b128 per k-block).
// bp: this lane's weight fragment; !TRANS: one ds_read_b128 per k-block, `bstep` floats apart;
//     TRANS: the lane's four k values are 4 floats apart (k-quad interleaved image), k-blocks 64 floats apart.
template <int NB, bool TRANS>
__device__ __forceinline__ void k_chunk(const float* ap, const float* bp, int bstep,
                                        f32x4& acc0, f32x4& acc1, f32x4& acc2, f32x4& acc3) {
    f32x4 av[NB], bv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        // (the two layouts request their operands in opposite order: with a common first instruction hipcc hoists that
        // read above the branch that selects the layout, and a read pending from another block makes it wait for ALL reads)
        if constexpr (TRANS) {
            const float* p = bp + 64 * j;
            bv[j][0] = p[0]; bv[j][1] = p[4]; bv[j][2] = p[8]; bv[j][3] = p[12];
            av[j] = *reinterpret_cast<const f32x4*>(ap + 16 * j);
        } else {
            av[j] = *reinterpret_cast<const f32x4*>(ap + 16 * j);
            bv[j] = *reinterpret_cast<const f32x4*>(bp + j * bstep);
        }
        // Pin "reads in consumption order, all requested before the first MFMA".  Left alone, hipcc either sinks each
        // LDS read to just before its MFMA or issues the reads in an order of its own, and then the first MFMA waits
        // for (nearly) all of them; pinned, the waits count down lgkmcnt(2 NB - 2), (2 NB - 4), ...
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        MFMA16(av[j][0], bv[j][0], acc0); MFMA16(av[j][1], bv[j][1], acc1);
        MFMA16(av[j][2], bv[j][2], acc2); MFMA16(av[j][3], bv[j][3], acc3);
    }
}

#define UNI(x) __builtin_amdgcn_readfirstlane(x)


// A stage's 16-byte output store: 0 none (everything stays in LDS), 1 plain global store (the next KERNEL reads it),
// 2 write-through `sc1` store (another workgroup of the SAME launch reads it: cdna guide, Guideline 16, R1).
template <int MODE>
__device__ __forceinline__ void stage_store(gfloat* p, f32x4 v) {
    if constexpr (MODE == 1) *(gf32x4*)p = v;
    if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");   // (s_nop: the store-data hazard hipcc would pad itself)
}

// What a stage needs to know about its workgroup.
struct StageCtx {
    float* lds;
    int wave, c, q, tid;
    int rows_here;            // valid rows of this workgroup's 16-row tile
    int Z, s_z;
    float lo, hi, inv_batch;
    const float* es;          // epsilon tile in LDS
    float* zs;                // z tile in LDS
    float* Zb_g;              // TO_GLOBAL only: z, blocked transposed
    long blk;                 // TO_GLOBAL only: this workgroup's 16-row block inside the global activation buffers
#ifdef SMRT_STAMPS
    unsigned long long* sdbg; // diagnostic builds: 4 shader-clock stamps per stage (entry, addresses ready, sums ready, stored)
#endif
};
#ifdef SMRT_STAMPS
#define XST(i) do { if (x.sdbg && x.tid == 0) x.sdbg[i] = clock64(); } while (0)
#else
#define XST(i) do { } while (0)
#endif

// ONE GEMM stage (forward through a layer, or backward through it) of a 16-row tile runs in two parts:
//   decode: the stage's record -> StageDec: the uniform scalars and this lane's addresses for the wave's first 16-column
//           block.  decode_stage reads the 80-byte FusedStage from registers (the per-launch kernel: ~170 instructions);
//           decode_stage_u takes the host-decoded StageU that one scalar load left in SGPRs (the persistent kernel, which
//           runs the same stages thousands of times per launch: ~25 instructions);
//   exec_stage: fragments -> MFMAs -> fused epilogue, for the wave's column blocks n0 = 16 wave, + 128, ...
// TO_GLOBAL (1 plain, 2 write-through stores, 0 nothing): also stream the stage's output to the blocked-transposed global
// buffer (the weight-gradient side reads it).  The caller runs lds_barrier() after exec_stage, then (after a FK_FWD_HEADS
// stage) reparam_after_heads().
struct StageDec {
    int nb_all, nn, N, N16, kind, act, trans, s_out, s_aux, Zq;                     // (uniform)
    int out_off, w_off;                                                             // (uniform)
    const float* ap;          // this lane's A fragment base (activations)
    const float* bp;          // this lane's weight fragment base, first column block
    float* o;                 // this lane's output element (row 4 q, column n) in the LDS tile, first column block
    gfloat* g;                // the same in the global blocked-transposed buffer
    const float* po;          // auxiliary tile element (x / producing layer's output), first column block
    const float* pb;          // bias image element, first column block
    int n;                    // this lane's column in the first block
};

// SK / SNB / SACT: the stage's kind, its number of 16-wide k-blocks and its activation when they are known at compile time
// (-1 / 0 / -1: read from the record).  With them the body is straight-line code: no dispatch on the kind, no chunk ladder,
// no activation switch.
template <int SK = -1, int SNB = 0, int SACT = -1>
__device__ __forceinline__ StageDec decode_stage(const int4 d0, const int4 d1, const int4 d2, const int4 d3, const int4 d4,
                                                 const StageCtx& x) {
    float* const lds = x.lds;
    const int wave = x.wave, c = x.c, q = x.q;
    const int in_off = UNI(d0.x), sin = UNI(d0.y), K16 = SNB > 0 ? 16 * SNB : UNI(d0.z), w_off = UNI(d0.w);
    const int nn = UNI(d1.x), N = UNI(d1.y), kind = SK >= 0 ? SK : UNI(d1.z), act = SACT >= 0 ? SACT : UNI(d1.w);
    const int out_off = UNI(d2.x), s_out = UNI(d2.y), b_off = UNI(d2.z), aux_off = UNI(d2.w);
    const int s_aux = UNI(d3.x), NF = UNI(d3.y), trans = SK >= 0 ? (SK >= FK_BWD_ACT ? 1 : 0) : UNI(d3.z);
    gfloat* gblk = (gfloat*)(((unsigned long long)(unsigned)UNI(d4.y) << 32) | (unsigned)UNI(d4.x)) + x.blk * NF * 16;
    StageDec s;
    s.nb_all = K16 >> 4; s.nn = nn; s.N = N; s.N16 = (N + 15) & ~15; s.kind = kind; s.act = act; s.trans = trans;
    s.s_out = s_out; s.s_aux = s_aux; s.out_off = out_off; s.w_off = w_off;
    const int n = wave * 16 + c;
    s.n = n;
    s.ap = lds + in_off + c * sin + 4 * q;
    s.bp = trans ? lds + w_off + ((((n >> 2) * nn) + 4 * q) << 2) + (n & 3)     // Wop[k][n] = W[kf = n][nf = k]
                 : lds + w_off + ((q * nn + n) << 2);                          // Wop[k][n] = W[kf = k][nf = n]
    s.o = lds + out_off + (4 * q) * s_out + n;
    s.g = gblk + n * 16 + 4 * q;
    s.po = lds + aux_off + (4 * q) * s_aux + n;
    s.pb = lds + b_off + 4 * n;
    s.Zq = (4 * q) * x.s_z;
    return s;
}

// SNT: the stage's padded output width in 16-column blocks (0: from the record).  With it a wave knows statically whether it
// has a column block, and (with SNB) the image's column count is an immediate, so are the weight fragments' k-block offsets.
template <int TO_GLOBAL, int SK = -1, int SNB = 0, int SACT = -1, int SNT = 0>
__device__ __forceinline__ int exec_stage(const StageDec& s, const StageCtx& x, float& bce) {
    float* const lds = x.lds;
    const int wave = x.wave, q = x.q, tid = x.tid, rows_here = x.rows_here, Z = x.Z, s_z = x.s_z;
    const float lo = x.lo, hi = x.hi, inv_batch = x.inv_batch;
    const float* const es = x.es;
    const int kind = SK >= 0 ? SK : s.kind, act = SACT >= 0 ? SACT : s.act;
    const int trans = SK >= 0 ? (SK >= FK_BWD_ACT ? 1 : 0) : s.trans;
    const int N = s.N, N16 = SNT > 0 ? 16 * SNT : s.N16, s_out = s.s_out, s_aux = s.s_aux;
    // the image's column count is rup16(fan-out) + 1: the padded OUTPUT width forward, the padded K extent backward
    constexpr bool NN_STATIC = SK >= 0 && ((SK >= FK_BWD_ACT) ? SNB > 0 : SNT > 0);
    const int nn = NN_STATIC ? 16 * ((SK >= FK_BWD_ACT) ? SNB : SNT) + 1 : s.nn;
    const int bstep = trans ? 64 : (nn << 4);
    XST(0);
    int n = s.n;
    const float* bp0 = s.bp;
    if constexpr (NN_STATIC)        // (the same address as decode's, with nn an immediate)
        bp0 = trans ? lds + (((__mul24(n >> 2, nn) + 4 * q) << 2) + (n & 3) + s.w_off) : lds + (((q * nn + n) << 2) + s.w_off);
    float* o = s.o;
    gfloat* g = s.g;
    const float* po = s.po;
    const float* pbias = s.pb;
    auto block = [&]() {        // one 16-column block of this wave
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        // k runs in chunks of NB 16-wide k-blocks, NB a compile-time 8 / 4 / 2 / 1: a chunk is straight-line code -- ALL its
        // fragments are requested first, then its 4 NB MFMAs run into 4 independent chains while the later reads are
        // still landing (the compiler counts lgkmcnt down block by block; with run-time guards around the blocks it
        // waited for every read before the first MFMA and copied the fragment registers around the guards).  Nets on
        // this path have K in {16, 32, 64, 128} almost always, i.e. exactly one chunk per stage.
        const float* apc = s.ap;
        const float* bp = bp0;
        int left = s.nb_all;
        XST(1);
        if constexpr (SNB > 0 && SK >= 0) {            // the whole k extent as ONE straight-line chunk
            k_chunk<SNB, (SK >= FK_BWD_ACT)>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            left = 0;
        }
#pragma unroll 1
        for (; left >= 8; left -= 8) {
            if (trans) k_chunk<8, true>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            else       k_chunk<8, false>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            apc += 128; bp += 8 * bstep;
        }
        if (left & 4) {
            if (trans) k_chunk<4, true>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            else       k_chunk<4, false>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            apc += 64; bp += 4 * bstep;
        }
        if (left & 2) {
            if (trans) k_chunk<2, true>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            else       k_chunk<2, false>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            apc += 32; bp += 2 * bstep;
        }
        if (left & 1) {
            if (trans) k_chunk<1, true>(apc, bp, bstep, acc0, acc1, acc2, acc3);
            else       k_chunk<1, false>(apc, bp, bstep, acc0, acc1, acc2, acc3);
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
#ifdef SMRT_STAMPS
        if (x.sdbg && x.tid == 0) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); x.sdbg[2] = clock64(); }
#endif
        // rows 4q .. 4q+3 of column n.  Workspace rows are padded to a multiple of 16, so rows past the batch are
        // computed and stored like the others (their dPre is forced to 0, nothing reads them).
        if (kind <= FK_FWD_HEADS) {
            // y = act(x.W + b)                                                     layer.py:368
            if (n < N) {
                const float bcol = *pbias;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bcol;
                act_fwd4(act, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r * s_out] = v[r];
                stage_store<TO_GLOBAL>(g, f32x4{v[0], v[1], v[2], v[3]});
            }
        } else if (kind == FK_FWD_BCE) {
            // sigmoid output + clipped Bernoulli NLL + dL/dpre (layer.py:333-338, _ae_utils.py:31-32)
            if (n < N) {
                const float bcol = *pbias;
                const float* xr = po;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (4 * q + r) < rows_here;
                    const float p = fast_sigmoid(v[r] + bcol);
                    const float xx = xr[r * s_aux];
                    const float pl = tf_maximum(p, lo), pc = tf_minimum(pl, hi), om = 1.0f - pc;
                    const float term = xx * __logf(pc) + (1.0f - xx) * __logf(om);
                    bce -= valid ? term : 0.0f;
                    // TF clip_by_value gradient gate: Maximum passes where p >= lo, Minimum where max(p,lo) <= hi
                    const bool pass = valid && (p >= lo) && (pl <= hi);
                    const float dp = pass ? -(xx * __builtin_amdgcn_rcpf(pc) - (1.0f - xx) * __builtin_amdgcn_rcpf(om)) * inv_batch : 0.0f;
                    v[r] = valid ? dp * p * (1.0f - p) : 0.0f;
                    o[r * s_out] = v[r];
                }
                stage_store<TO_GLOBAL>(g, f32x4{v[0], v[1], v[2], v[3]});
            }
        } else if (kind == FK_BWD_ACT) {
            // dPre_prev = (dPre . W^T) * act'(y_prev); pad columns come out as exact zeros (zero weights, zero y)
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = po[r * s_aux];
            act_bwd4(act, yv, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r * s_out] = v[r];
            stage_store<TO_GLOBAL>(g, f32x4{v[0], v[1], v[2], v[3]});
        } else {
            // FK_BWD_LATENT: v = dL/dz through the reparameterisation and the KL term into both heads
            //   dmu = dz + mu/B ; dls = dz*eps*exp(ls) + (exp(2 ls) - 1)/B     layer.py:279, _ae_utils.py:49
            if (n < N) {
                const float* er = es + s.Zq + n;
                float mu[4], ls[4], g_mu[4], g_ls[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mu[r] = po[r * s_aux];
                    ls[r] = po[r * s_aux + Z];
                    const float e = er[r * s_z];
                    const float el = __expf(ls[r]);
                    g_mu[r] = v[r] + mu[r] * inv_batch;
                    g_ls[r] = v[r] * e * el + (el * el - 1.0f) * inv_batch;
                }
                act_bwd4(act, mu, g_mu);
                act_bwd4(act, ls, g_ls);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (4 * q + r) < rows_here;
                    g_mu[r] = valid ? g_mu[r] : 0.0f;
                    g_ls[r] = valid ? g_ls[r] : 0.0f;
                    o[r * s_out] = g_mu[r];
                    o[r * s_out + Z] = g_ls[r];
                }
                stage_store<TO_GLOBAL>(g, f32x4{g_mu[0], g_mu[1], g_mu[2], g_mu[3]});
                stage_store<TO_GLOBAL>(g + Z * 16, f32x4{g_ls[0], g_ls[1], g_ls[2], g_ls[3]});
            }
        }
    };
    if constexpr (SNT > 0 && SNT <= FWAVES) {
        if (wave < SNT) block();                        // at most one block per wave
    } else {
#pragma unroll 1
        for (int n0 = wave * 16; n0 < N16; n0 += 16 * FWAVES) {
            block();
            n += 16 * FWAVES; bp0 += (trans ? 128 * nn : 512); o += 16 * FWAVES; g += 16 * FWAVES * 16; po += 16 * FWAVES; pbias += 4 * 16 * FWAVES;
        }
    }
    XST(3);
    if (kind == FK_BWD_LATENT) {
        // the next stage reads k up to rup16(2Z): clear the tail this stage did not write
        const int wrote = 2 * Z, tail = ((wrote + 15) & ~15) - wrote;
        for (int i = tid; i < FTM * tail; i += FTHREADS) {
            const int r = i / tail;
            lds[s.out_off + r * s_out + wrote + (i - r * tail)] = 0.f;
        }
    }
    return kind;
}

// decode + execute in one call (the per-launch two-kernel step)
template <int TO_GLOBAL>
__device__ __forceinline__ int run_stage(const int4 d0, const int4 d1, const int4 d2, const int4 d3, const int4 d4,
                                         const StageCtx& x, float& bce) {
    const StageDec s = decode_stage<>(d0, d1, d2, d3, d4, x);
    return exec_stage<TO_GLOBAL>(s, x, bce);
}

static_assert(STAGE_ACT_SIGMOID == ACT_SIGMOID, "stage_variant's activation id");
// The host-decoded record, as one scalar load left it in SGPRs -> StageDec.  24-bit multiplies: every factor is an LDS
// offset or a lane index.
typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ StageDec decode_stage_u(const i32x16 u, const StageCtx& x) {
    float* const lds = x.lds;
    const int c = x.c, q = x.q, n = x.wave * 16 + c;
    StageDec s;
    const int nn = u[4], packed = u[6];
    const int trans = (packed >> 16) & 255;
    s.nb_all = u[2]; s.nn = nn; s.N = u[5]; s.N16 = u[7]; s.kind = packed & 255; s.act = (packed >> 8) & 255; s.trans = trans;
    s.s_out = u[9]; s.s_aux = u[12]; s.out_off = u[8]; s.w_off = u[3];
    s.n = n;
    s.ap = lds + (__mul24(c, u[1]) + 4 * q + u[0]);
    s.bp = trans ? lds + (((__mul24(n >> 2, nn) + 4 * q) << 2) + (n & 3) + u[3])
                 : lds + (((__mul24(q, nn) + n) << 2) + u[3]);
    s.o = lds + (__mul24(4 * q, u[9]) + n + u[8]);
    s.po = lds + (__mul24(4 * q, u[12]) + n + u[11]);
    s.pb = lds + (4 * n + u[10]);
    gfloat* gout = (gfloat*)(((unsigned long long)(unsigned)u[15] << 32) | (unsigned)u[14]);
    s.g = gout + (x.blk * u[13] + (n * 16 + 4 * q));
    s.Zq = (4 * q) * x.s_z;
    return s;
}

// exec_stage through the variant the record names (vae_kernels.hpp: stage_variant): the structure-specialised copy of the
// body when there is one, else the generic body.  The switch is wave-uniform (one scalar jump).
template <int TO_GLOBAL>
__device__ __forceinline__ int exec_stage_variant(int variant, const StageDec& cur, const StageCtx& x, float& bce) {
    switch (variant) {
#define SMRT_SV1(K_, NB_) \
        case stage_variant(K_, 16 * NB_, ACT_SIGMOID): return exec_stage<TO_GLOBAL, K_, NB_, ACT_SIGMOID>(cur, x, bce); \
        case stage_variant(K_, 16 * NB_, -1):          return exec_stage<TO_GLOBAL, K_, NB_, -1>(cur, x, bce);
#define SMRT_SV(K_) SMRT_SV1(K_, 1) SMRT_SV1(K_, 2) SMRT_SV1(K_, 4) SMRT_SV1(K_, 8)
        SMRT_SV(FK_FWD_ACT) SMRT_SV(FK_FWD_HEADS) SMRT_SV(FK_FWD_BCE) SMRT_SV(FK_BWD_ACT) SMRT_SV(FK_BWD_LATENT)
#undef SMRT_SV
#undef SMRT_SV1
        default: return exec_stage<TO_GLOBAL>(cur, x, bce);
    }
}

// z = mu + eps * exp(log_sigma) and the KL row terms, after the heads stage's barrier; ends with its own barrier.
template <int TO_GLOBAL>
__device__ __forceinline__ void reparam_after_heads(const int4 d2, const StageCtx& x, float& kl) {
    float* const lds = x.lds;
    float* const zs = x.zs;
    const float* const es = x.es;
    const int tid = x.tid, rows_here = x.rows_here, Z = x.Z, s_z = x.s_z;
    const int out_off = UNI(d2.x), s_out = UNI(d2.y);
        {
            // z = mu + eps * exp(log_sigma)  (layer.py:279) and the KL row terms (_ae_utils.py:49)
            const float* o = lds + out_off;
            gfloat* zg = (gfloat*)x.Zb_g + x.blk * ((Z + 15) & ~15) * 16;    // blocked transposed [rup16(Z)][16]
            for (int i = tid; i < FTM * Z; i += FTHREADS) {
                const int r = i / Z, k = i - r * Z;
                float z = 0.0f;                                              // finite filler for the padded rows
                if (r < rows_here) {
                    const float mu = o[r * s_out + k], ls = o[r * s_out + Z + k];
                    const float e = es[r * s_z + k];                         // staged by the prologue
                    const float el = __expf(ls);
                    z = mu + e * el;
                    zs[r * s_z + k] = z;
                    kl += -0.5f * (1.0f + 2.0f * ls - mu * mu - el * el);
                }
                if constexpr (TO_GLOBAL == 1) zg[k * 16 + r] = z;
                if constexpr (TO_GLOBAL == 2) __hip_atomic_store((float*)(zg + k * 16 + r), z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lds_barrier();
        }
}

typedef const __attribute__((address_space(4))) i32x16 stage_rec_t;      // constant address space: uniform index -> s_load_dwordx16

// ---- the stage sequence of one mini-batch, two forms (TG: the stages' store mode, as exec_stage's TO_GLOBAL).
// SeqGeneric: a loop over the records; each stage runs through exec_stage_variant's switch.
// SeqVae1<TG, DB, HB, ZB, Z2B>: the seven stages of a VAE with ONE hidden layer and sigmoid activations whose padded widths are
//   16 DB (features), 16 HB (hidden), 16 ZB (latent), 16 Z2B (both heads), written out with each stage's structure-specialised
//   body -- no loop, no switch.  Measured on the bench net (64-128-(16,16)-128-64), same bodies: 15.2 us per mini-batch through
//   the loop + switch, 13.45 us written out (the generic body written out: no gain; it is the combination that pays).
//   The host picks the instantiation from the stage table itself (persist_seq_kernel), else the generic kernel.
template <int TG>
struct SeqGeneric {
    template <class Stamp>
    static __device__ __forceinline__ void run(int n_stages, stage_rec_t* stage_tab, StageCtx& sx, float& bce, float& kl, Stamp stamp) {
        i32x16 u = stage_tab[0];
#pragma unroll 1
        for (int st = 0; st < n_stages; ++st) {
            const StageDec cur = decode_stage_u(u, sx);
            const int out_off = u[8], s_out = u[9];
            const int kind = exec_stage_variant<TG>((u[6] >> 24) & 255, cur, sx, bce);
            // the next stage's record is requested in front of this stage's barrier, behind its last LDS access (a scalar
            // load in flight would turn the fragment reads' counted waits into waits for everything)
            __builtin_amdgcn_sched_barrier(0);
            u = stage_tab[st + 1];
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            if (kind == FK_FWD_HEADS) reparam_after_heads<TG>(make_int4(out_off, s_out, 0, 0), sx, kl);
            stamp(st);
        }
    }
};
template <int TG, int DB, int HB, int ZB, int Z2B>
struct SeqVae1 {
    template <int ST, int K_, int NB_, int NT_, class Stamp>
    static __device__ __forceinline__ void stage(stage_rec_t* stage_tab, StageCtx& sx, float& bce, float& kl, Stamp stamp) {
        const i32x16 u = stage_tab[ST];
        const StageDec d = decode_stage_u(u, sx);
        exec_stage<TG, K_, NB_, ACT_SIGMOID, NT_>(d, sx, bce);
        lds_barrier();
        if constexpr (K_ == FK_FWD_HEADS) reparam_after_heads<TG>(make_int4(u[8], u[9], 0, 0), sx, kl);
        stamp(ST);
    }
    template <class Stamp>
    static __device__ __forceinline__ void run(int, stage_rec_t* stage_tab, StageCtx& sx, float& bce, float& kl, Stamp stamp) {
        stage<0, FK_FWD_ACT, DB, HB>(stage_tab, sx, bce, kl, stamp);
        stage<1, FK_FWD_HEADS, HB, Z2B>(stage_tab, sx, bce, kl, stamp);
        stage<2, FK_FWD_ACT, ZB, HB>(stage_tab, sx, bce, kl, stamp);
        stage<3, FK_FWD_BCE, HB, DB>(stage_tab, sx, bce, kl, stamp);
        stage<4, FK_BWD_ACT, DB, HB>(stage_tab, sx, bce, kl, stamp);
        stage<5, FK_BWD_LATENT, HB, ZB>(stage_tab, sx, bce, kl, stamp);
        stage<6, FK_BWD_ACT, Z2B, HB>(stage_tab, sx, bce, kl, stamp);
    }
};


}  // namespace smrt
