// Persistent training kernel for nets that fit in LDS: ONE launch runs every mini-batch of an epoch
// (smrt/autoencode/base.py:202-217, the `for batch in batches` loop and everything inside its sess.run).
//
// Grid: G = ceil(batch_size / 16) COMPUTE workgroups + ceil(tiles / 4) OWNER workgroups, 8 waves each, one per CU, all
// resident for the whole launch.
//   compute workgroup b carries rows [16 b, 16 b + 16) of every mini-batch through all forward / backward stages with the
//            WEIGHTS RESIDENT IN LDS (the same stage body as the per-launch kernel, fused_stage.hpp); every layer output
//            and pre-activation gradient is also streamed, write-through, to the blocked-transposed global workspace;
//   owner wave: every 16x16 tile of every layer's [x | 1]^T . dPre has ONE owner wave, which keeps that tile's weights
//            and both optimizer slots IN REGISTERS for the whole launch.  Per mini-batch it multiplies the tile over all
//            rows straight from the workspace (1 KiB-contiguous fragments, fixed order: deterministic), adds L2, clips,
//            applies the optimizer and stores the new weights, in the LDS-image layout, write-through into the image twin;
//   L-phase  every compute workgroup copies the image twin back into its LDS and starts the next mini-batch.
// (A first version let the compute workgroups reduce 16-row partial tiles themselves; in-kernel stamps showed 4.6 us of
// tile work and 2.9 us of owner work per step on the compute workgroups' own critical path -- 22.6 us a step against the
// two-kernel step's 19.5.  Owners on CUs of their own take all of it off that path.)
// One monotonic flag per producer workgroup orders the phases (cdna guide, Guideline 16, the flag form, sharded by
// producer; MI355X_MICROARCH "Valid forms", first table row): every byte that crosses is stored `sc1` (write-through),
// each storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane stores the step
// number into the workgroup's flag with an agent-scope `sc1` store; the consumer's first wave polls EVERY producer's flag
// with `sc1` loads, the workgroup meets at a barrier, and EVERY load of crossed bytes is an `sc1` buffer load into
// registers.  No dispatch order, timing or XCD placement is assumed.  Every spin is bounded
// (2 s of wall clock): on a time-out the workgroup raises an error word that all others poll, and every wave leaves.
// Flags and the error word are zeroed by a hipMemsetAsync in front of every launch.
//
// Reference arithmetic: smrt/autoencode/layer.py:272-279,299-340,365-368; _ae_utils.py:15-49; base.py:127-149,198-222.
#pragma once
#include "fused_stage.hpp"

namespace smrt {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int AUX_SC1 = 16;             // buffer-instruction cache policy: sc1 = write-through store / L1-bypassing load
__device__ __forceinline__ f32x4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX_SC1);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, AUX_SC1);
}
__device__ __forceinline__ void store4_sc1(float* p, float v) {        // global_store_dword ... sc1
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0 waits until every one of the n flags has reached `target` (each flag is written by ONE producer workgroup and
// only grows).  Lane i polls flags i, i + 64, ...: one relaxed sc1 load per lane and round trip, a short sleep between.
// false: gave up (time-out here, or another workgroup's).  A single counter that every producer adds to cost more: the
// adds of one step serialise on their line in L2 (11-13 ns each, 28 owners) and the add is a round trip where the flag is
// a fire-and-forget store.  (Keeping four polls in flight to see a change a quarter round trip earlier was 3 % SLOWER in round 2; round 3 tried them STAGGERED a
// fraction of a round trip apart, oldest waited for with a counted vmcnt: 13.36 us per mini-batch with one poll, 13.64 with two, 13.95 with four.)
constexpr int PCNT_FG = 0, PCNT_FW = 64, PCNT_ERR = 192;      // word offsets in PersistArgs::cnt (256 words, zeroed per launch)
constexpr int PCNT_FW_MAX = 128;
__device__ __forceinline__ bool wait_flags(unsigned* f, int n, unsigned target, unsigned* err, int lane) {
    const unsigned long long t0 = wall_clock64();                    // 100 MHz
    for (unsigned spins = 0;; ++spins) {
        bool behind = false;
        for (int i = lane; i < n; i += 64) {
            const unsigned v = __hip_atomic_load(f + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            behind |= (int)(v - target) < 0;
        }
        if (__builtin_amdgcn_ballot_w64(behind) == 0ull) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63u) == 63u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (wall_clock64() - t0 > 200000000ull) {                // 2 s: a peer is not resident or has died
                if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

constexpr int OWNER_TILES = 4;          // tiles (= working waves) per owner workgroup

#ifdef SMRT_STAMPS
#define PST(i) do { if (a.dbg && wg == 0 && tid == 0 && step == 4) a.dbg[i] = wall_clock64(); } while (0)
#define OST(i) do { if (a.dbg && wg == G && tid == 0 && step == 4) a.dbg[i] = wall_clock64(); } while (0)
#else
#define PST(i) do { } while (0)
#define OST(i) do { } while (0)
#endif

// ---- owner role: one wave per 16x16 tile of one layer's [x | 1]^T . dPre, weights and slots in registers
__device__ __forceinline__ void owner_role(const PersistArgs& a, int* go_flag, float* red) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int G = a.G, wg = blockIdx.x;
    const int ow = wg - G;                                  // owner workgroup index
    // four tiles per owner workgroup, on waves 0..3 = one wave per SIMD: a tile's 4 x rows/16 MFMAs have the matrix pipe of
    // their SIMD to themselves (with eight tiles per workgroup the multiplication took twice as long); waves 4..7 only
    // keep the barriers company (they issue no load and no MFMA: `live` guards the whole step body)
    const int t = ow * OWNER_TILES + wave;
    const bool live = wave < OWNER_TILES && t < a.total_tiles;      // (idle waves read the last tile's record and touch nothing else)
    const DwTile T = a.tiles[live ? t : a.total_tiles - 1];
    unsigned* const flag_g = a.cnt + PCNT_FG;
    unsigned* const flag_w = a.cnt + PCNT_FW;
    unsigned* const err = a.cnt + PCNT_ERR;
    const int m0 = T.m0, n0 = T.n0, fi = T.fi, fo = T.fo;
    const int col = n0 + c, row0 = m0 + 4 * q;
    float w[4], s0[4], s1[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + r;
        ok[r] = live && row <= fi && col < fo;              // row == fi is the bias
        const long ix = T.off + (long)min(row, fi) * T.ld + min(col, fo - 1);
        w[r] = a.W[ix]; s0[r] = a.s0[ix]; s1[r] = a.s1[ix];
    }
    float b1pow = a.st->b1pow, b2pow = a.st->b2pow;
    const bool adam = a.opt == 2;
    // operand fragments: blocked transposed [row block][feature][16 rows]; this lane's four rows of one feature = 16 bytes
    const int NFA = (fi + 15) & ~15, NFB = (fo + 15) & ~15;
    const int mA = min(m0 + c, fi - 1), nB = min(n0 + c, fo - 1);
    const bool ones = (m0 + c) == fi;                       // virtual ones column -> bias gradient
    const unsigned capblk = (unsigned)a.cap_blocks;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(T.A, capblk * (unsigned)NFA * 64u);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(T.B, capblk * (unsigned)NFB * 64u);
    const __amdgpu_buffer_rsrc_t r_img = make_rsrc(a.Wimg, (unsigned)(a.img_total * 4));
    const unsigned offA = ((unsigned)mA * 16u + 4u * q) * 4u, offB = ((unsigned)nB * 16u + 4u * q) * 4u;
    const bool l2on = a.has_l2 && T.l2;
    // First-layer tiles when the epoch's X was transposed once in front of the launches (XT_all): their A operand is not a
    // hand-off at all -- an earlier kernel wrote it -- so its fragments are requested BEFORE the wait for the rows (plain loads),
    // and only the dPre fragments are left for the round trip behind the flag.
    const bool xall = live && a.XT_all != nullptr && T.A == a.XT;           // (wave-uniform)
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.XT_all, (unsigned)a.n_steps * (unsigned)G * (unsigned)NFA * 64u);

    for (int step = 0; step < a.n_steps; ++step) {
        const int rows = (int)min((long)a.batch, a.n_rows - (long)step * a.batch);
        const int n_active = (rows + FTM - 1) / FTM;
        const bool last = step == a.n_steps - 1;
        f32x4 ax[8];
        if (xall) {
            const unsigned sbase = (unsigned)step * (unsigned)G * (unsigned)NFA * 64u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned jb = (unsigned)min(j, n_active - 1);
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rX, sbase + jb * (unsigned)NFA * 64u + offA, 0, 0);
                ax[j] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
            }
        }
        OST(45);
        if (wave == 0) {
            const bool go = wait_flags(flag_g, G, (unsigned)(step + 1), err, lane);
            if (lane == 0) *go_flag = go ? 1 : 0;
        }
        __syncthreads();
        if (!*go_flag) return;
        OST(46);
        float l2acc = 0.f;
        if (live) {          // (wave-uniform) waves without a tile only keep the barriers company: no loads, no MFMAs
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            for (int j0 = 0; j0 < n_active; j0 += 8) {          // 8 row blocks = 16 loads per lane in ONE round trip
                f32x4 av[8], bv[8];
    #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned jb = (unsigned)min(j0 + j, n_active - 1);
                    if (!xall) av[j] = load16_sc1(rA, jb * (unsigned)NFA * 64u + offA);
                    else if (j0 == 0) av[j] = ax[j];
                    else {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                            rX, ((unsigned)step * (unsigned)G + jb) * (unsigned)NFA * 64u + offA, 0, 0);
                        av[j] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
                    }
                    bv[j] = load16_sc1(rB, jb * (unsigned)NFB * 64u + offB);
                }
                __builtin_amdgcn_sched_barrier(0);
                // row blocks in pairs, in load order: the first MFMAs start when the first loads have landed (the waits count
                // down vmcnt); block j feeds chain j & 3, as in the two-kernel step (same summation order per chain)
    #pragma unroll
                for (int j = 0; j < 8; j += 2) {
    #pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const bool kin = (j0 + j + u) < n_active;
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            av[j + u][e] = ones ? 1.0f : av[j + u][e];
                            bv[j + u][e] = kin ? bv[j + u][e] : 0.0f;
                        }
                    }
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if ((j & 3) == 0) { MFMA16(av[j][e], bv[j][e], acc0); MFMA16(av[j + 1][e], bv[j + 1][e], acc1); }
                        else              { MFMA16(av[j][e], bv[j][e], acc2); MFMA16(av[j + 1][e], bv[j + 1][e], acc3); }
                    }
                }
            }
            OST(47);
            float alpha = 0.f;
            if (adam) alpha = a.lr * sqrtf(1.0f - b2pow) / (1.0f - b1pow);
            float g[4], wn[4];
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                g[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
                const bool reg = ok[r] && l2on && (row0 + r) < fi;
                g[r] = reg ? g[r] + a.l2 * w[r] : g[r];
                l2acc += reg ? 0.5f * w[r] * w[r] : 0.f;
                if (a.clip) g[r] = tf_clip(g[r], a.gmin, a.gmax);
            }
            opt_apply4(a.opt, a.lr, alpha, w, g, s0, s1, wn);
    #pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = ok[r] ? wn[r] : w[r];
            if (adam) { b1pow *= 0.9f; b2pow *= 0.999f; }
            // New weights into the image twin, 16 bytes per store: this lane's four fan-in rows of one column are one k-quad
            // (rows past fan_in are the image's zero padding); the bias sits in a quad of its own whose other three words
            // are unused.
            if (live && col < fo) {
                if (row0 < fi)
                    store16_sc1(r_img, (unsigned)(T.img_w + ((((row0 >> 2) * T.nn) + col) << 2)) * 4u,
                                f32x4{w[0], row0 + 1 < fi ? w[1] : 0.f, row0 + 2 < fi ? w[2] : 0.f, row0 + 3 < fi ? w[3] : 0.f});
                if (row0 <= fi && fi <= row0 + 3) {
                    const int rb = fi - row0;
                    const float bval = rb == 0 ? w[0] : (rb == 1 ? w[1] : (rb == 2 ? w[2] : w[3]));
                    store16_sc1(r_img, (unsigned)(T.img_b + 4 * col) * 4u, f32x4{bval, 0.f, 0.f, 0.f});
                }
            }
        }
        if (last) {             // L2 term of the last mini-batch's cost (pre-update weights), one partial per owner workgroup
            const float tot = wave_sum(l2acc);
            if (lane == 0) red[wave] = tot;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        OST(48);
        if (tid == 0) {
            __hip_atomic_store(flag_w + ow, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < FWAVES; ++i) s += red[i];
                a.l2_p[ow] = s;
            }
        }
    }
    // launch epilogue: weights and slots back to the flat buffers (read by later launches / get_params)
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (ok[r]) {
            const long ix = T.off + (long)(row0 + r) * T.ld + col;
            a.W[ix] = w[r]; a.s0[ix] = s0[r]; a.s1[ix] = s1[r];
        }
}

template <class Seq>
__device__ __forceinline__ void persist_train_body(const PersistArgs& a) {
    extern __shared__ __align__(16) float lds[];
    __shared__ float red[FWAVES], red2[FWAVES];
    __shared__ int go_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int wg = blockIdx.x, G = a.G;
    // The error word is sticky across the launches of one smrt_vae_train_epochs call (api.hip: enqueue_persist_epoch): a
    // launch that follows a timed-out one leaves at once -- weights, slots and step scalars stay as that epoch found them.
    if (__hip_atomic_load(a.cnt + PCNT_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (wg >= G) { owner_role(a, &go_flag, red); return; }

    const int Z = a.Z;
    unsigned* const flag_g = a.cnt + PCNT_FG;      // activations / gradients published: flag b = steps done by compute workgroup b
    unsigned* const flag_w = a.cnt + PCNT_FW;      // new weights published:             flag o = steps done by owner workgroup o
    unsigned* const err = a.cnt + PCNT_ERR;
    const int n_owner = (int)gridDim.x - G;
    const __amdgpu_buffer_rsrc_t r_img = make_rsrc(a.Wimg, (unsigned)(a.img_total * 4));

    // ---- launch prologue: zeroed activation tiles
    stage_rec_t* const stage_tab = (stage_rec_t*)(unsigned long long)a.stage_u;
    {
        float4* l4 = reinterpret_cast<float4*>(lds);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = (a.img_total >> 2) + tid; i < a.lds_total / 4; i += FTHREADS) l4[i] = z4;
    }
    const uint32_t step0 = a.st->step;

    float* const xs = lds + a.lds_x;
    float* const zs = lds + a.lds_z;
    float* const es = lds + a.lds_e;
    const int xf4_per_row = (a.D + 3) >> 2;
    const bool x_fast = (FTM * xf4_per_row <= FTHREADS) && ((a.D & 3) == 0);
    const float lo = a.lo, hi = a.hi;
    const int s_z = a.s_z;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);       // X tile of the NEXT step (x_fast): requested a phase early

    auto rows_of = [&](int step) -> int { return (int)min((long)a.batch, a.n_rows - (long)step * a.batch); };
    auto request_x = [&](int step) {
        if (!x_fast || step >= a.n_steps) return;
        const int rh = min(FTM, rows_of(step) - wg * FTM);
        xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < rh * xf4_per_row) {
            const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
            xv = *reinterpret_cast<const float4*>(a.X + ((long)step * a.batch + wg * FTM + r) * a.ldx + c4);
        }
    };
    // epsilon of a step (Philox, indexed by step and row of the mini-batch) into its LDS tile
    auto draw_eps = [&](int step) {
        if (step >= a.n_steps) return;
        const int rh = min(FTM, rows_of(step) - wg * FTM);
        for (int i = tid; i < rh * Z; i += FTHREADS) {
            const int r = i / Z, k = i - r * Z;
            es[r * s_z + k] = philox_normal1(a.seed, STREAM_EPS, step0 + (uint32_t)step, (uint64_t)(wg * FTM) * Z + i);
        }
    };
    request_x(0);
    draw_eps(0);

    for (int step = 0; step < a.n_steps; ++step) {
        const int rows = rows_of(step);
        const int rows_here = min(FTM, rows - wg * FTM);        // <= 0: no rows of this mini-batch: only keeps in step
        const bool active = rows_here > 0;
        const bool last = step == a.n_steps - 1;
        const float inv_batch = 1.0f / (float)rows;

        PST(0);
        // ---- L-phase: the weights of this step into LDS.  Step 0 reads what the host (or the previous launch) left in
        // the image twin; later steps wait until every owner workgroup has published its tiles of the previous step.
        if (wave == 0) {
            const bool go = step == 0 || wait_flags(flag_w, n_owner, (unsigned)step, err, lane);
            if (lane == 0) go_flag = go ? 1 : 0;
        }
        __syncthreads();
        if (!go_flag) return;
        PST(1);
        {
            const int n4 = a.img_total >> 2;
            constexpr int LB = 12;                              // 16-byte pieces in flight per lane: one round trip for 96 KB
            for (int i0 = tid; i0 < n4; i0 += LB * FTHREADS) {
                f32x4 v[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int i = min(i0 + u * FTHREADS, n4 - 1);
                    v[u] = load16_sc1(r_img, (unsigned)i * 16u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int i = i0 + u * FTHREADS;
                    if (i < n4) *reinterpret_cast<f32x4*>(lds + 4 * i) = v[u];
                }
            }
        }
        PST(2);
        // the X tile (requested a phase ago); this step's epsilon is in its tile already (drawn while the owners worked)
        if (active) {
            const int r0 = wg * FTM;
            if (x_fast) {
                if (tid < FTM * xf4_per_row) {
                    const int r = tid / xf4_per_row, c4 = (tid - r * xf4_per_row) << 2;
                    *reinterpret_cast<float4*>(xs + r * a.s_x + c4) = xv;
                }
            } else {
                for (int i = tid; i < FTM * a.D; i += FTHREADS) {
                    const int r = i / a.D, cc = i - r * a.D;
                    xs[r * a.s_x + cc] = (r < rows_here) ? a.X[((long)step * a.batch + r0 + r) * a.ldx + cc] : 0.0f;
                }
            }
        }
        __syncthreads();
        PST(3);

        float bce = 0.f, kl = 0.f;
        if (active) {
            // blocked transposed copy of the X tile for the owners of the first layer's tiles (write-through, fire and forget)
            // -- unless the whole epoch's X was transposed once in front of the launches (XT_all)
            if (!a.XT_all) {
                float* xt = a.XT + (long)wg * ((a.D + 15) & ~15) * 16;
                for (int i = tid; i < FTM * a.D; i += FTHREADS) {
                    const int cc = i >> 4, r = i & 15;
                    store4_sc1(xt + i, xs[r * a.s_x + cc]);             // i == cc * 16 + r: fully coalesced
                }
            }
            StageCtx sx;
            sx.lds = lds; sx.wave = wave; sx.c = c; sx.q = q; sx.tid = tid; sx.rows_here = rows_here; sx.Z = Z; sx.s_z = s_z;
            sx.lo = lo; sx.hi = hi; sx.inv_batch = inv_batch; sx.es = es; sx.zs = zs; sx.Zb_g = a.Zb_g; sx.blk = wg;
#ifdef SMRT_STAMPS
            sx.sdbg = nullptr;
#endif
            Seq::run(a.n_stages, stage_tab, sx, bce, kl, [&](int st) { (void)st; PST(4 + st); });
        }
        PST(40);
        // the X tile is free now: request the next step's while the update is under way
        request_x(step + 1);

        if (last) {             // cost of the epoch's LAST batch (base.py:222): per-workgroup partial sums for finalize
            bce = wave_sum(bce); kl = wave_sum(kl);
            if (lane == 0) { red[wave] = bce; red2[wave] = kl; }
            lds_barrier();
            if (tid == 0) {
                float bsum = 0.f, ksum = 0.f;
#pragma unroll
                for (int i = 0; i < FWAVES; ++i) { bsum += red[i]; ksum += red2[i]; }
                a.bce_p[wg] = bsum; a.kl_p[wg] = ksum;
            }
        }

        // ---- publish this workgroup's rows: every storing wave drains, the workgroup meets, one lane signals
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PST(41);
        if (tid == 0) __hip_atomic_store(flag_g + wg, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        draw_eps(step + 1);          // the epsilon tile is free (the stages are done): next step's draw hides behind the owners
    }

    // ---- launch epilogue: workgroup 0 advances the step scalars (every workgroup read them in its prologue, before the
    // first counter add; this store comes after the last counter wait)
    if (wg == 0) {
        if (wave == 0) {
            const bool go = wait_flags(flag_w, n_owner, (unsigned)a.n_steps, err, lane);
            if (lane == 0) go_flag = go ? 1 : 0;
        }
        __syncthreads();
        if (go_flag && tid == 0) {
            float p1 = a.st->b1pow, p2 = a.st->b2pow;
            for (int i = 0; i < a.n_steps; ++i) { p1 *= 0.9f; p2 *= 0.999f; }
            a.st->step = step0 + (uint32_t)a.n_steps;
            a.st->b1pow = p1; a.st->b2pow = p2;
        }
    }
}


}  // namespace smrt
