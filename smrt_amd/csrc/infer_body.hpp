// encode / decode / generate of LDS-resident nets in ONE persistent launch (see fused_small.hip for the design notes): the
// kernel body, instantiated for any chain (fused_small.hip) and written out per two-layer chain shape (infer_seq.hip).
#pragma once
#include "fused_stage.hpp"

namespace smrt {

constexpr int IMAXT = 8;               // activations up to 128 wide
// ONE workgroup per CU (several smaller ones were slower: a persistent grid is not spread evenly over the CUs by the
// dispatcher) of 16 waves = 4 per SIMD at 128 VGPRs: a wave's chain draw -> layer -> activation -> layer -> store is serial,
// so what keeps the matrix pipe fed is the OTHER waves of the SIMD (2 per SIMD: C2's generate 0.086 ms, 3: 0.081, 4: 0.078).
// The 8-wave variant remains for chains whose wave-private I/O tiles (ragged rows only) would not fit in LDS 16 times.

// (omode: 0 = the outputs become the next layer's fragments; 1 = last layer into the wave's LDS tile; 2 = last layer
// straight to global memory, 16 bytes per lane and tile: every row gets 64 contiguous bytes per store instruction.)
// One layer for one wave, all NN output tiles at once: acc[t] starts at the bias, and for every 16-wide k-block the NN
// weight fragments are requested in one batch and multiplied into NN independent accumulators (consecutive MFMAs never
// depend on each other once NN >= 2).  The outputs overwrite `cur` in place -- they are the next layer's B fragments.
// KS > 0: the layer's k extent in 16-wide blocks is known at compile time (a chain written out per shape, see fused_infer_kernel):
// no per-block guard, and the fragments past it are dead registers.
template <int NN, int KS = 0>
__device__ __forceinline__ void infer_layer(const float* lds, f32x4 (&cur)[IMAXT], int nK, int N, int w_off, int nn,
                                            int b_off, int act, int c, int q, int omode, float* tile, int s_t, float* grow) {
    f32x4 acc[NN];
#pragma unroll
    for (int t = 0; t < NN; ++t) {
        const float* bb = lds + b_off + 4 * (16 * t + 4 * q);             // bias image: b[col] at 4 * col, rup16(fan_out) entries
        acc[t] = f32x4{bb[0], bb[4], bb[8], bb[12]};
    }
    const float* wp = lds + w_off + ((q * nn + c) << 2);
    const int bstep = nn << 4;
#pragma unroll
    for (int j = 0; j < (KS > 0 ? KS : IMAXT); ++j) {
        if (KS > 0 || j < nK) {                                            // uniform; the block below is straight-line
            // weight fragments in groups of up to four tiles (16 registers): a group's reads are requested before its first
            // MFMA, and four independent accumulators are enough to keep consecutive MFMAs from depending on each other
            constexpr int GT = NN > 4 ? 4 : NN;
#pragma unroll
            for (int t0 = 0; t0 < NN; t0 += GT) {
                f32x4 w[GT];
#pragma unroll
                for (int t = 0; t < GT; ++t)
                    if (t0 + t < NN) w[t] = *reinterpret_cast<const f32x4*>(wp + j * bstep + 64 * (t0 + t));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < GT; ++t)
                        if (t0 + t < NN) MFMA16(w[t][e], cur[j][e], acc[t0 + t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NN; ++t) {
        float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
        act_fwd4(act, v);                                                  // y = act(x.W + b), layer.py:368
        const int col0 = 16 * t + 4 * q;
        if (omode == 2) {                                                  // last layer, straight to this lane's row (N % 4 == 0)
            if (grow && col0 < N) *reinterpret_cast<f32x4*>(grow + col0) = f32x4{v[0], v[1], v[2], v[3]};
        } else if (omode == 1) {
            *reinterpret_cast<f32x4*>(tile + c * s_t + col0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            // pad columns must be finite zeros: they meet the zero rows of the next layer's image
            if (16 * t + 16 <= N) cur[t] = f32x4{v[0], v[1], v[2], v[3]};      // (uniform: only a ragged last tile is masked)
            else cur[t] = f32x4{col0 < N ? v[0] : 0.f, col0 + 1 < N ? v[1] : 0.f, col0 + 2 < N ? v[2] : 0.f, col0 + 3 < N ? v[3] : 0.f};
        }
    }
}

#ifdef SMRT_STAMPS
#define IST(i) do { if (a.dbg && blockIdx.x == 0 && tid == 0 && it == 2) { a.dbg[i] = wall_clock64(); a.dbg[32 + i] = clock64(); } } while (0)
#define IWG(i) do { if (a.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) a.dbg[(blockIdx.x == 0 ? 20 : 24) + i] = wall_clock64(); } while (0)
#else
#define IST(i) do { } while (0)
#define IWG(i) do { } while (0)
#endif

// NK0 / NN0 / NN1 > 0: a TWO-layer chain (decode and generate: latent -> hidden -> features; encode: features -> hidden -> heads)
// whose padded widths are 16 NK0 -> 16 NN0 -> 16 NN1, written out: no loop over stage records and no switch over tile counts
// per 16-row block (the same loop + switch cost the training kernels paid, fused_stage.hpp), k extents and tile counts compile-time.
// Measured on the bench net (round 3): decode of 1 M rows 0.276 -> 0.242 ms, generate 176 470 rows 0.0726 -> 0.0678 ms.
// 0 / 0 / 0: any chain, from the records.  Same arithmetic in the same order either way.
template <int ITHREADS, int NK0, int NN0, int NN1>
__device__ __forceinline__ void fused_infer_body(const InferArgs& a) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    IWG(0);
    {
        const float4* src = reinterpret_cast<const float4*>(a.img);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < a.img_n4; i += ITHREADS) dst[i] = src[i];
    }
    __syncthreads();
    IWG(1);
    const long nblk = (a.n + 15) >> 4;
    // 16-row blocks go round-robin over (wave index, workgroup): block b belongs to wave (b / gridDim.x) % waves of workgroup
    // b % gridDim.x, so the ragged last round (2 838 of 4 096 wave slots for C2's rows) leaves every SIMD of the chip with
    // 10 or 11 blocks instead of 12 on some CUs and 8 on others
    const long wstride = (long)gridDim.x * (ITHREADS / 64);
    constexpr bool CHAIN = NK0 > 0;
    constexpr int KIN = CHAIN ? NK0 : IMAXT;                 // input fragments that can be live
    const int K = a.K, nK0 = CHAIN ? NK0 : a.st[0].K16 >> 4, k4n = a.st[0].K16 >> 2, last = a.n_stages - 1, Nl = a.st[last].N;
    // (a chain's records are loop invariants: read once, not per 16-row block)
    const int c0N = UNI(a.st[0].N), c0w = UNI(a.st[0].w_off), c0nn = UNI(a.st[0].nn), c0b = UNI(a.st[0].b_off), c0act = UNI(a.st[0].act);
    const int c1N = UNI(a.st[last].N), c1w = UNI(a.st[last].w_off), c1nn = UNI(a.st[last].nn), c1b = UNI(a.st[last].b_off), c1act = UNI(a.st[last].act);
    const int s_t = a.tile_stride;
    float* tile = lds + 4 * a.img_n4 + wave * (16 * s_t);
    [[maybe_unused]] int it = -1;
#pragma unroll 1
    for (long blk = (long)wave * gridDim.x + blockIdx.x; blk < nblk; blk += wstride) {         // wave-uniform
        const long r0 = blk << 4;
        ++it;
        IST(0);
        f32x4 cur[IMAXT];
        if (!a.in) {
            // generate: lane (c, q) DRAWS its own fragments: k-block j, quad q of row c is Philox block 4 j + q of the row
            // (K = Z is a multiple of 4 and the stream offset is block-aligned: host-checked) -- no z buffer, no launch of
            // its own in front of this one, no LDS
            const int kq_n = K >> 2;
            const bool rowok = r0 + c < a.n;
            const uint64_t qb = (a.gen_elem0 + (uint64_t)(r0 + c) * K) >> 2;
#pragma unroll
            for (int j = 2; j < KIN; ++j) cur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < (KIN < 2 ? KIN : 2); ++j) {                            // K <= 32 (host-checked)
                cur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (CHAIN || j < nK0) {                                                // uniform
                    const int kq = 4 * j + q;
                    const uint64_t qi = qb + (uint64_t)kq;
                    const Philox4 p = philox4x32_10((uint32_t)qi, (uint32_t)(qi >> 32), 0u, STREAM_GENERATE,
                                                    (uint32_t)a.gen_seed, (uint32_t)(a.gen_seed >> 32));
                    float n0, n1, n2, n3;
                    box_muller(p.x, p.y, n0, n1);
                    box_muller(p.z, p.w, n2, n3);
                    if (kq < kq_n && rowok)
                        cur[j] = f32x4{shift_scale(a.gen_loc, a.gen_scale, n0), shift_scale(a.gen_loc, a.gen_scale, n1),
                                       shift_scale(a.gen_loc, a.gen_scale, n2), shift_scale(a.gen_loc, a.gen_scale, n3)};
                }
            }
        } else if (a.vec_in) {
            // whole quads straight from global memory into the fragments (K % 4 == 0, ldin % 4 == 0, aligned): every row
            // gets 64 contiguous bytes per load instruction.  Rows past the end shadow the last row (never stored).
            const float* src = a.in + min(r0 + c, a.n - 1) * a.ldin + 4 * q;
#pragma unroll
            for (int j = 0; j < KIN; ++j) {
                cur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((CHAIN || j < nK0) && 16 * j + 4 * q < K) cur[j] = *reinterpret_cast<const f32x4*>(src + 16 * j);
            }
        } else {
            // Ragged rows enter through a wave-private LDS tile so that global traffic is whole rows (16 bytes per lane,
            // consecutive lanes along a row).  Rows past the end shadow the last row (never stored), columns past K are zero.
            // (no integer division: a row takes 4 or 16 lanes, whichever the row length fills)
            const int sh = (k4n >= 16) ? 4 : 2, lpr = 1 << sh, rpi = 64 >> sh;         // lanes per row, rows per instruction
            const int lr = lane >> sh, lc = lane & (lpr - 1);
            for (int rr = 0; rr < 16; rr += rpi) {
                const int r = rr + lr;
                const float* src = a.in + min(r0 + r, a.n - 1) * a.ldin;
                for (int m = lc; m < k4n; m += lpr) {
                    const int c4 = m << 2;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (c4 < K) v = *reinterpret_cast<const f32x4*>(src + c4);             // c4 + 3 < rup4(K) <= ldin
                    v[1] = (c4 + 1 < K) ? v[1] : 0.f; v[2] = (c4 + 2 < K) ? v[2] : 0.f; v[3] = (c4 + 3 < K) ? v[3] : 0.f;
                    *reinterpret_cast<f32x4*>(tile + r * s_t + c4) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");             // written by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < KIN; ++j) {
                cur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (CHAIN || j < nK0) cur[j] = *reinterpret_cast<const f32x4*>(tile + c * s_t + 16 * j + 4 * q);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        IST(2);
        float* const grow = (a.vec_out && r0 + c < a.n) ? a.out0 + (r0 + c) * a.ld0 : nullptr;
        if constexpr (CHAIN) {
            infer_layer<NN0, NK0>(lds, cur, NK0, c0N, c0w, c0nn, c0b, c0act, c, q, 0, tile, s_t, grow);
            IST(3);
            infer_layer<NN1, NN0>(lds, cur, NN0, c1N, c1w, c1nn, c1b, c1act, c, q, a.vec_out ? 2 : 1, tile, s_t, grow);
            IST(4);
        } else
#pragma unroll 1
        for (int s = 0; s <= last; ++s) {
            const int nK = UNI(a.st[s].K16) >> 4, N = UNI(a.st[s].N), nN = UNI(a.st[s].N16) >> 4, w_off = UNI(a.st[s].w_off),
                      nn = UNI(a.st[s].nn), b_off = UNI(a.st[s].b_off), act = UNI(a.st[s].act);
            const int tt = (s == last) ? (a.vec_out ? 2 : 1) : 0;
            switch (nN) {
                case 1: infer_layer<1>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 2: infer_layer<2>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 3: infer_layer<3>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 4: infer_layer<4>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 5: infer_layer<5>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 6: infer_layer<6>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                case 7: infer_layer<7>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
                default: infer_layer<8>(lds, cur, nK, N, w_off, nn, b_off, act, c, q, tt, tile, s_t, grow); break;
            }
            IST(3 + s);
        }
        IST(10);
        if (!a.vec_out) {
            // encode (mu | log_sigma into two arrays) and ragged widths: the tile holds the last layer's [16][N] block
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int lr = lane >> 4, lc = lane & 15;
            for (int rr = 0; rr < 16; rr += 4) {
                const int r = rr + lr;
                if (r0 + r < a.n) {
                    for (int k = lc; k < Nl; k += 16) {
                        const float v = tile[r * s_t + k];
                        if (k < a.split) a.out0[(r0 + r) * a.ld0 + k] = v;
                        else a.out1[(r0 + r) * a.ld1 + (k - a.split)] = v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");             // the tile is rewritten by the next block of rows
            __builtin_amdgcn_wave_barrier();
        }
        IST(11);
    }
    IWG(2);
}


}  // namespace smrt
