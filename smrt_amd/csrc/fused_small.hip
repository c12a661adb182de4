// Two-kernel training step for nets that fit in LDS (tabular-sized VAEs, e.g. 64-128-(16,16)-128-64:
// 96 KB of weights + 43 KB of activations per workgroup).
//
//   fused_fwd_bwd_kernel : one workgroup (8 waves) owns 16 consecutive rows of the mini-batch and carries
//       them through EVERY layer, forward, loss and backward-through-activations.  All weights are staged
//       into LDS by ONE batch of wide loads; after that nothing on the critical path touches global
//       memory: activations stay in LDS between layers, both MFMA operands are LDS reads, stage
//       barriers wait for LDS only.  Layer outputs and pre-activation gradients are streamed to global
//       memory fire-and-forget for the second kernel.
//   fused_dw_opt_kernel  : one wave per 16x16 tile of [X | 1]^T . dPre (all layers in one grid); the
//       epilogue adds L2, undoes dropout, clips and applies the optimizer in place.  The step counter / Adam powers
//       are advanced by workgroup 0 at once, or -- under Adam, which reads them here -- by the last workgroup to
//       finish (ticket counter, no spinning).
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
#include "fused_body.hpp"
#include "infer_body.hpp"

#include <algorithm>
#include <cstdlib>

namespace smrt {

__global__ __launch_bounds__(FTHREADS) void fused_fwd_bwd_kernel(const FusedArgs a) { fused_fwd_bwd_body<SeqGeneric<1>>(a); }

// persist_seq.hip: the instantiations written out per net shape (nullptr when there is none)
fused_kernel_t fused_seq_kernel_h2(int db, int zb, int z2b);
fused_kernel_t fused_seq_kernel_h4(int db, int zb, int z2b);
fused_kernel_t fused_seq_kernel_h8(int db, int zb, int z2b);

fused_kernel_t fused_pick_kernel(const StageU* st, int n_stages) {
    int db, hb, zb, z2b;
    if (!seq_vae1_shape(st, n_stages, db, hb, zb, z2b)) return fused_fwd_bwd_kernel;
#ifdef SMRT_SEQ_ONLY8              // (diagnostic and A/B builds carry the 128-wide instantiations only)
    fused_kernel_t k = hb == 8 ? fused_seq_kernel_h8(db, zb, z2b) : nullptr;
#else
    fused_kernel_t k = hb == 2 ? fused_seq_kernel_h2(db, zb, z2b) : hb == 4 ? fused_seq_kernel_h4(db, zb, z2b)
                     : hb == 8 ? fused_seq_kernel_h8(db, zb, z2b) : nullptr;
#endif
    return k ? k : fused_fwd_bwd_kernel;
}

void launch_fused_fwd_bwd(hipStream_t s, fused_kernel_t k, const FusedArgs& a) {
    const int grid = (a.rows + FTM - 1) / FTM;
    SMRT_LAUNCH(k, dim3(grid), dim3(FTHREADS), (size_t)a.lds_total * sizeof(float), s, a);
}

int fused_set_max_lds(fused_kernel_t k, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// =====================================================================================
// Weight + bias gradients of every layer with the optimizer fused into the epilogue.
// One wave per 16x16 tile of  [X | 1]^T . dPre ; K = rows of the mini-batch (no split).
// =====================================================================================
__global__ __launch_bounds__(BTHREADS) void fused_dw_opt_kernel(const DwOptArgs a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
#ifdef SMRT_STAMPS
#define BST(i) do { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[i] = wall_clock64(); } while (0)
#else
#define BST(i) do { } while (0)
#endif
    BST(20);
    const int t = blockIdx.x * 4 + wave;
    const bool live = t < a.total_tiles;
    // ONE 64-byte scalar load from a device-resident table describes this wave's tile (idle waves shadow the
    // last tile and store nothing).  Walking the per-layer kernel arguments instead costs a dependent memory
    // round trip per layer.
    const DwTile T = a.tiles[live ? t : a.total_tiles - 1];
    const int m0 = T.m0, n0 = T.n0, fi = T.fi, fo = T.fo, rows = a.rows;

    // ---- batch 1 of loads: the optimizer operands of this lane's four output elements.  Every load is UNCONDITIONAL and
    // from a clamped, always-valid address (both slot buffers exist under every optimizer; without dropout the weight
    // buffer stands in for the post-dropout weights and the mask): a load under a branch or a select-on-condition makes
    // the compiler wait for it at the join -- four dependent memory round trips here before the GEMM loads were requested.
    const bool has_s0 = a.opt != 0 && a.opt != 5;
    const bool has_s1 = a.opt == 1 || a.opt == 2 || a.opt == 4;
    const float* wdp = a.has_drop ? a.Wd : a.W;
    const float* mkp = a.has_drop ? a.mask : a.W;
    const int col = n0 + c;
    long idx[4]; bool ok[4];
    float w[4], s0[4], s1[4], wd[4], mk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * q + r;
        ok[r] = live && row <= fi && col < fo;               // row == fi is the bias
        idx[r] = T.off + (long)min(row, fi) * T.ld + min(col, fo - 1);
        w[r] = a.W[idx[r]];
        s0[r] = a.s0[idx[r]];
        s1[r] = a.s1[idx[r]];
        wd[r] = wdp[idx[r]];
        mk[r] = mkp[idx[r]];
    }
    // Step-dependent scalars.  Only Adam reads them here (its bias-correction powers), so only Adam has to wait for
    // every workgroup before advancing them (ticket below: a device-scope atomic round trip at the tail of the launch).
    // Any other optimizer lets workgroup 0 advance them right away, off the critical path.
    const bool adam = a.opt == 2;
    const float b1pow = adam ? a.st->b1pow : 0.f, b2pow = adam ? a.st->b2pow : 0.f;
    if (!adam && blockIdx.x == 0 && tid == 0) {
        a.st->step += 1;
        a.st->b1pow *= 0.9f;
        a.st->b2pow *= 0.999f;
    }

    BST(21);
    // ---- [X | 1]^T . dPre over all rows.  Both operands are stored blocked transposed ([row block][feature][16 rows])
    // by the forward/backward kernel: a lane's four k values are one 16-byte load and a wave's load is 1 KiB
    // contiguous; rows past the batch hold dPre = 0 (and finite activations), so no k masking is needed up to
    // rup16(rows).
    const int m = m0 + c, mm = min(m, fi - 1);                // A operand: output row m (input feature); m == fi -> ones
    const int n = n0 + c, nn = min(n, fo - 1);                // B operand: output col n
    const int NFA = (fi + 15) & ~15, NFB = (fo + 15) & ~15;   // feature counts of the blocked buffers
    const float* Ap = T.A + mm * 16 + 4 * q;                  // row block j of feature f: ((j * NF + f) * 16) floats
    const float* Bp = T.B + nn * 16 + 4 * q;
    const bool ones = (m == fi);
    const int rows16 = (rows + 15) & ~15;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int k0 = 0; k0 < rows16; k0 += 128) {               // 128 rows (8 k-blocks, 16 loads per lane) per batch
        f32x4 av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long jb = min(k0 + 16 * j, rows16 - 16);    // clamped + unconditional; == 16 * row block
            av[j] = *reinterpret_cast<const f32x4*>(Ap + jb * NFA);
            bv[j] = *reinterpret_cast<const f32x4*>(Bp + jb * NFB);
        }
        // all sixteen loads are requested before the first select below touches one of them (left alone, hipcc requests
        // one operand's eight, waits for them, and only then requests the other's: two round trips instead of one)
        __builtin_amdgcn_sched_barrier(0);
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
    // ---- epilogue: L2, dropout backward, value clip, optimizer (base.py:127-149).  Straight-line over the four elements
    // (uniform flags select, one switch on the optimizer around all four); only the stores are predicated.
    float alpha = 0.f;
    if (a.opt == 2) alpha = a.lr * sqrtf(1.0f - b2pow) / (1.0f - b1pow);
    float l2acc = 0.f;
    float g[4], wn[4];
    const bool l2on = a.has_l2 && T.l2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        g[r] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
        const bool is_w = (m0 + 4 * q + r) < fi;
        const bool reg = ok[r] && is_w && l2on;
        g[r] = reg ? g[r] + a.l2 * wd[r] : g[r];
        l2acc += reg ? 0.5f * wd[r] * wd[r] : 0.f;
        if (a.has_drop) g[r] = is_w ? (g[r] / a.keep) * mk[r] : g[r];
    }
    if (a.grad_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ok[r]) a.grad_out[idx[r]] = g[r];
    }
    if (a.clip) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = tf_clip(g[r], a.gmin, a.gmax);
    }
    opt_apply4(a.opt, a.lr, alpha, w, g, s0, s1, wn);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (ok[r]) {
            a.W[idx[r]] = wn[r];
            if (has_s0) a.s0[idx[r]] = s0[r];
            if (has_s1) a.s1[idx[r]] = s1[r];
        } else {
            wn[r] = w[r];
        }
    }
    // the LDS-image twin read by the next step's forward/backward kernel: this lane's four fan-in rows of one
    // column are one k-quad of the interleaved image, i.e. one 16-byte store
    if (a.Wimg && live && col < fo) {
        const int row0 = m0 + 4 * q;
        if (row0 + 3 < fi) {
            *reinterpret_cast<float4*>(a.Wimg + T.img_w + ((((row0 >> 2) * T.nn) + col) << 2)) =
                make_float4(wn[0], wn[1], wn[2], wn[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                if (row < fi) a.Wimg[T.img_w + ((((row >> 2) * T.nn) + col) << 2) + (row & 3)] = wn[r];
                else if (row == fi) a.Wimg[T.img_b + 4 * col] = wn[r];
            }
        }
    }
    BST(23);
    const float tot = block_sum_256(l2acc, red);            // __syncthreads inside: every lane's st reads are done
    if (tid == 0) {
        a.l2_partials[blockIdx.x] = tot;
        if (adam) {
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
    BST(24);
}

// =====================================================================================
// The same weight + bias gradient tiles for LARGER mini-batches of a net that fits the fused forward/backward kernel:
// the rows are cut into `splits` ranges of whole 16-row blocks (blockIdx.y), a wave sums its 16x16 tile of
// [X | 1]^T . dPre over one range and writes it into slab blockIdx.y of the gradient buffer, in the flat parameter
// layout optimizer_kernel reduces (deterministic: one writer per element and slab, fixed summation order).
// =====================================================================================
__global__ __launch_bounds__(BTHREADS) void fused_dw_slab_kernel(const DwSlabArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int t = blockIdx.x * 4 + wave;
    const bool live = t < a.total_tiles;
    const DwTile T = a.tiles[live ? t : a.total_tiles - 1];
    const int m0 = T.m0, n0 = T.n0, fi = T.fi, fo = T.fo;
    const int m = m0 + c, mm = min(m, fi - 1);
    const int n = n0 + c, nn = min(n, fo - 1);
    const int NFA = (fi + 15) & ~15, NFB = (fo + 15) & ~15;
    const float* Ap = T.A + mm * 16 + 4 * q;
    const float* Bp = T.B + nn * 16 + 4 * q;
    const bool ones = (m == fi);
    const int nblk = (a.rows + 15) >> 4;                         // 16-row blocks in all
    const int b0 = blockIdx.y * a.blocks_per_split, b1 = min(nblk, b0 + a.blocks_per_split);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int kb = b0; kb < b1; kb += 8) {                        // 8 row blocks (16 loads per lane) per batch
        f32x4 av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long jb = 16L * min(kb + j, b1 - 1);           // clamped + unconditional
            av[j] = *reinterpret_cast<const f32x4*>(Ap + jb * NFA);
            bv[j] = *reinterpret_cast<const f32x4*>(Bp + jb * NFB);
        }
        __builtin_amdgcn_sched_barrier(0);                       // all sixteen requested before the first select
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool kin = (kb + j) < b1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[j][e] = ones ? 1.0f : av[j][e];
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
    float* G = a.G + (long)blockIdx.y * a.slab_stride;
    const int col = n0 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * q + r;                          // row == fi is the bias
        if (live && row <= fi && col < fo)
            G[T.off + (long)row * T.ld + col] = (acc0[r] + acc1[r]) + (acc2[r] + acc3[r]);
    }
}

void launch_fused_dw_slab(hipStream_t s, const DwSlabArgs& a, int splits) {
    SMRT_LAUNCH(fused_dw_slab_kernel, dim3(dw_opt_blocks(a.total_tiles), splits), dim3(BTHREADS), 0, s, a);
}

// flat parameter buffer -> LDS-image twin (after set_params, after a per-layer-path update, or from the
// post-dropout weights every step)
__global__ __launch_bounds__(256) void build_image_kernel(const float* __restrict__ Wflat, const int* __restrict__ tab,
                                                          float* __restrict__ img, int n_f4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_f4) return;
    const float4 v = *reinterpret_cast<const float4*>(Wflat + 4L * i);
    float* d = img + tab[i];                    // consecutive fan-out columns are 4 floats apart in the image
    d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
}

void launch_build_image(hipStream_t s, const float* Wflat, const int* tab, float* img, int n_f4) {
    SMRT_LAUNCH(build_image_kernel, dim3((n_f4 + 255) / 256), dim3(256), 0, s, Wflat, tab, img, n_f4);
}

// =====================================================================================
// Inference through a chain of dense layers (encode: X -> [mu | log_sigma], autoencoder.py:488-500; decode / generate:
// z -> decoder, autoencoder.py:502-517,549-572) for nets on the fused path.  The layer-by-layer GEMM chain writes every
// intermediate to HBM and reads it back, one launch per layer and 32 768-row chunk; here ONE persistent launch does it:
//   * a workgroup stages the image of the layers it needs (a contiguous range of the LDS-image twin) once;
//   * then every WAVE owns 16 rows at a time and carries them through all layers on its own, with the activations in
//     REGISTERS: the product is computed transposed, Y^T = W^T . X^T (weights are the MFMA A operand, the 16 rows the B
//     operand), so lane (c, q) ends up with Y[row c][16 t + 4 q + r] in register r of output tile t -- which is exactly
//     the B fragment of k-block t of the next layer (same k permutation inside a 16-block as everywhere in this file).
//     No transpose, no LDS traffic for activations, no barrier after the staging; the only LDS reads are the weight
//     fragments (one ds_read_b128 per four MFMAs).
// HBM traffic is the algorithmic 4 (K + N) bytes per row.  Layers up to 128 wide (IMAXT 16-column tiles); wider nets on
// the fused path keep the per-layer chain.
// =====================================================================================
template <int ITHREADS>
__global__ __launch_bounds__(ITHREADS) void fused_infer_kernel(const InferArgs a) { fused_infer_body<ITHREADS, 0, 0, 0>(a); }

// infer_seq.hip, one object per hidden width: the 16-wave kernel written out for the chain 16 nk0 -> 16 HB -> 16 nn1 (nullptr: none)
infer_kernel_t infer_seq_kernel_h2(int nk0, int nn1);
infer_kernel_t infer_seq_kernel_h4(int nk0, int nn1);
infer_kernel_t infer_seq_kernel_h8(int nk0, int nn1);

static infer_kernel_t infer_pick_chain(const InferArgs& a) {
    const char* e = std::getenv("SMRT_HIP_INFER_SEQ");                      // (read per call: tests compare the two forms)
    if ((e && std::atoi(e) == 0) || a.n_stages != 2) return nullptr;
    if ((a.st[0].K16 & 15) || (a.st[0].N16 & 15) || (a.st[1].N16 & 15) || a.st[1].K16 != a.st[0].N16) return nullptr;
    const int nk0 = a.st[0].K16 >> 4, hb = a.st[0].N16 >> 4, nn1 = a.st[1].N16 >> 4;
#ifdef SMRT_SEQ_ONLY8
    return hb == 8 ? infer_seq_kernel_h8(nk0, nn1) : nullptr;
#else
    return hb == 2 ? infer_seq_kernel_h2(nk0, nn1) : hb == 4 ? infer_seq_kernel_h4(nk0, nn1) : hb == 8 ? infer_seq_kernel_h8(nk0, nn1) : nullptr;
#endif
}

static int launch_fused_infer_k(hipStream_t s, infer_kernel_t k, int ithreads, const InferArgs& a, int lds_bytes) {
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    const long nblk = (a.n + 15) >> 4;
    const long wgs = (nblk + ithreads / 64 - 1) / (ithreads / 64);
    SMRT_LAUNCH(k, dim3((unsigned)std::min<long>(wgs, 256)), dim3(ithreads), (size_t)lds_bytes, s, a);
    return 0;
}

int launch_fused_infer(hipStream_t s, const InferArgs& a) {
    if (a.maxt > IMAXT) return -1;                                         // the caller takes the per-layer path
    // the wave-private tiles are only touched by ragged input rows / split or ragged output rows
    const bool tiles = !((!a.in || a.vec_in) && a.vec_out);
    const int tile_bytes = tiles ? 16 * a.tile_stride * (int)sizeof(float) : 0;
    const int lds16 = a.img_n4 * 16 + 16 * tile_bytes, lds8 = a.img_n4 * 16 + 8 * tile_bytes;
    if (lds16 <= 160 * 1024) {
        infer_kernel_t k = infer_pick_chain(a);
        return launch_fused_infer_k(s, k ? k : fused_infer_kernel<1024>, 1024, a, lds16);
    }
    if (lds8 <= 160 * 1024) return launch_fused_infer_k(s, fused_infer_kernel<512>, 512, a, lds8);
    return -1;
}

int dw_opt_blocks(int total_tiles) { return (total_tiles + 3) / 4; }

void launch_fused_dw_opt(hipStream_t s, const DwOptArgs& a) {
    SMRT_LAUNCH(fused_dw_opt_kernel, dim3(dw_opt_blocks(a.total_tiles)), dim3(BTHREADS), 0, s, a);
}

}  // namespace smrt
