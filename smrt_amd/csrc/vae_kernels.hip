// HIP kernels of the VAE train-and-generate hot path (gfx950 / MI355X).
//
// Reference call sites replaced (paths relative to the reference root):
//   dense layer  act(x . dropout(W) + b)                       smrt/autoencode/layer.py:365-368
//   reparameterisation z = mu + eps * exp(log_sigma)           smrt/autoencode/layer.py:272-279
//   reconstruction loss (clipped Bernoulli NLL) + its gradient smrt/autoencode/_ae_utils.py:15-32
//   KL divergence + its gradient                               smrt/autoencode/_ae_utils.py:35-49
//   L2 penalty, value clip, optimizer.apply_gradients          smrt/autoencode/base.py:127-149
#include "vae_kernels.hpp"

#include <cstdlib>
#include "device_utils.hpp"
#include "gemm_mfma_f32.hpp"

namespace smrt {

// =====================================================================================
// Dense-layer GEMM with fused epilogues
// =====================================================================================
template <int ALAY, int BLAY, int WM, int WN, int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(const GemmArgs a) {   // 2 waves per SIMD: <= 256 VGPR + AGPR
    constexpr int BM = 64 * WM, BN = 64 * WN;
    __shared__ __align__(16) float ldsA[2 * lds_floats<ALAY, BM>()];      // two tile images each (gemm_tile)
    __shared__ __align__(16) float ldsB[2 * lds_floats<BLAY, BN>()];
    __shared__ float red[4];

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so XCD x
    // takes the x-th contiguous chunk of the row-major tile list: the column tiles that re-read one row block
    // of A then share an L2 instead of pulling it over the fabric eight times.  Bijective for any grid size.
    int m0, n0;
    {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
        const int xcd = lin & 7, idx = lin >> 3, qq = nwg >> 3, rr = nwg & 7;
        const int t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        m0 = (t / (int)gridDim.x) * BM; n0 = (t % (int)gridDim.x) * BN;
    }
#ifdef SMRT_STAMPS
    const unsigned long long c_beg = clock64(), w_beg = wall_clock64();
#endif
    f32x16 acc[WM][WN];
    const int arr = gemm_tile<ALAY, BLAY, WM, WN, false>(a.A, a.lda, a.M, a.B, a.ldb, a.N, m0, n0, 0, a.K, ldsA, ldsB, acc);

#ifdef SMRT_STAMPS
    if (a.dbg && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 1) {     // main-loop clock of one workgroup
        a.dbg[0] = clock64() - c_beg; a.dbg[1] = wall_clock64() - w_beg;
    }
#endif
    if constexpr (EPI == EPI_BIAS_ACT) {
        // y = act(x.W + b)                                                     layer.py:368
        // The bias is fetched once per column (the stores to C may alias it as far as the compiler knows, so inside
        // the element loop it would be re-loaded 16 times) and interior tiles skip the bounds checks.
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (m0 + BM <= a.M && n0 + BN <= a.N) {             // (interior: arrangement 0)
            float bj[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bj[j] = a.bias[n0 + (wave & 1) * 32 * WN + j * 32 + (lane & 31)];
            for_each_c<WM, WN>(acc, m0, n0, 0, [&](int row, int col, float v) {
                const int j = ((col - n0) >> 5) % WN;
                a.C[(long)row * a.ldc + col] = act_fwd(a.act, v + bj[j]);
            });
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + blk_col<WM, WN>(arr, wave, i, j) + r;
                    const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                    const float b = a.bias[min(col, a.N - 1)];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rowb + (e & 3) + 8 * (e >> 2);
                        if (row < a.M && col < a.N) a.C[(long)row * a.ldc + col] = act_fwd(a.act, acc[i][j][e] + b);
                    }
                }
        }
    } else if constexpr (EPI == EPI_HEADS_SPLIT) {
        // fused (z_mean | z_log_sigma) heads written to two contiguous [n][Z] outputs   autoencoder.py:500
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + blk_col<WM, WN>(arr, wave, i, j) + r;
                const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                const float b = a.bias[min(col, a.N - 1)];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rowb + (e & 3) + 8 * (e >> 2);
                    if (row < a.M && col < a.N) {
                        const float y = act_fwd(a.act, acc[i][j][e] + b);
                        if (col < a.Z) a.C[(long)row * a.Z + col] = y;
                        else           a.out2[(long)row * a.Z + (col - a.Z)] = y;
                    }
                }
            }
    } else if constexpr (EPI == EPI_BCE) {
        // last decoder layer: p = sigmoid(.), clipped Bernoulli NLL summed into a block partial, and
        // dL/dpre emitted directly.                     layer.py:333-338, _ae_utils.py:31-32
        float lsum = 0.f;
        auto bce_elem = [&](float v, float x, float b, float& dpre) {
            const float p = fast_sigmoid(v + b);
            const float pl = tf_maximum(p, a.lo);
            const float pc = tf_minimum(pl, a.hi);
            const float om = 1.0f - pc;
            lsum -= x * __logf(pc) + (1.0f - x) * __logf(om);
            // TF clip_by_value gradient gate: Maximum passes where p >= lo, Minimum where max(p,lo) <= hi
            const bool pass = (p >= a.lo) && (pl <= a.hi);
            const float dp = pass ? -(x * __builtin_amdgcn_rcpf(pc) - (1.0f - x) * __builtin_amdgcn_rcpf(om)) * a.inv_batch : 0.0f;
            dpre = dp * p * (1.0f - p);
            return p;
        };
        if (m0 + BM <= a.M && n0 + BN <= a.N && !a.out2) {
            // interior tiles: bias once per column, the 16 targets of a sub-tile fetched in one batch (see ACTGRAD)
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + wc * 32 * WN + j * 32 + r;
                    const int rowb = m0 + wr * 32 * WM + i * 32 + 4 * h;
                    const float b = a.bias[col];
                    float x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) x[e] = a.aux[(long)(rowb + (e & 3) + 8 * (e >> 2)) * a.ldaux + col];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float d;
                        bce_elem(acc[i][j][e], x[e], b, d);
                        a.C[(long)(rowb + (e & 3) + 8 * (e >> 2)) * a.ldc + col] = d;
                    }
                }
        } else {
            // Edge tiles (and the variant that also stores p): the same batches from CLAMPED addresses, only the stores and
            // the loss sum are predicated.  A load inside the bounds check is waited for on the spot, and a lane with
            // valid columns then walks through its 32-64 elements one memory round trip at a time.
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + blk_col<WM, WN>(arr, wave, i, j) + r, colc = min(col, a.N - 1);
                    const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                    const float b = a.bias[colc];
                    float x[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        x[e] = a.aux[(long)min(rowb + (e & 3) + 8 * (e >> 2), a.M - 1) * a.ldaux + colc];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rowb + (e & 3) + 8 * (e >> 2);
                        if (row < a.M && col < a.N) {
                            float d;
                            const float p = bce_elem(acc[i][j][e], x[e], b, d);
                            a.C[(long)row * a.ldc + col] = d;
                            if (a.out2) a.out2[(long)row * a.ldout2 + col] = p;
                        }
                    }
                }
        }
        const float tot = block_sum_256(lsum, red);
        if (threadIdx.x == 0) a.partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    } else if constexpr (EPI == EPI_MSE) {
        // plain auto-encoder output layer: y = act(.), squared error summed into a block partial and
        // dL/dpre = 2 (y - x) / (B D) * act'(y) emitted directly (autoencoder.py:260); a.inv_batch = 1 / (B D)
        float lsum = 0.f;
        {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + blk_col<WM, WN>(arr, wave, i, j) + r, colc = min(col, a.N - 1);
                    const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                    const float b = a.bias[colc];
                    float x[16];                      // targets: one batch from clamped addresses (see BCE)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        x[e] = a.aux[(long)min(rowb + (e & 3) + 8 * (e >> 2), a.M - 1) * a.ldaux + colc];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rowb + (e & 3) + 8 * (e >> 2);
                        if (row < a.M && col < a.N) {
                            const float y = act_fwd(a.act, acc[i][j][e] + b);
                            const float d = y - x[e];
                            lsum += d * d;
                            a.C[(long)row * a.ldc + col] = act_bwd(a.act, y, 2.0f * d * a.inv_batch);
                        }
                    }
                }
        }
        const float tot = block_sum_256(lsum, red);
        if (threadIdx.x == 0) a.partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    } else if constexpr (EPI == EPI_ACTGRAD) {
        // dPre_prev = (dPre . W^T) * act'(y_prev)
        if (m0 + BM <= a.M && n0 + BN <= a.N) {
            // Interior tiles: the 16 y_prev values of a 32x32 sub-tile are fetched in one batch, then combined and
            // stored.  Element by element, every load would queue behind the previous element's store (C may alias
            // aux as far as the compiler knows): 64 dependent memory round trips per lane.
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + wc * 32 * WN + j * 32 + r;
                    const int rowb = m0 + wr * 32 * WM + i * 32 + 4 * h;
                    float y[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        y[e] = a.aux[(long)(rowb + (e & 3) + 8 * (e >> 2)) * a.ldaux + col];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        a.C[(long)(rowb + (e & 3) + 8 * (e >> 2)) * a.ldc + col] = act_bwd(a.act, y[e], acc[i][j][e]);
                }
        } else {
            // edge tiles: the same batches from clamped addresses, predicated stores (see BCE)
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int col = n0 + blk_col<WM, WN>(arr, wave, i, j) + r, colc = min(col, a.N - 1);
                    const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                    float y[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        y[e] = a.aux[(long)min(rowb + (e & 3) + 8 * (e >> 2), a.M - 1) * a.ldaux + colc];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rowb + (e & 3) + 8 * (e >> 2);
                        if (row < a.M && col < a.N) a.C[(long)row * a.ldc + col] = act_bwd(a.act, y[e], acc[i][j][e]);
                    }
                }
        }
    } else if constexpr (EPI == EPI_LATENT) {
        // v = dL/dz.  Through the reparameterisation and the KL term into both heads:
        //   dmu = dz + mu/B ; dls = dz*eps*exp(ls) + (exp(2 ls) - 1)/B     layer.py:279, _ae_utils.py:49
        // N = Z is a fraction of a tile: every tile is an edge tile.  mu, log sigma and epsilon of a 32x32 sub-tile come in
        // one batch from clamped addresses (see BCE); sub-tiles that lie wholly past N are skipped (wave-uniform).
        {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int colb = n0 + blk_col<WM, WN>(arr, wave, i, j);
                    if (colb >= a.N) continue;
                    const int col = colb + r, colc = min(col, a.N - 1);
                    const int rowb = m0 + blk_row<WM, WN>(arr, wave, i, j) + 4 * h;
                    float mu[16], ls[16], ee[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const long rowc = min(rowb + (e & 3) + 8 * (e >> 2), a.M - 1);
                        mu[e] = a.aux[rowc * a.ldaux + colc];
                        ls[e] = a.aux[rowc * a.ldaux + a.Z + colc];
                        ee[e] = a.aux2[rowc * a.ldaux2 + colc];
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rowb + (e & 3) + 8 * (e >> 2);
                        if (row < a.M && col < a.N) {
                            const float v = acc[i][j][e];
                            const float dmu = v + mu[e] * a.inv_batch;
                            const float dls = v * ee[e] * expf(ls[e]) + (expf(2.0f * ls[e]) - 1.0f) * a.inv_batch;
                            a.C[(long)row * a.ldc + col] = act_bwd(a.act, mu[e], dmu);
                            a.C[(long)row * a.ldc + a.Z + col] = act_bwd(a.act, ls[e], dls);
                        }
                    }
                }
        }
    }
#ifdef SMRT_STAMPS
    __syncthreads();
    if (a.dbg && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 1) {
        a.dbg[2] = clock64() - c_beg; a.dbg[3] = wall_clock64() - w_beg;
    }
#endif
}

// Tile shape of one dense-layer GEMM: 128x128 when that alone gives the chip 1.5+ workgroups per CU to interleave,
// 128x64 for the mid-sized ones (batch 4096 x 1024 features is 256 tiles of 128x128: one workgroup per CU and a
// matrix pipe that idles through every barrier; measured +3 % on N5 at batch 4096), 64x64 below that (for the
// 450- and 784-wide layers of N1 at batch 4096 the 64x64 tiles stay ahead of 128x64 by 2 %).
static inline int tile_sel(int M, int N) {
    const long b128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    return b128 >= 384 ? 0 : (b128 >= 192 ? 1 : 2);
}

int gemm_grid_blocks(int kind, const GemmArgs& a) {
    (void)kind;
    if (a.M <= SMALL_M_MAX) return small_gemm_blocks(a);
    const int sel = tile_sel(a.M, a.N);
    const int tm = sel == 2 ? 64 : 128, tn = sel == 0 ? 128 : 64;
    return ((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn);
}

template <int ALAY, int BLAY, int EPI>
static void launch_gemm_t(hipStream_t s, const GemmArgs& a) {
    const int sel = tile_sel(a.M, a.N);
    if (sel == 0) {
        dim3 grid((a.N + 127) / 128, (a.M + 127) / 128);
        SMRT_LAUNCH((gemm_kernel<ALAY, BLAY, 2, 2, EPI>), grid, dim3(GEMM_THREADS), 0, s, a);
    } else if (sel == 1) {
        dim3 grid((a.N + 63) / 64, (a.M + 127) / 128);
        SMRT_LAUNCH((gemm_kernel<ALAY, BLAY, 2, 1, EPI>), grid, dim3(GEMM_THREADS), 0, s, a);
    } else {
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
        SMRT_LAUNCH((gemm_kernel<ALAY, BLAY, 1, 1, EPI>), grid, dim3(GEMM_THREADS), 0, s, a);
    }
}

// Skinny layers of a large mini-batch: the latent heads (h -> 2 Z) and the first decoder layer's backward product
// (dz = dPre . W^T, h -> Z) are one pass over a [rows x h] matrix into <= 32 columns.  The LDS-tiled kernel walks them as
// 64-wide tiles that are mostly padding; the register-fed 16-row tiles (small_gemm.hip) take them straight from global
// memory, whatever M is.  Measured on N1 at 8 192 rows (rocprofv3, us per launch, LDS-tiled -> register-fed): heads forward
// 18.5 -> 12.8, dz + latent epilogue 19.2 -> 9.7.  NOT the K <= 32 products (first decoder layer forward 11.3 -> 12.8, heads
// backward 10.4 -> 24.5: their 450-wide outputs are 15 column tiles of two k-blocks each), and not beyond 32 768 rows, where
// the LDS-tiled kernel's longer-lived workgroups stream better (65 536 rows: 36 -> 49).  Not the output layer either (its
// loss partials are counted per tile of the kernel that runs it).
static bool skinny_gemm(int epi, const GemmArgs& a) {
    static const int mode = []() { const char* e = std::getenv("SMRT_HIP_SKINNY"); return e ? std::atoi(e) : 1; }();
    static const int max_m = []() { const char* e = std::getenv("SMRT_HIP_SKINNY_MAXM"); return e ? std::atoi(e) : 32768; }();
    if (!mode || epi == EPI_BCE || epi == EPI_MSE || a.M > max_m) return false;
    return a.N <= 32 || (mode == 2 && a.K <= 32);
}

void launch_gemm(hipStream_t s, int kind, int epi, const GemmArgs& a) {
    if (a.M <= SMALL_M_MAX || skinny_gemm(epi, a)) { launch_small_gemm(s, kind, epi, a); return; }
    if (kind == GEMM_NN) {
        if (epi == EPI_BIAS_ACT)         launch_gemm_t<INNER, OUTER, EPI_BIAS_ACT>(s, a);
        else if (epi == EPI_BCE)         launch_gemm_t<INNER, OUTER, EPI_BCE>(s, a);
        else if (epi == EPI_MSE)         launch_gemm_t<INNER, OUTER, EPI_MSE>(s, a);
        else                             launch_gemm_t<INNER, OUTER, EPI_HEADS_SPLIT>(s, a);
    } else {
        if (epi == EPI_ACTGRAD)          launch_gemm_t<INNER, INNER, EPI_ACTGRAD>(s, a);
        else                             launch_gemm_t<INNER, INNER, EPI_LATENT>(s, a);
    }
}

// =====================================================================================
// Weight + bias gradients of every layer in one launch:  [X | 1]^T . dPre   (split-K into slabs)
// =====================================================================================
template <int WM, int WN>
__global__ __launch_bounds__(GEMM_THREADS, 2) void dw_kernel(const DwArgs a) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    __shared__ __align__(16) float ldsA[2 * lds_floats<OUTER, BM>()];
    __shared__ __align__(16) float ldsB[2 * lds_floats<OUTER, BN>()];

    // Work unit -> (layer, tile, K range).  Edge tiles of 784 / 450 / 20 / 10-wide layers multiply 1-3 of their 16 blocks'
    // worth per wave (gemm_tile's arrangements), so they are cut into FEWER, longer K ranges than interior tiles: every
    // workgroup of the launch then runs about equally long, and the launch is as long as the chip-wide average, not as long
    // as an interior tile's range (one round of 512 resident workgroups lasts as long as its slowest member).
    // Units are ordered K-range-major (all tiles' range 0, then every tile that has a range 1, ...): neighbouring
    // workgroups then work on neighbouring tiles of the SAME rows and share their operand panels in L2.
    int y = 0;
#pragma unroll 1
    for (int i = 1; i < a.splits; ++i) if ((int)blockIdx.x >= a.ycum[i]) y = i;
    int v = (int)blockIdx.x - a.ycum[y];
    int li = 0, tr = 0, tc = 0, S = 1;
    {
        bool found = false;
#pragma unroll 1
        for (int l = 0; l < a.n_layers && !found; ++l) {
            const DwLayer& Q = a.L[l];
            const int tm1 = Q.tiles_m - 1, tn1 = Q.tiles_n - 1;
            const int n_c[4] = {tm1 * tn1, tm1, tn1, 1};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (found || Q.s_cls[c] <= y) continue;
                if (v < n_c[c]) {
                    found = true; li = l; S = Q.s_cls[c];
                    tr = c == 0 ? v / tn1 : (c == 1 ? v : tm1);
                    tc = c == 0 ? v - tr * tn1 : (c == 2 ? v : tn1);
                } else v -= n_c[c];
            }
        }
    }
    const DwLayer& L = a.L[li];
    const int m0 = tr * BM, n0 = tc * BN;
    const int kps = (((a.rows + S - 1) / S) + 31) & ~31;
    const int kbeg = min(a.rows, y * kps);
    const int kend = min(a.rows, kbeg + kps);

    f32x16 acc[WM][WN];
    // The bias gradient is the column sum of dPre: the workgroups of the first row of tiles add up the B values they
    // stage anyway (16 VALU adds per K-tile), instead of a virtual ones row that costs a whole extra row of MFMA tiles
    // whenever fan_in is a multiple of the tile height.
    const bool want_db = (m0 == 0);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int arr = gemm_tile<OUTER, OUTER, WM, WN, false>(L.A, L.lda, L.fi, L.B, L.ldb, L.fo, m0, n0, kbeg, kend, ldsA, ldsB,
                                                           acc, want_db ? &cs : nullptr);

    float* C = L.C + (long)y * a.slab_stride;
    for_each_c<WM, WN>(acc, m0, n0, arr, [&](int row, int col, float v) {
        if (row < L.fi && col < L.fo) C[(long)row * L.ldc + col] = v;
    });
    if (want_db) {
        // thread t staged columns 4*(t % F4).. of k rows t / F4 (+ multiples of RPP): reduce the RPP partials per column
        constexpr int F4 = BN / 4, RPP = GEMM_THREADS / F4;
        __syncthreads();                                   // every wave is done reading the tile images
        float* red = ldsB;                                 // [RPP][BN]
        *reinterpret_cast<float4*>(red + (threadIdx.x / F4) * BN + (threadIdx.x % F4) * 4) = cs;
        __syncthreads();
        if (threadIdx.x < BN) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < RPP; ++i) t += red[i * BN + threadIdx.x];
            const int col = n0 + threadIdx.x;
            if (col < L.fo) C[(long)L.fi * L.ldc + col] = t;        // row fan_in of the augmented matrix = bias
        }
    }
    // The optimizer sums a.splits slabs for every element: a tile cut into S < a.splits ranges zeroes its region of the
    // slabs it does not produce (range y takes slabs S + y, 2 S + y, ...).
    for (int z = S + y; z < a.splits; z += S) {
        float* Z = L.C + (long)z * a.slab_stride;
        const int cq = (threadIdx.x & (BN / 4 - 1)) * 4, r0 = threadIdx.x / (BN / 4);
        constexpr int RP = GEMM_THREADS / (BN / 4);
        const int rows_here = min(BM, L.fi - m0);
        for (int r = r0; r < rows_here; r += RP) {
            float* dst = Z + (long)(m0 + r) * L.ldc + n0 + cq;
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n0 + cq + c < L.fo) dst[c] = 0.f;
        }
        if (want_db && threadIdx.x < BN && n0 + (int)threadIdx.x < L.fo) Z[(long)L.fi * L.ldc + n0 + threadIdx.x] = 0.f;
    }
}

void launch_dw(hipStream_t s, const DwArgs& a, bool big) {
    dim3 grid(a.total_tiles);            // (work units: see dw_kernel)
    if (big) SMRT_LAUNCH((dw_kernel<2, 2>), grid, dim3(GEMM_THREADS), 0, s, a);
    else     SMRT_LAUNCH((dw_kernel<1, 1>), grid, dim3(GEMM_THREADS), 0, s, a);
}

// =====================================================================================
// Reparameterisation + KL (forward)
// =====================================================================================
__device__ __forceinline__ void reparam_body(const ReparamArgs& a) {
    __shared__ float red[4];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int Z = a.Z;
    float kl = 0.f;
    if (idx < (long)a.rows * Z) {
        const int row = (int)(idx / Z), k = (int)(idx % Z);
        const float mu = a.ML[(long)row * a.ldml + k];
        const float ls = a.ML[(long)row * a.ldml + Z + k];
        const float e = a.eps_in ? a.eps_in[idx]
                                 : philox_normal1(a.seed, STREAM_EPS, a.st->step, (uint64_t)(a.row0_global + row) * Z + k);
        a.Zb[(long)row * a.ldz + k] = mu + e * expf(ls);                     // layer.py:279
        a.Eb[(long)row * a.ldz + k] = e;
        kl = -0.5f * (1.0f + 2.0f * ls - mu * mu - expf(2.0f * ls));       // _ae_utils.py:49
    }
    const float tot = block_sum_256(kl, red);
    if (threadIdx.x == 0) a.kl_partials[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void reparam_kernel(const ReparamArgs a) { reparam_body(a); }
__global__ __launch_bounds__(256) void reparam_multi(const MultiSlot<ReparamArgs>* __restrict__ t) {
    const MultiSlot<ReparamArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx) return;
    reparam_body(s.a);
}
static MultiReg<ReparamArgs> reg_reparam(reparam_kernel, reparam_multi);

int reparam_blocks(int rows, int Z) { return (int)(((long)rows * Z + 255) / 256); }

void launch_reparam(hipStream_t s, const float* ML, long ldml, const float* eps_in, float* Zb, float* Eb, long ldz,
                    int rows, int Z, uint64_t seed, const DevState* st, long row0_global, float* kl_partials) {
    ReparamArgs a{ML, ldml, eps_in, Zb, Eb, ldz, rows, Z, seed, st, row0_global, kl_partials};
    SMRT_LAUNCH(reparam_kernel, dim3(reparam_blocks(rows, Z)), dim3(256), 0, s, a);
}

// One thread per Philox block = FOUR consecutive normals of the stream (elements 4 q .. 4 q + 3): ten rounds and two
// Box-Muller pairs yield four outputs (drawing one element per thread computed the whole block and threw three quarters
// of it away: 23.6 us of C2's 100 us generate call).  The stream is unchanged: element e still comes from block e >> 2.
__global__ __launch_bounds__(256) void sample_normal_kernel(float* out, long ld, long rows, int cols, uint64_t seed,
                                                            uint32_t stream, uint32_t step, uint64_t elem0,
                                                            float loc, float scale) {
    const uint64_t q = (elem0 >> 2) + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const long n = rows * cols;
    const long i0 = (long)(4 * q) - (long)elem0;                    // position of the block's first element in `out`
    if (i0 >= n) return;
    const Philox4 p = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    box_muller(p.x, p.y, v[0], v[1]);
    box_muller(p.z, p.w, v[2], v[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = shift_scale(loc, scale, v[e]);
    if (i0 >= 0 && i0 + 3 < n && (cols & 3) == 0 && (i0 & 3) == 0 && (ld & 3) == 0) {
        // whole block inside one row, 16-byte aligned (out is 16-byte aligned, ld a multiple of 4)
        const long row = i0 / cols; const int k = (int)(i0 - row * cols);
        *reinterpret_cast<float4*>(out + row * ld + k) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long i = i0 + e;
            if (i >= 0 && i < n) { const long row = i / cols; out[row * ld + (int)(i - row * cols)] = v[e]; }
        }
    }
}

void launch_sample_normal(hipStream_t s, float* out, long ld, long rows, int cols, uint64_t seed, uint32_t stream,
                          uint32_t step, uint64_t elem0, float loc, float scale) {
    const long n = rows * cols;
    if (n <= 0) return;
    const uint64_t nq = ((elem0 + (uint64_t)n - 1) >> 2) - (elem0 >> 2) + 1;          // Philox blocks the range touches
    SMRT_LAUNCH(sample_normal_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, out, ld, rows, cols,
                       seed, stream, step, elem0, loc, scale);
}

// =====================================================================================
// Dropout on the weights: W' = W / keep * floor(keep + U)                     layer.py:365
// =====================================================================================
__device__ __forceinline__ int find_seg(const OptSeg* seg, int n_seg, long i) {
    int s = 0;
#pragma unroll 1
    for (int j = 1; j < n_seg; ++j) if (i >= seg[j].start) s = j;
    return s;
}

__global__ __launch_bounds__(256) void dropout_kernel(const DropArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const OptSeg& sg = a.seg[find_seg(a.seg, a.n_seg, i)];
    const long loc = i - sg.start;
    const int c = (int)(loc % sg.ld);
    const long rrow = loc / sg.ld;
    const float w = a.W[i];
    if (loc >= sg.w_count || c >= sg.fo) {          // bias row / pad column: never dropped
        a.Wd[i] = w;
        if (!a.injected) a.mask[i] = 1.0f;
        return;
    }
    float m;
    if (a.injected) {
        m = a.mask[i];
    } else {
        long e;
        if (sg.is_heads) e = (c < sg.Z) ? sg.abi_w_off + rrow * sg.Z + c
                                        : sg.abi_w_off + (long)sg.hfi * sg.Z + rrow * sg.Z + (c - sg.Z);
        else             e = sg.abi_w_off + rrow * sg.fo + c;
        m = floorf(a.keep + philox_uniform1(a.seed, STREAM_DROPOUT, a.st->step, (uint64_t)e));
        a.mask[i] = m;
    }
    a.Wd[i] = (w / a.keep) * m;
}

void launch_dropout(hipStream_t s, const DropArgs& a) {
    SMRT_LAUNCH(dropout_kernel, dim3((unsigned)((a.P + 255) / 256)), dim3(256), 0, s, a);
}

// =====================================================================================
// Fused slab-reduce + L2 + dropout-backward + value clip + optimizer update     base.py:127-149
// =====================================================================================
__device__ __forceinline__ float opt_update(const OptArgs& a, float w, float g, float& s0, float& s1, float alpha) {
    return opt_update_ieee(a.opt, a.lr, alpha, w, g, s0, s1);
}

__device__ __forceinline__ void optimizer_body(const OptArgs& a) {
    __shared__ float red[4];
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    float l2acc = 0.f;
    const float b1pow = a.st->b1pow, b2pow = a.st->b2pow;      // (requested with everything else; block 0 advances them below)
    __builtin_amdgcn_sched_barrier(0);
    if (a.advance && blockIdx.x == 0 && threadIdx.x == 0) {
        // every earlier kernel of the step has read the step scalars (stream order) and no block of this one does
        a.st->step += 1;
        a.st->b1pow *= 0.9f;
        a.st->b2pow *= 0.999f;
    }
    if (i0 < a.P) {
        // every operand is requested up front, unconditionally (both slot buffers exist under every optimizer; without
        // dropout the weight buffer stands in for the post-dropout weights and the mask): a load under a branch is waited
        // for at the join, one memory round trip per operand
        const float* wdp = a.has_drop ? a.Wd : a.W;
        const float* mkp = a.has_drop ? a.mask : a.W;
        float4 g = *reinterpret_cast<const float4*>(a.G + i0);
        const float4 w = *reinterpret_cast<const float4*>(a.W + i0);
        const float4 wd4 = *reinterpret_cast<const float4*>(wdp + i0);
        const float4 mk4 = *reinterpret_cast<const float4*>(mkp + i0);
        const float4 s0 = *reinterpret_cast<const float4*>(a.s0 + i0);
        const float4 s1 = *reinterpret_cast<const float4*>(a.s1 + i0);
        // the segment (layer) this quad belongs to: the records are kernel arguments, read with scalar loads for a uniform j
        // and selected per lane (indexing the argument struct with a per-lane index turns into dependent vector loads)
        long seg_start = a.seg[0].start, seg_wcount = a.seg[0].w_count;
        int seg_l2 = a.seg[0].l2;
#pragma unroll
        for (int j = 1; j < MAX_LAYERS; ++j) {
            const bool in = j < a.n_seg && i0 >= a.seg[j].start;
            seg_start = in ? a.seg[j].start : seg_start;
            seg_wcount = in ? a.seg[j].w_count : seg_wcount;
            seg_l2 = in ? a.seg[j].l2 : seg_l2;
        }
        const bool is_w = (i0 - seg_start) < seg_wcount;
        // the other slabs, four requests at a time (summed in slab order: the same bits as one by one)
        int s = 1;
        for (; s + 4 <= a.splits; s += 4) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(a.G + (long)(s + u) * a.slab_stride + i0);
#pragma unroll
            for (int u = 0; u < 4; ++u) { g.x += t[u].x; g.y += t[u].y; g.z += t[u].z; g.w += t[u].w; }
        }
        for (; s < a.splits; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(a.G + (long)s * a.slab_stride + i0);
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
        }
        float gv[4] = {g.x, g.y, g.z, g.w};
        float wv[4] = {w.x, w.y, w.z, w.w};
        if (is_w) {
            const float wd[4] = {wd4.x, wd4.y, wd4.z, wd4.w};
            const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (a.has_l2 && seg_l2) {
                    gv[j] += a.l2 * wd[j];                      // tf.nn.l2_loss on the post-dropout tensor
                    l2acc += 0.5f * wd[j] * wd[j];
                }
                if (a.has_drop) gv[j] = (gv[j] / a.keep) * mk[j];   // back through dropout to the variable
            }
        }
        if (a.grad_out) *reinterpret_cast<float4*>(a.grad_out + i0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        float alpha = 0.f;
        if (a.opt == 2) alpha = a.lr * sqrtf(1.0f - b2pow) / (1.0f - b1pow);
        const bool has_s0 = a.opt != 0 && a.opt != 5;
        const bool has_s1 = a.opt == 1 || a.opt == 2 || a.opt == 4;
        float s0v[4] = {s0.x, s0.y, s0.z, s0.w}, s1v[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = gv[j];
            if (a.clip) gj = tf_clip(gj, a.gmin, a.gmax);      // tf.clip_by_value, base.py:144
            wv[j] = opt_update(a, wv[j], gj, s0v[j], s1v[j], alpha);
        }
        *reinterpret_cast<float4*>(a.W + i0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
        if (has_s0) *reinterpret_cast<float4*>(a.s0 + i0) = make_float4(s0v[0], s0v[1], s0v[2], s0v[3]);
        if (has_s1) *reinterpret_cast<float4*>(a.s1 + i0) = make_float4(s1v[0], s1v[1], s1v[2], s1v[3]);
    }
    const float tot = block_sum_256(l2acc, red);
    if (threadIdx.x == 0) a.l2_partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void optimizer_kernel(const OptArgs a) { optimizer_body(a); }
__global__ __launch_bounds__(256) void optimizer_multi(const MultiSlot<OptArgs>* __restrict__ t) {
    const MultiSlot<OptArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx) return;
    optimizer_body(s.a);
}
static MultiReg<OptArgs> reg_optimizer(optimizer_kernel, optimizer_multi);

int optimizer_blocks(long P) { return (int)((P / 4 + 255) / 256); }

void launch_optimizer(hipStream_t s, const OptArgs& a, int* n_blocks_out) {
    const int nb = optimizer_blocks(a.P);
    if (n_blocks_out) *n_blocks_out = nb;
    SMRT_LAUNCH(optimizer_kernel, dim3(nb), dim3(256), 0, s, a);
}

// Sum the split-K slabs into one contiguous gradient (data-parallel path hands this to the all-reduce).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ G, int splits, long slab_stride,
                                                           long P, float* __restrict__ out) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= P) return;
    float4 g = *reinterpret_cast<const float4*>(G + i0);
    for (int s = 1; s < splits; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(G + (long)s * slab_stride + i0);
        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
    *reinterpret_cast<float4*>(out + i0) = g;
}

void launch_reduce_slabs(hipStream_t s, const float* G, int splits, long slab_stride, long P, float* out) {
    SMRT_LAUNCH(reduce_slabs_kernel, dim3(optimizer_blocks(P)), dim3(256), 0, s, G, splits, slab_stride, P, out);
}

// =====================================================================================
// Cost = mean(rec + kl) + l2 * sum 1/2 |W'|^2 ; advance the step-dependent scalars   autoencoder.py:479-480
// =====================================================================================
__device__ __forceinline__ void finalize_body(const FinalizeArgs& a) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < a.n_bce; i += 256) s += a.bce[i];
    const float bce = block_sum_256(s, red);
    s = 0.f;
    for (int i = threadIdx.x; i < a.n_kl; i += 256) s += a.kl[i];
    const float kl = block_sum_256(s, red);
    s = 0.f;
    for (int i = threadIdx.x; i < a.n_l2; i += 256) s += a.l2p[i];
    const float l2s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float data = a.presummed ? a.presummed[0] : (bce + kl) * a.inv_batch;
        if (a.loss_sum_out) a.loss_sum_out[0] = data;
        if (a.cost_out) a.cost_out[0] = data + a.l2 * l2s;
        if (a.ring_ctr) {                                // the epoch ends here: cost and time of epoch e, straight to the host
            const unsigned e = a.ring_ctr[0];
            a.ring_cost[e] = data + a.l2 * l2s;
            a.ring_time[e + 1] = wall_clock64();
            if (a.ring_err) a.ring_err[e] = a.err_src[0];
            a.ring_ctr[0] = e + 1;
        }
        if (a.advance) {
            a.st->step += 1;
            a.st->b1pow *= 0.9f;
            a.st->b2pow *= 0.999f;
        }
    }
    if (a.zero192 && threadIdx.x < 192) a.zero192[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(256) void finalize_kernel(const FinalizeArgs a) { finalize_body(a); }
__global__ __launch_bounds__(256) void finalize_multi(const MultiSlot<FinalizeArgs>* __restrict__ t) {
    const MultiSlot<FinalizeArgs>& s = t[blockIdx.z];
    if (blockIdx.x >= s.gx) return;
    finalize_body(s.a);
}
static MultiReg<FinalizeArgs> reg_finalize(finalize_kernel, finalize_multi);

__global__ void ring_start_kernel(unsigned long long* ring_time, unsigned* ring_ctr, unsigned e0) {
    ring_time[e0] = wall_clock64();
    ring_ctr[0] = e0;
}
void launch_ring_start(hipStream_t s, unsigned long long* ring_time, unsigned* ring_ctr, unsigned e0) {
    SMRT_LAUNCH(ring_start_kernel, dim3(1), dim3(1), 0, s, ring_time, ring_ctr, e0);
}

void launch_finalize(hipStream_t s, const FinalizeArgs& a) {
    SMRT_LAUNCH(finalize_kernel, dim3(1), dim3(256), 0, s, a);
}

}  // namespace smrt
