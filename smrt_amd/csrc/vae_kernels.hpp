// Kernel argument blocks and launcher declarations (host <-> vae_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace smrt {

constexpr int MAX_LAYERS = 16;      // internal dense layers: 2*L + 2 <= 14

// Per-kernel timing (smrt_vae_profile_epochs): when the host thread has set an event pair, the next launch goes
// through hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end times (what
// rocprofv3 --kernel-trace reports), not with the time a separately recorded event happens to retire.
extern thread_local hipEvent_t g_prof_start, g_prof_stop;

// Epoch graphs are BUILT, not captured: while the host thread has a graph open (smrt_vae_train_epochs), every launch
// becomes a kernel node chained behind the previous one.  Stream capture is avoided on purpose: while any stream of
// the process is capturing, the runtime refuses device-wide synchronises and legacy-stream copies from every other
// host thread, and smrt_balance trains several classes from several threads (PyTorch uploads on the legacy stream).
struct GraphBuild {
    hipGraph_t graph = nullptr;
    hipGraphNode_t prev = nullptr;
    hipError_t err = hipSuccess;
    long nodes = 0;
};
extern thread_local GraphBuild g_build;

template <class... KArgs, size_t... I>
inline void smrt_arg_ptrs(std::tuple<KArgs...>& vals, void** ptrs, std::index_sequence<I...>) {
    ((ptrs[I] = (void*)&std::get<I>(vals)), ...);
}
template <class... KArgs, class... Args>
inline void smrt_add_node(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args&&... args) {
    std::tuple<KArgs...> vals(std::forward<Args>(args)...);          // the kernel's exact parameter types
    void* ptrs[sizeof...(KArgs)];
    smrt_arg_ptrs(vals, ptrs, std::index_sequence_for<KArgs...>{});
    hipKernelNodeParams kp{};
    kp.func = (void*)kernel;
    kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = (unsigned)shmem;
    kp.kernelParams = ptrs; kp.extra = nullptr;
    hipGraphNode_t node = nullptr;
    const hipError_t e = hipGraphAddKernelNode(&node, g_build.graph, g_build.prev ? &g_build.prev : nullptr,
                                               g_build.prev ? 1 : 0, &kp);
    if (e == hipSuccess) { g_build.prev = node; ++g_build.nodes; }
    else if (g_build.err == hipSuccess) g_build.err = e;
}

// ---- grouped launches (api.hip: smrt_vae_train_epochs_multi).  Several independent handles that train the same kind of
// net side by side -- the per-class VAEs of one smrt_balance call on one GPU -- go through the SAME kernel sequence, each
// with grids far too small for the chip (4 to 120 workgroups).  While a host thread RECORDS, SMRT_LAUNCH stores the launch
// instead of issuing it; the recorded sequences of all handles are then merged position by position, and launches of one
// kernel become ONE launch of its grouped twin: gridDim.z = handle, the argument structs in a device table.
template <class A> struct alignas(16) MultiSlot { A a; unsigned gx, gy, pad0, pad1; };   // + the handle's own grid extent

struct RecordedLaunch {
    const void* kernel = nullptr;           // the plain kernel
    dim3 grid, block; unsigned shmem = 0;
    std::vector<char> args;                 // bytes of its single argument struct (empty: has no grouped twin)
    std::function<void(hipStream_t)> plain; // issues the launch as it was recorded
};
struct Recorder { std::vector<RecordedLaunch>* out = nullptr; int group = 1; };   // group: handles recorded side by side
extern thread_local Recorder g_record;

struct MultiInfo { const void* multi; size_t arg_size, slot_size, gx_off; };
void smrt_register_multi(const void* kernel, const MultiInfo& info);
const MultiInfo* smrt_multi_of(const void* kernel);
template <class A> struct MultiReg {
    MultiReg(void (*kernel)(A), void (*multi)(const MultiSlot<typename std::remove_const<A>::type>*)) {
        using T = typename std::remove_const<A>::type;
        smrt_register_multi((const void*)kernel, MultiInfo{(const void*)multi, sizeof(T), sizeof(MultiSlot<T>), offsetof(MultiSlot<T>, gx)});
    }
};

// a launch outside recording: graph node while a graph is being built, plain launch otherwise
template <class... KArgs>
inline void smrt_issue(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream,
                       std::tuple<KArgs...>& vals) {
    void* ptrs[sizeof...(KArgs)];
    smrt_arg_ptrs(vals, ptrs, std::index_sequence_for<KArgs...>{});
    if (g_build.graph) {
        hipKernelNodeParams kp{};
        kp.func = (void*)kernel; kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = (unsigned)shmem;
        kp.kernelParams = ptrs; kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        const hipError_t e = hipGraphAddKernelNode(&node, g_build.graph, g_build.prev ? &g_build.prev : nullptr,
                                                   g_build.prev ? 1 : 0, &kp);
        if (e == hipSuccess) { g_build.prev = node; ++g_build.nodes; }
        else if (g_build.err == hipSuccess) g_build.err = e;
    } else {
        (void)hipLaunchKernel((const void*)kernel, grid, block, ptrs, shmem, stream);
    }
}
template <class... KArgs, class... Args>
inline void smrt_record(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args&&... args) {
    RecordedLaunch r;
    r.kernel = (const void*)kernel; r.grid = grid; r.block = block; r.shmem = (unsigned)shmem;
    std::tuple<typename std::remove_const<KArgs>::type...> vals(std::forward<Args>(args)...);
    if constexpr (sizeof...(KArgs) == 1) {
        using A = typename std::tuple_element<0, decltype(vals)>::type;
        if (std::is_trivially_copyable<A>::value && smrt_multi_of((const void*)kernel)) {
            r.args.resize(sizeof(A));
            std::memcpy(r.args.data(), &std::get<0>(vals), sizeof(A));
        }
    }
    r.plain = [kernel, grid, block, shmem, vals](hipStream_t s) mutable {
        std::tuple<KArgs...> v2(vals);
        smrt_issue(kernel, grid, block, shmem, s, v2);
    };
    g_record.out->push_back(std::move(r));
}

#define SMRT_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                   \
    do {                                                                                                      \
        if (smrt::g_record.out) smrt::smrt_record(kernel, grid, block, shmem, stream, __VA_ARGS__);           \
        else if (smrt::g_build.graph) smrt::smrt_add_node(kernel, grid, block, shmem, __VA_ARGS__);           \
        else if (g_prof_start) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, g_prof_start, g_prof_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                             \
    } while (0)

// Device-resident scalars that change every step (read by kernels so a captured graph replays).
struct DevState {
    float b1pow, b2pow;      // Adam beta power accumulators (TF keeps them as fp32 variables)
    uint32_t step;           // number of optimizer steps applied so far
    uint32_t pad;
};

enum Epi : int { EPI_BIAS_ACT = 0, EPI_BCE = 1, EPI_HEADS_SPLIT = 2, EPI_ACTGRAD = 3, EPI_LATENT = 4, EPI_MSE = 5 };
enum GemmKind : int { GEMM_NN = 0, GEMM_NT = 1 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    const float* bias;
    const float* aux;  long ldaux;      // BCE: target X; ACTGRAD: producing layer's output; LATENT: fused heads output
    const float* aux2; long ldaux2;     // LATENT: epsilon
    float* out2;       long ldout2;     // BCE: optional p store; HEADS_SPLIT: log-sigma output
    float* partials;                    // BCE: one partial loss sum per block
    int act, Z;
    float lo, hi, inv_batch;
    unsigned long long* dbg;            // diagnostic builds only (-DSMRT_STAMPS)
    int xmap, pad_;                     // small-batch kernels: workgroup -> tile map (0 as dispatched, 1 / 2 XCD chunks, small_gemm.hip)
};

struct DwLayer {
    const float* A;      // layer input  [rows][lda]   (K = rows, M = fan_in (+1 virtual ones row -> bias grad))
    const float* B;      // dPre         [rows][ldb]
    float* C;            // gradient segment [fan_in + 1][ldc] inside slab 0
    int fi, fo;
    long lda, ldb, ldc;
    int tiles_n, tile_begin;
    // tiled (large-batch) kernel only: work units = (tile, K range).  A tile's class -- interior / last column of tiles /
    // last row of tiles / corner -- decides into how many K ranges it is cut (see plan_dw_units in api.hip)
    int tiles_m;
    int s_cls[4];
};
struct DwArgs {
    DwLayer L[MAX_LAYERS];
    int n_layers, total_tiles;
    int rows, k_per_split, splits;
    long slab_stride;
    int ycum[33];        // tiled kernel: work units in front of K-range index y (units are ordered range-major)
};

// Fused weight-gradient + update launch of small mini-batches (small_gemm.hip: small_dw_opt): one record
// per output tile, built once per workspace (api.hip: ensure_train), read by its workgroup with ONE 64-byte scalar load.
struct DwRec {
    const float* A;      // layer input [rows][lda]; NULL: the mini-batch itself (DwFusedArgs::X, ldx), which moves every step
    const float* B;      // dPre [rows][ldb]
    long off;            // start of the layer's augmented matrix [fan_in + 1][ldc] in the flat parameter / slot buffers
    int fi, fo;
    int lda, ldb, ldc;
    int m0, n0;          // tile origin (fan-in row, fan-out column)
    int l2, pad_;        // the layer is regularised (the heads are not, layer.py:96)
};
static_assert(sizeof(DwRec) == 64, "DwRec is read as one 64-byte record");
struct DwFusedArgs {
    const DwRec* tiles; int total_tiles, rows;
    const float* X; long ldx;
    float* W; float* s0; float* s1;
    float* grad_out;     // optional: pre-clip gradient wrt the variables (internal layout)
    float* l2_partials;  // one per tile
    DevState* st;
    int opt, clip, has_l2, advance;
    int want_l2, pad_;   // the step's cost is read: leave the L2 partial sums (one per tile) for finalize
    float lr, l2, gmin, gmax;
    unsigned long long* dbg;   // diagnostic builds only (-DSMRT_STAMPS)
};

struct OptSeg {
    long start;          // first float of the segment in the internal flat buffer
    long w_count;        // fan_in * ld (the bias row follows)
    long total;          // (fan_in + 1) * ld
    int ld, fo, l2, is_heads;
    long abi_w_off;      // first element of this layer's W in the W-only ABI enumeration (dropout stream)
    int Z, hfi;
};
struct OptArgs {
    OptSeg seg[MAX_LAYERS];
    int n_seg;
    long P;
    float* W; const float* Wd; const float* mask;
    const float* G; int splits; long slab_stride;
    float* s0; float* s1;
    float* grad_out;                 // optional: pre-clip gradient wrt the variables (internal layout)
    float* l2_partials;
    DevState* st;
    int opt, clip, has_drop, has_l2;
    float lr, l2, gmin, gmax, keep;
    int advance;                     // block 0 bumps step / beta powers (never under Adam, whose blocks all read them here)
};

struct DropArgs {
    OptSeg seg[MAX_LAYERS];
    int n_seg;
    long P;
    const float* W; float* Wd; float* mask;
    const DevState* st;
    uint64_t seed;
    float keep;
    int injected;                    // masks already in `mask`
};

struct ReparamArgs {
    const float* ML; long ldml;          // fused heads output [rows][ldml]: mu | log_sigma
    const float* eps_in;                 // injected epsilon [rows][Z] contiguous, or null: the Philox stream
    float* Zb; float* Eb; long ldz;      // z and the epsilon used, [rows][ldz]
    int rows, Z;
    uint64_t seed; const DevState* st; long row0_global;
    float* kl_partials;                  // one partial KL sum per block
};

// The middle of the VAE at small mini-batches (small_gemm.hip): heads -> reparameterisation -> first decoder layer in one
// launch forward, first decoder layer -> latent -> heads in one launch backward (2 Z <= 32).
struct MidArgs {
    const float* H; long ldh;                        // output of the last encoder layer [M][ldh]
    const float* Wh; long ldwh; const float* bh;     // fused heads [hL][ldwh] (mu | log_sigma columns), bias [2Z]
    const float* Wd; long ldwd; const float* bd;     // first decoder layer [Z][ldwd], bias [hL]
    float* ML; long ldml;                            // heads output [M][ldml]
    float* Zb; float* Eb; long ldz;                  // z and the epsilon used [M][ldz]
    float* OUT; long ldo;                            // forward: first decoder layer's output [M][ldo]
    const float* eps_in;                             // injected epsilon [M][Z] contiguous, or null: the Philox stream
    float* kl_partials;                              // one partial KL sum per 32-row block
    const float* DP; long lddp;                      // backward: dPre of the first decoder layer [M][lddp]
    float* DML; long lddml;                          // dPre of the heads [M][lddml]
    float* DOUT; long lddo;                          // dPre of the last encoder layer [M][lddo]
    int M, hL, Z, act;
    float inv_batch;
    uint64_t seed; const DevState* st; long row0_global;
    int xmap, pad_;                                  // workgroup -> tile map (small_gemm.hip: xcd_tile)
    unsigned long long* dbg;                         // diagnostic builds only (-DSMRT_STAMPS)
};
void launch_mid_fwd(hipStream_t s, const MidArgs& a);
void launch_mid_bwd(hipStream_t s, const MidArgs& a);
int  mid_row_blocks(int rows, int hL);
int  small_tile_shape(int M, int N);                  // 22 | 12 | 11: (16 MT) x (16 NT) tile of a small-batch launch

struct FinalizeArgs {
    const float* bce; int n_bce;
    const float* kl;  int n_kl;
    const float* l2p; int n_l2;
    const float* presummed;          // DP: already reduced, already scaled (bce+kl)/B_global
    float inv_batch, l2;
    float* cost_out;
    float* loss_sum_out;             // optional: (bce+kl)*inv_batch without the L2 term
    DevState* st;
    int advance;                     // bump step / beta powers
    // Epoch ring (api.hip: train_epochs without the persistent kernel), or null: the launch that closes an epoch also
    // writes its cost and a 100 MHz time stamp into pinned host memory at the epoch's index, kept in device memory --
    // an epoch graph then needs no copy node and no event pair around it.
    float* ring_cost; unsigned long long* ring_time; unsigned* ring_ctr;
    // ... after a persistent launch also the launch's error word (persist_body.hpp) into ring_err at the same index, and
    // the 192 producer flags are cleared for the next launch (stream order: the persistent kernel has ended)
    const unsigned* err_src; unsigned* ring_err; unsigned* zero192;
};
void launch_ring_start(hipStream_t s, unsigned long long* ring_time, unsigned* ring_ctr, unsigned e0);   // time[e0] = now, index = e0

// Row-tile fused forward + backward (fused_small.hip)
struct FusedLayer {       // host-side LDS plan of one dense layer
    int fi, fo, ld, act;
    int lds_out, s_out;   // LDS offset / row stride of the layer output tile (floats)
    int lds_w, nn;        // LDS offset of the k-quad interleaved weight image [rup16(fi)/4][nn][4], nn = rup16(fo)+1
    int lds_b;            // LDS offset of the bias vector, stride 4: b[col] at lds_b + 4*col
    int w_off;            // offset of this layer's augmented matrix inside the flat parameter buffer (floats)
};
// One GEMM stage of the kernel (forward through a layer, or backward through it): ONE code path runs every
// stage from this 80-byte record, so the instruction footprint of the kernel is one stage, not 2L+3 of them.
enum FusedKind : int { FK_FWD_ACT = 0, FK_FWD_HEADS = 1, FK_FWD_BCE = 2, FK_BWD_ACT = 3, FK_BWD_LATENT = 4 };
// Stage bodies whose STRUCTURE is fixed at compile time (fused_stage.hpp: run_stage_variant): the stage kind, its k extent
// as 1 / 2 / 4 / 8 16-wide k-blocks and whether the activation is the sigmoid.  0 = the generic body.
constexpr int STAGE_ACT_SIGMOID = 3;           // == ACT_SIGMOID (device_utils.hpp) == SMRT_ACT_SIGMOID (smrt_hip.h)
constexpr int stage_variant(int kind, int k16, int act) {
    const int nb = k16 >> 4;
    const int ix = (k16 & 15) ? -1 : (nb == 1 ? 0 : (nb == 2 ? 1 : (nb == 4 ? 2 : (nb == 8 ? 3 : -1))));
    return ix < 0 ? 0 : 1 + ((kind * 4 + ix) << 1) + (act == STAGE_ACT_SIGMOID ? 1 : 0);
}
struct FusedStage {
    int in_off, sin, K16, w_off;      // A operand tile (LDS offset, row stride), padded k extent, weight image offset
    int nn, N, kind, act;             // image column count, valid output columns, FusedKind, activation id
    int out_off, s_out, b_off, aux_off;   // LDS tile written by the epilogue; bias image; auxiliary tile (x / producing layer's output)
    int s_aux, NF, trans, variant;    // aux row stride; feature count of the global buffer; 1: sum over fan-out (backward); stage_variant()
    float* gout; long pad2;           // global [row block][NF][16] buffer the epilogue streams to
};
static_assert(sizeof(FusedStage) == 80, "FusedStage is read as five 16-byte LDS words");
// The same stage, decoded on the host for the persistent kernel: 16 words in device memory that ONE scalar load brings into
// SGPRs (the kernel runs the same stages thousands of times per launch; decoding the 80-byte record per stage and step was a
// third of the instructions its compute waves issued).
struct StageU {
    int in_off, sin, nb_all, w_off;       // A tile (LDS offset, row stride), 16-wide k-blocks, weight image offset
    int nn, N, packed, N16;               // image column count, valid / padded output columns, kind | act << 8 | trans << 16 | variant << 24
    int out_off, s_out, b_off, aux_off;
    int s_aux, NF16, gout_lo, gout_hi;    // aux row stride; floats per 16-row block of the global buffer; its address
};
static_assert(sizeof(StageU) == 64, "StageU is read by one s_load_dwordx16");
struct FusedArgs {
    const StageU* stage_u;                // n_stages + 1 decoded records in device memory (as PersistArgs)
    int n_stages;
    int NL, L, Z, D;
    const float* X; long ldx; int rows;
    const float* Wimg; int img_total;     // LDS-image twin of the parameters in use (floats, multiple of 256):
                                          // copied verbatim to LDS offset 0 by LDS-DMA
    const float* eps_in;
    float* Zb_g;          // z,       blocked transposed [row block][rup16(Z)][16]
    float* XT;            // X tile copy, blocked transposed [row block][rup16(D)][16]
    int lds_x, s_x, lds_z, lds_e, s_z, lds_d0, lds_d1, s_d, lds_total;
    uint64_t seed; const DevState* st; long row0_global;
    float lo, hi, inv_batch;
    float* bce_p; float* kl_p;
    unsigned long long* dbg;      // diagnostic builds only (-DSMRT_STAMPS): phase time stamps of block 0
};
typedef void (*fused_kernel_t)(const FusedArgs);
fused_kernel_t fused_pick_kernel(const StageU* stages, int n_stages);      // the instantiation written out for this table, else the generic kernel
void launch_fused_fwd_bwd(hipStream_t s, fused_kernel_t k, const FusedArgs& a);
// Whether a stage table is a one-hidden-layer sigmoid VAE of the padded widths the written-out kernels exist for (in 16-wide
// blocks: features db, hidden hb, latent zb, both heads z2b); SMRT_HIP_PERSIST_SEQ=0 answers no.
bool seq_vae1_shape(const StageU* st, int n_stages, int& db, int& hb, int& zb, int& z2b);

// dW + optimizer fused (fused_small.hip)
struct DwTile {            // one 16x16 tile of one layer's [X | 1]^T . dPre: 64 bytes, read by one scalar load
    const float* A;      // layer input, blocked transposed [row block][rup16(fi)][16]
    const float* B;      // dPre,        blocked transposed [row block][rup16(fo)][16]
    long off;            // start of the layer's augmented matrix in the flat parameter buffer
    int fi, fo, ld, l2;
    int m0, n0;          // tile origin (fan-in row, fan-out column)
    int img_w, img_b, nn, pad;   // the layer's weight / bias image offsets and column count in the LDS-image twin
};
static_assert(sizeof(DwTile) == 64, "DwTile is read as one 64-byte record");
struct DwOptArgs {
    const DwTile* tiles;        // device table, total_tiles records
    int total_tiles, rows;
    float* W; const float* Wd; const float* mask;
    float* Wimg;                // LDS-image twin to keep current (null: the caller rebuilds it)
    float* s0; float* s1;
    float* grad_out;
    float* l2_partials;
    DevState* st;
    unsigned* ticket;
    unsigned long long* dbg;   // B: diagnostic builds only (-DSMRT_STAMPS)
    int opt, clip, has_drop, has_l2;
    float lr, l2, gmin, gmax, keep;
};
void launch_fused_dw_opt(hipStream_t s, const DwOptArgs& a);
struct DwSlabArgs {            // weight-gradient tiles of a row range into gradient slabs (fused path, larger mini-batches)
    const DwTile* tiles;
    int total_tiles, rows, blocks_per_split;
    float* G; long slab_stride;
};
void launch_fused_dw_slab(hipStream_t s, const DwSlabArgs& a, int splits);
// Persistent whole-epoch training kernel (persist_train.hip)
struct PersistArgs {
    const StageU* stage_u;                // n_stages + 1 decoded records (the last one a copy: read ahead, never run), 64-byte aligned
    int n_stages;
    int G;                                // compute workgroups (blockIdx < G); the rest of the grid are owner workgroups
    int NL, L, Z, D;
    const float* X; long ldx; long n_rows; int batch, n_steps;    // the epoch: n_steps contiguous mini-batches, ragged tail last
    float* Wimg; int img_total;           // LDS-image twin: read by every compute workgroup each step, written by the tile owners
    float* W; float* s0; float* s1;       // flat parameter / slot buffers: read at launch start, written at launch end
    const DwTile* tiles; int total_tiles;
    float* XT; float* Zb_g;               // blocked transposed X tile copy and z (the stage records carry the other buffers)
    const float* XT_all;                  // the WHOLE epoch's X, blocked transposed [step][16-row block][rup16(D)][16], built once per
                                          // smrt_vae_train_epochs call (launch_xt_all); null: the compute workgroups publish XT per step
    int cap_blocks;                       // 16-row blocks the blocked transposed workspace holds
    unsigned* cnt;                        // 64 words zeroed before every launch: [0] rows published, [16] weights, [32] error
    int lds_x, s_x, lds_z, lds_e, s_z, lds_total;
    uint64_t seed; DevState* st;
    float lo, hi;
    float* bce_p; float* kl_p;            // cost partials of the LAST mini-batch, one per compute workgroup
    float* l2_p;                          // ... and its L2 term, one per owner workgroup
    int opt, clip, has_l2;
    float lr, l2, gmin, gmax;
    unsigned long long* dbg;              // diagnostic builds only (-DSMRT_STAMPS): phase time stamps, step 4
};
void launch_xt_all(hipStream_t s, const float* X, long ldx, long n_rows, int batch, int G, int D, float* XT_all);
typedef void (*persist_kernel_t)(const PersistArgs);
persist_kernel_t persist_pick_kernel(const StageU* stages, int n_stages);   // the instantiation written out for this table, else the generic kernel
void launch_persist_train(hipStream_t s, persist_kernel_t k, const PersistArgs& a, int n_workgroups);
int  persist_set_max_lds(persist_kernel_t k, int bytes);
int  persist_owner_workgroups(int total_tiles);     // one owner wave per 16x16 weight-gradient tile, 4 per workgroup
int  persist_max_owner_workgroups();               // one flag word per owner workgroup

// Fused inference (encode / decode / generate) for nets whose weights fit in LDS: fused_small.hip
struct InferStage { int K16, N, N16, w_off, nn, b_off, act, pad; };   // image offsets relative to the staged range
struct InferArgs {
    InferStage st[MAX_LAYERS]; int n_stages;
    const float* img; int img_n4;         // staged range of the LDS-image twin: img_n4 16-byte pieces to LDS offset 0
    const float* in; long ldin; int K;    // input rows, row-major; K valid columns
    long n;                               // rows
    float* out0; long ld0; int split;     // output columns [0, split) -> out0, [split, N) -> out1 (encode: mu | log_sigma)
    float* out1; long ld1;
    int vec_in;                           // input rows are whole aligned quads (K % 4 == 0, ldin % 4 == 0, 16-byte aligned)
    int vec_out;                          // out0 takes whole 16-byte stores (split == N, N % 4 == 0, ld0 % 4 == 0, aligned)
    int tile_stride;                      // floats per row of the wave-private I/O tile: max(rup16(K), rup16(N_last)) + 4
    int maxt;                             // widest activation of the chain in 16-column tiles (kernel variant: 8 or 16)
    // generate: in == nullptr -> the input rows are DRAWN in the kernel, z = loc + scale * N(0,1) from the Philox stream
    // (seed, stream 3, step 0) at element gen_elem0 + row * K + k (K % 4 == 0 and gen_elem0 % 4 == 0: whole Philox blocks)
    uint64_t gen_seed, gen_elem0; float gen_loc, gen_scale;
    unsigned long long* dbg;              // diagnostic builds only (-DSMRT_STAMPS)
};
typedef void (*infer_kernel_t)(const InferArgs);
int  launch_fused_infer(hipStream_t s, const InferArgs& a);
int  dw_opt_blocks(int total_tiles);
int  fused_set_max_lds(fused_kernel_t k, int bytes);
void launch_build_image(hipStream_t s, const float* Wflat, const int* tab, float* img, int n_f4);

constexpr int SMALL_M_MAX = 512;    // mini-batches up to this many rows take the register-fed small-tile kernels (small_gemm.hip)
void launch_gemm(hipStream_t s, int kind, int epi, const GemmArgs& a);
void launch_small_gemm(hipStream_t s, int kind, int epi, const GemmArgs& a);
int  small_gemm_blocks(const GemmArgs& a);
void launch_small_dw(hipStream_t s, const DwArgs& a);
void launch_small_dw_opt(hipStream_t s, const DwFusedArgs& a);
int  gemm_grid_blocks(int kind, const GemmArgs& a);          // number of blocks (= BCE partial count)
void launch_dw(hipStream_t s, const DwArgs& a, bool big);
void launch_optimizer(hipStream_t s, const OptArgs& a, int* n_blocks_out);
int  optimizer_blocks(long P);
void launch_dropout(hipStream_t s, const DropArgs& a);
void launch_reparam(hipStream_t s, const float* ML, long ldml, const float* eps_in, float* Zb, float* Eb, long ldz,
                    int rows, int Z, uint64_t seed, const DevState* st, long row0_global, float* kl_partials);
int  reparam_blocks(int rows, int Z);
void launch_finalize(hipStream_t s, const FinalizeArgs& a);
void launch_reduce_slabs(hipStream_t s, const float* G, int splits, long slab_stride, long P, float* out);
void launch_sample_normal(hipStream_t s, float* out, long ld, long rows, int cols, uint64_t seed, uint32_t stream,
                          uint32_t step, uint64_t elem0, float loc, float scale);

}  // namespace smrt
