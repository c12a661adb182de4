// C ABI (include/smrt_hip.h): handle management, workspace, step scheduling.
#include "../../include/smrt_hip.h"
#include "vae_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

using namespace smrt;

static thread_local std::string g_err;
namespace smrt {
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
thread_local GraphBuild g_build;
thread_local Recorder g_record;
// kernel -> grouped twin (filled by static registration objects next to the kernels; read-only afterwards)
static std::vector<std::pair<const void*, MultiInfo>>& multi_registry() {
    static std::vector<std::pair<const void*, MultiInfo>> r;
    return r;
}
void smrt_register_multi(const void* kernel, const MultiInfo& info) { multi_registry().emplace_back(kernel, info); }
const MultiInfo* smrt_multi_of(const void* kernel) {
    for (auto& e : multi_registry()) if (e.first == kernel) return &e.second;
    return nullptr;
}
}
static int fail(int code, const std::string& m) { g_err = m; return code; }
int smrt_set_error(int code, const std::string& m) { return fail(code, m); }     // for diag.hip

#define HIPC(expr)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(-2, std::string(#expr) + ": " + hipGetErrorString(e__));                \
    } while (0)

static inline int rup4(int v) { return (v + 3) & ~3; }

// One device allocation carved into 256-byte aligned pieces.  Everything a training step touches sits in two
// arenas (parameters; per-batch workspace), each a multiple of 2 MiB: a short kernel then needs a couple of
// address translations instead of one per buffer -- in these few-microsecond launches a cold translation costs
// more than the arithmetic.
struct Arena { char* base = nullptr; size_t cap = 0, off = 0; };
struct LayerI { int fi, fo, ld; long off; int act; int l2; };

struct smrt_vae {
    smrt_vae_cfg cfg{};
    int device = 0;
    hipStream_t stream = nullptr;        // stream in use (own_stream unless the caller lent one)
    hipStream_t own_stream = nullptr;
    int L = 0, NL = 0, Z = 0, D = 0;
    bool ae = false;                     // plain auto-encoder (no latent heads, MSE loss)
    int hl = 0, dec0 = 0, Zc = 0;        // index of the fused heads layer (-2: none), first decoder layer, code width
    LayerI lay[MAX_LAYERS];
    long P_int = 0, P_abi = 0;
    int maxld = 0;
    bool has_drop = false, has_l2 = false, masks_injected = false;

    Arena pa, ta;                        // parameter arena, training-workspace arena
    float *W = nullptr, *Wd = nullptr, *mask = nullptr, *s0 = nullptr, *s1 = nullptr;
    float *G = nullptr, *gtmp = nullptr;
    int g_splits_cap = 0;
    DevState* st = nullptr;

    int cap_rows = 0;
    float* out[MAX_LAYERS] = {nullptr};
    float* dpre[MAX_LAYERS] = {nullptr};
    float *Zb = nullptr, *Eb = nullptr;
    // transposed workspace of the fused small-net path ([feature][capT], rows contiguous)
    float* outT[MAX_LAYERS] = {nullptr};
    float* dpreT[MAX_LAYERS] = {nullptr};
    float *XT = nullptr, *ZT = nullptr;
    float* xt_all = nullptr; size_t xt_all_cap = 0;      // the epoch's X, blocked transposed once per train_epochs call (persistent path)
    bool xt_all_on = false;
    float *bce_p = nullptr, *kl_p = nullptr, *l2_p = nullptr;
    int n_l2 = 0, l2_cap = 0;            // l2 partials of the optimizer sweep; capacity of l2_p
    float* cost_dev = nullptr; long cost_cap = 0;
    float* cost_pin = nullptr; long cost_pin_cap = 0;
    unsigned long long* time_pin = nullptr;   // epoch ring: time stamps (cost_pin_cap + 1 of them; FinalizeArgs::ring_time)
    unsigned* ep_ctr = nullptr;          // epoch ring: index of the running epoch (device)
    bool ring_on = false;                // steps that close an epoch write the ring (set for the duration of a train_epochs call)
    bool g_ring = false;                 // ... and the cached epoch graph was built that way
    unsigned* err_pin = nullptr;         // per-epoch copies of the persistent kernel's error word (same capacity as cost_pin)

    long icap = 0, izcap = 0;
    float* ibuf[2] = {nullptr, nullptr};
    float* izb = nullptr;

    int64_t steps = 0;
    int last_epoch_launches = 0;
    int use_graph = 1;
    std::vector<hipEvent_t> events;

    // cached epoch graph (valid while X, the schedule and the workspace pointers are unchanged)
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    const float* g_X = nullptr; long g_ldx = 0, g_n = 0, g_batch = 0;
    int g_launches = 0;                  // kernel nodes of the cached epoch graph

    // row-tile fused forward/backward (small nets): LDS plan
    bool fused = false;
    FusedArgs fplan{};
    FusedLayer flay[MAX_LAYERS];
    unsigned* ticket = nullptr;
    int tiles16 = 0;
    int* stage_tab = nullptr;        // image float offset of every 16-byte piece of the flat parameter buffer
    float* Wimg = nullptr;           // LDS-image twin of the parameters (fplan.img_total floats)
    bool img_valid = false;          // Wimg matches W (kept current by the fused optimizer epilogue)
    DwTile* tile_tab = nullptr;      // per-tile records of the fused dW + optimizer kernel
    StageU* stage_u = nullptr;       // the persistent kernel's stage records, decoded (ensure_train)
    int n_stage_u = 0;
    fused_kernel_t fkernel = nullptr;     // ... and the two-kernel step's forward / backward launch
    persist_kernel_t pkernel = nullptr;   // the persistent kernel for this net: a sequence-specialised instantiation or the generic one
    DwRec* dwrec = nullptr;          // small-batch fused weight-gradient + update launch: 32x32 tile records, then 64x64 ones
    int n_rec32 = 0;
    // persistent whole-epoch kernel (persist_train.hip)
    bool persist_ok = false;         // the net qualifies (fits in LDS, no dropout, VAE)
    unsigned* pcnt = nullptr;        // producer flags + error word (256 words)
    int persist_mode = -1;           // -1: first come, first served inside the device's budget; 0: never; 1: the caller
                                     //     has planned the budget itself (smrt_vae_set_persistent): always granted

    // per-kernel profiling (eager launches bracketed by events)
    bool prof = false;
    std::vector<hipEvent_t> prof_pool;
    std::vector<int> prof_kind;
    size_t prof_used = 0;
};

enum ProfKind : int { PK_DROPOUT = 0, PK_FWD, PK_FWD_BCE, PK_REPARAM, PK_BWD, PK_BWD_LATENT, PK_DW, PK_OPT,
                      PK_FINAL, PK_REDUCE, PK_FUSED, PK_FUSED_DW_OPT, PK_PERSIST, PK_COUNT };

static void drop_graph(smrt_vae* h) {
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->gexec = nullptr; h->graph = nullptr; h->g_X = nullptr;
}

// Bracket one launch with events when profiling; a no-op otherwise.
struct ProfScope {
    smrt_vae* h; 
    ProfScope(smrt_vae* hh, int kind) : h(hh) {
        if (!h->prof) return;
        if (h->prof_used + 2 > h->prof_pool.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; (void)hipEventCreate(&e); h->prof_pool.push_back(e); }
        }
        h->prof_kind.push_back(kind);
        g_prof_start = h->prof_pool[h->prof_used];           // stamped by the launch itself (SMRT_LAUNCH)
        g_prof_stop = h->prof_pool[h->prof_used + 1];
    }
    ~ProfScope() {
        if (!h->prof) return;
        g_prof_start = g_prof_stop = nullptr;
        h->prof_used += 2;
    }
};

// ------------------------------------------------------------------ layout helpers
static void build_layers(smrt_vae* h, int ld_align) {
    const smrt_vae_cfg& c = h->cfg;
    h->ae = (c.model == SMRT_MODEL_AE);
    h->L = c.n_hidden_layers; h->D = c.n_features;
    h->Z = h->ae ? 1 : c.n_latent;
    h->NL = h->ae ? 2 * h->L : 2 * h->L + 2;
    std::vector<int> w(1, c.n_features);
    for (int i = 0; i < h->L; ++i) w.push_back(c.hidden[i]);
    int li = 0; long off = 0, pabi = 0; int maxld = rup4(c.n_features);
    auto add = [&](int fi, int fo, int act, int l2) {
        LayerI& y = h->lay[li++];
        y.fi = fi; y.fo = fo; y.ld = (fo + ld_align - 1) / ld_align * ld_align; y.off = off; y.act = act; y.l2 = l2;
        off += (long)(fi + 1) * y.ld;
        maxld = std::max(maxld, y.ld);
    };
    for (int i = 0; i < h->L; ++i) { add(w[i], w[i + 1], c.activation, 1); pabi += (long)w[i] * w[i + 1] + w[i + 1]; }
    if (h->ae) {
        // SymmetricalAutoEncoderTopography (layer.py:163-190): the decoder mirrors the encoder, every layer with
        // the user's activation, every layer regularised
        h->hl = -2; h->dec0 = h->L; h->Zc = w[h->L];
        for (int i = h->L; i >= 1; --i) { add(w[i], w[i - 1], c.activation, 1); pabi += (long)w[i] * w[i - 1] + w[i - 1]; }
    } else {
        h->hl = h->L; h->dec0 = h->L + 1; h->Zc = h->Z;
        add(w[h->L], 2 * h->Z, c.activation, 0);          // fused heads: never regularised (layer.py:96)
        pabi += 2 * ((long)w[h->L] * h->Z + h->Z);
        add(h->Z, w[h->L], c.activation, 1); pabi += (long)h->Z * w[h->L] + w[h->L];
        for (int i = h->L; i >= 1; --i) {
            const int act = (i == 1) ? SMRT_ACT_SIGMOID : c.activation;       // layer.py:333-338
            add(w[i], w[i - 1], act, 1); pabi += (long)w[i] * w[i - 1] + w[i - 1];
        }
    }
    h->P_int = off; h->P_abi = pabi; h->maxld = std::max(maxld, rup4(h->Z));
}

// ABI flat <-> internal flat (host side).  to_internal=true: abi -> internal.
static void convert(const smrt_vae* h, float* abi, float* internal, bool to_internal) {
    long o = 0;
    auto plain = [&](const LayerI& y) {
        for (int r = 0; r <= y.fi; ++r)
            for (int c = 0; c < y.fo; ++c) {
                float& a = abi[o + (long)r * y.fo + c];
                float& b = internal[y.off + (long)r * y.ld + c];
                if (to_internal) b = a; else a = b;
            }
        o += (long)(y.fi + 1) * y.fo;
    };
    if (h->ae) { for (int l = 0; l < h->NL; ++l) plain(h->lay[l]); return; }
    for (int l = 0; l < h->L; ++l) plain(h->lay[l]);
    {
        const LayerI& y = h->lay[h->L]; const int Z = h->Z;
        for (int half = 0; half < 2; ++half) {
            for (int r = 0; r <= y.fi; ++r)
                for (int c = 0; c < Z; ++c) {
                    float& a = abi[o + (long)r * Z + c];
                    float& b = internal[y.off + (long)r * y.ld + half * Z + c];
                    if (to_internal) b = a; else a = b;
                }
            o += (long)(y.fi + 1) * Z;
        }
    }
    for (int l = h->L + 1; l < h->NL; ++l) plain(h->lay[l]);
}

static void fill_segs(const smrt_vae* h, OptSeg* seg) {
    long abi_w = 0;
    for (int l = 0; l < h->NL; ++l) {
        const LayerI& y = h->lay[l];
        OptSeg& s = seg[l];
        s.start = y.off; s.w_count = (long)y.fi * y.ld; s.total = (long)(y.fi + 1) * y.ld;
        s.ld = y.ld; s.fo = y.fo; s.l2 = y.l2; s.is_heads = (l == h->hl);
        s.abi_w_off = abi_w; s.Z = h->Z; s.hfi = y.fi;
        abi_w += (long)y.fi * y.fo;
    }
}

static inline int rup16(int v) { return (v + 15) & ~15; }

// LDS plan of the fused row-tile kernel; h->fused says whether the net is small enough to use it.
// The weight / bias images of all layers come first and contiguous (they are filled by one straight LDS-DMA copy of
// the device-resident image twin), the activation tiles follow.
static void plan_fused(smrt_vae* h) {
    FusedArgs& f = h->fplan;
    f = FusedArgs{};
    f.NL = h->NL; f.L = h->L; f.Z = h->Z; f.D = h->D;
    int off = 0, maxw = rup16(h->D);
    for (int l = 0; l < h->NL; ++l) {
        const LayerI& y = h->lay[l];
        FusedLayer& q = h->flay[l];
        q.fi = y.fi; q.fo = y.fo; q.ld = y.ld; q.act = y.act; q.w_off = (int)y.off;
        maxw = std::max(maxw, std::max(rup16(y.fo), rup16(y.fi)));
        q.nn = rup16(y.fo) + 1; q.lds_w = off; off += (rup16(y.fi) / 4) * q.nn * 4;
        q.lds_b = off; off += 4 * rup16(y.fo);
    }
    f.img_total = (off + 255) & ~255;                  // whole 1 KiB wave-instructions
    off = f.img_total;
    f.s_x = rup16(h->D) + 4; f.lds_x = off; off += 16 * f.s_x;
    for (int l = 0; l < h->NL - 1; ++l) {
        FusedLayer& q = h->flay[l];
        q.s_out = rup16(h->lay[l].fo) + 4; q.lds_out = off; off += 16 * q.s_out;
    }
    f.s_z = rup16(h->Z) + 4; f.lds_z = off; off += 16 * f.s_z; f.lds_e = off; off += 16 * f.s_z;
    f.s_d = maxw + 4; f.lds_d0 = off; off += 16 * f.s_d; f.lds_d1 = off; off += 16 * f.s_d;
    f.lds_total = off;
    h->tiles16 = 0;
    for (int l = 0; l < h->NL; ++l) h->tiles16 += ((h->lay[l].fi + 1 + 15) / 16) * ((h->lay[l].fo + 15) / 16);
    const char* e = std::getenv("SMRT_HIP_FUSED");
    const bool want = e ? std::atoi(e) != 0 : true;
    const int bytes = off * (int)sizeof(float);
    h->fused = want && bytes <= 150 * 1024 && !h->ae;
    if (h->fused && bytes > 48 * 1024 && fused_set_max_lds(fused_pick_kernel(nullptr, 0), bytes) != 0) { (void)hipGetLastError(); h->fused = false; }
    const char* pe = std::getenv("SMRT_HIP_PERSIST");
    h->persist_ok = h->fused && (pe ? std::atoi(pe) != 0 : true) && h->cfg.dropout_keep >= 1.0f;
    if (h->persist_ok && bytes > 48 * 1024 && persist_set_max_lds(persist_pick_kernel(nullptr, 0), bytes) != 0) { (void)hipGetLastError(); h->persist_ok = false; }
}

// Workgroups of the persistent kernel for a mini-batch size; 0: this schedule does not take the persistent path.
// One compute workgroup per 16 rows (+ one owner wave per weight-gradient tile, persist_owner_workgroups).  Co-residency
// of the whole grid is what its flags rely on: all persistent launches of the process on one device together are kept
// to persist_budget(device) workgroups (one workgroup per CU: LDS), the rest take the two-kernel step.
constexpr int PERSIST_ROWS_MAX = 512;
constexpr int SMRT_MAX_DEVICES = 64;
static std::atomic<int> g_persist_wgs[SMRT_MAX_DEVICES];
// Co-resident persistent workgroups a process may hold on one device: 5/8 of the device's CUs (160 of 256; the rest stays
// free for the other streams' kernels), divided by the number of processes that share the GPU when a launcher says so
// (LOCAL_WORLD_SIZE ranks over fewer visible devices: bench.py's one-GPU rehearsal of the multi-rank paths), or
// SMRT_HIP_PERSIST_BUDGET.  A grid that would not be co-resident deadlocks its flag waits (bounded: 2 s, then -6 and the
// two-launch fallback), so the figure errs low.
static int persist_budget(int device) {
    static int cached[SMRT_MAX_DEVICES];
    static std::once_flag once[SMRT_MAX_DEVICES];
    if (device < 0 || device >= SMRT_MAX_DEVICES) return 0;
    std::call_once(once[device], [device]() {
        int cus = 256, ndev = 1;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        (void)hipGetDeviceCount(&ndev);
        int budget = cus * 5 / 8;
        const char* lws = std::getenv("LOCAL_WORLD_SIZE");
        const int ranks = lws ? std::max(1, std::atoi(lws)) : 1;
        const int share = (ranks + std::max(ndev, 1) - 1) / std::max(ndev, 1);        // ranks per visible GPU
        budget /= std::max(share, 1);
        cached[device] = budget;
    });
    const char* e = std::getenv("SMRT_HIP_PERSIST_BUDGET");          // (read per call: tests vary it)
    return (e && *e) ? std::max(0, std::atoi(e)) : cached[device];
}
static int persist_grid(const smrt_vae* h, long batch_rows) {
    if (!h->persist_ok || h->persist_mode == 0 || batch_rows > PERSIST_ROWS_MAX || batch_rows < 1) return 0;
    return (int)((batch_rows + 15) / 16);
}

// Zero-fills go on the handle's stream and are waited for there: a device-wide synchronise is refused by the runtime
// while ANY stream of the process is being captured (a caller's own graph capture, for instance), and several host
// threads drive handles at the same time.
template <class T> static int dalloc(T** p, size_t n_elems, hipStream_t s) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (n_elems == 0) return 0;
    hipError_t e = hipMalloc((void**)p, n_elems * sizeof(T));
    if (e != hipSuccess) return fail(-3, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(*p, 0, n_elems * sizeof(T), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(-3, std::string("zero-fill: ") + hipGetErrorString(e));
    return 0;
}
#define TRY(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)

template <class T> static void carve(Arena& a, T** p, size_t n_elems) {
    *p = (n_elems && a.base) ? reinterpret_cast<T*>(a.base + a.off) : nullptr;
    a.off += (n_elems * sizeof(T) + 255) & ~size_t(255);
}
static int arena_alloc(Arena& a, size_t bytes, hipStream_t s) {
    if (a.base) { (void)hipFree(a.base); a.base = nullptr; }
    a.cap = (bytes + (2u << 20) - 1) & ~size_t((2u << 20) - 1);
    hipError_t e = hipMalloc((void**)&a.base, a.cap);
    if (e != hipSuccess) { a.base = nullptr; return fail(-3, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemsetAsync(a.base, 0, a.cap, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(-3, std::string("arena zero-fill: ") + hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------ dW tiling / split-K policy
static int dw_tiles(const smrt_vae* h, int t) {
    int tot = 0;
    for (int l = 0; l < h->NL; ++l)
        tot += ((h->lay[l].fi + t - 1) / t) * ((h->lay[l].fo + t - 1) / t);     // bias row: column sums, no tile
    return tot;
}
static bool dw_big(const smrt_vae* h, int rows) { return rows >= 2048 && dw_tiles(h, 128) >= 48; }
// Work units of the tiled weight-gradient launch for a base split factor S (the number of K ranges of an INTERIOR tile):
// a tile whose busiest wave multiplies `cost` of 4 blocks (gemm_tile's arrangements; 4 = full tile) is cut into
// ceil(S min(4, cost + pen) / 4) ranges, so that every workgroup runs about equally long.  pen = 1 block-time is what the
// bounds-checked staging of an edge tile costs on top of its MFMAs (measured, N1 at batch 65 536: dW per step 1.09 ms with
// uniform ranges, 1.24 with pen 0, 1.04 / 0.98 / 1.04 with pen 0.5 / 1 / 1.5-2; SMRT_DW_EDGE_PEN overrides, < 0 = uniform).
// Returns the number of units; *t_rel = the longest workgroup's time relative to a full tile over all rows.
static float dw_edge_pen() {
    static const float v = []() { const char* e = std::getenv("SMRT_DW_EDGE_PEN"); return e ? (float)std::atof(e) : 1.0f; }();
    return v;
}
static int plan_dw_units(const smrt_vae* h, bool big, int S, DwArgs* d, double* t_rel) {
    const int t = big ? 128 : 64;
    int units = 0;
    double worst = 0.0;
    for (int l = 0; l < h->NL; ++l) {
        const int fi = h->lay[l].fi, fo = h->lay[l].fo;
        const int tm = (fi + t - 1) / t, tn = (fo + t - 1) / t;
        int s_cls[4] = {S, S, S, S};
        if (big && dw_edge_pen() >= 0.f) {
            const int MBl = std::min(4, (fi - (tm - 1) * 128 + 31) / 32), NBl = std::min(4, (fo - (tn - 1) * 128 + 31) / 32);
            auto cost = [](int MB, int NB) { return std::min(std::min(std::min(MB, 2) * std::min(NB, 2), MB), NB); };
            const int mb[4] = {4, 4, MBl, MBl}, nb[4] = {4, NBl, 4, NBl};
            for (int c = 0; c < 4; ++c) {
                const bool edge = mb[c] < 4 || nb[c] < 4;
                const double w = edge ? std::min(4.0, cost(mb[c], nb[c]) + (double)dw_edge_pen()) : 4.0;
                s_cls[c] = std::max(1, std::min(S, (int)std::ceil(S * w / 4.0 - 1e-9)));
                const int n_c = c == 0 ? (tm - 1) * (tn - 1) : (c == 1 ? tm - 1 : (c == 2 ? tn - 1 : 1));
                if (n_c > 0) worst = std::max(worst, w / 4.0 / s_cls[c]);
            }
        } else {
            worst = std::max(worst, 1.0 / S);
        }
        const int n_c[4] = {(tm - 1) * (tn - 1), tm - 1, tn - 1, 1};
        if (d) {
            DwLayer& q = d->L[l];
            q.tiles_m = tm; q.tiles_n = tn;
            for (int c = 0; c < 4; ++c) {
                q.s_cls[c] = s_cls[c];
                for (int y = 0; y < s_cls[c]; ++y) d->ycum[y + 1] += n_c[c];        // (per-range counts; prefix-summed below)
            }
        }
        for (int c = 0; c < 4; ++c) units += n_c[c] * s_cls[c];
    }
    if (d) for (int y = 1; y <= 32; ++y) d->ycum[y] += d->ycum[y - 1];
    if (t_rel) *t_rel = worst;
    return units;
}
// Split-K factor of the weight-gradient launch: the launch lasts rounds x (longest workgroup), rounds = units over the 512
// workgroups the chip holds at once (256 CUs x 2): 440 tiles x 3 splits is 2.6 rounds and runs as long as 3.
static int choose_splits(const smrt_vae* h, int rows) {
    const bool big = dw_big(h, rows);
    const int slots = 512;
    const int smax = std::max(1, std::min(32, rows / 512));          // keep >= 512 rows (16 K-tiles) per split
    int best = 1; double best_t = 1e30;
    for (int s = 1; s <= smax; ++s) {
        double t_rel;
        const int units = plan_dw_units(h, big, s, nullptr, &t_rel);
        const double t = std::ceil((double)units / slots) * t_rel;
        if (t < best_t * 0.98) { best_t = t; best = s; }             // smaller split (less slab traffic) on near ties
    }
    return best;
}

// Fused forward/backward kernel beyond 512 rows (nets that fit in LDS): up to this many rows one 16-row workgroup per
// CU and round still beats the per-layer GEMM chain, whose launches are latency-bound at these widths.
constexpr int FUSED_ROWS_MAX = 8192;
// Row ranges of the slab weight-gradient kernel: at least 8 row blocks (one batch of loads) per range, at most 32 ranges.
static int fused_slab_splits(int rows) {
    const int nblk = (rows + 15) / 16;
    return std::max(1, std::min(32, nblk / 8));
}

static int build_stages(const smrt_vae* h, const FusedArgs& f, FusedStage* stages);
static int ensure_train(smrt_vae* h, int rows) {
    int splits = choose_splits(h, rows);
    if (h->fused && rows > 512 && rows <= FUSED_ROWS_MAX) splits = std::max(splits, fused_slab_splits(rows));
    if (splits <= h->g_splits_cap && rows <= h->cap_rows) return 0;
    drop_graph(h);
    HIPC(hipStreamSynchronize(h->stream));
    const int nsplit = std::max(splits, h->g_splits_cap);
    const int cap = (std::max(std::max(rows, h->cap_rows), (int)h->cfg.max_batch) + 15) & ~15;   // fused kernels store whole 16-row tiles
    // loss partials: one per workgroup of the output-layer GEMM (64x64 tiles, or down to 16x16 in the small-batch kernels)
    const int nb = std::max(std::max(((cap + 63) / 64) * ((h->D + 63) / 64), (cap + 15) / 16),
                            ((std::min(cap, SMALL_M_MAX) + 15) / 16) * ((h->D + 15) / 16));
    auto layout = [&](Arena& a) {
        a.off = 0;
        carve(a, &h->G, (size_t)nsplit * h->P_int);
        for (int l = 0; l < h->NL; ++l) {
            carve(a, &h->out[l], (l < h->NL - 1) ? (size_t)cap * h->lay[l].ld : 0);
            carve(a, &h->dpre[l], (size_t)cap * h->lay[l].ld);
        }
        carve(a, &h->Zb, (size_t)cap * rup4(h->Z));
        carve(a, &h->Eb, (size_t)cap * rup4(h->Z));
        carve(a, &h->bce_p, (size_t)nb);
        carve(a, &h->kl_p, (size_t)std::max(reparam_blocks(cap, h->Z), (cap + 15) / 16));
        carve(a, &h->dwrec, (size_t)h->n_rec32);
        if (h->fused) {
            for (int l = 0; l < h->NL; ++l) {
                carve(a, &h->outT[l], (l < h->NL - 1) ? (size_t)cap * rup16(h->lay[l].fo) : 0);
                carve(a, &h->dpreT[l], (size_t)cap * rup16(h->lay[l].fo));
            }
            carve(a, &h->XT, (size_t)cap * rup16(h->D));
            carve(a, &h->ZT, (size_t)cap * rup16(h->Z));
            carve(a, &h->tile_tab, (size_t)h->tiles16);
            carve(a, &h->stage_u, (size_t)(2 * MAX_LAYERS + 1));
        }
    };
    h->n_rec32 = 0;
    for (int l = 0; l < h->NL; ++l) {
        h->n_rec32 += ((h->lay[l].fi + 1 + 31) / 32) * ((h->lay[l].fo + 31) / 32);
    }
    Arena dry;
    layout(dry);
    TRY(arena_alloc(h->ta, dry.off, h->stream));
    layout(h->ta);
    h->g_splits_cap = nsplit;
    h->cap_rows = cap;
    {   // tile records of the small-batch fused weight-gradient + update launch (the workspace pointers are fixed from here on)
        std::vector<DwRec> recs;
        const int t = 32;
        for (int l = 0; l < h->NL; ++l) {
            const LayerI& y = h->lay[l];
            const bool from_x = (l == 0), from_z = (l == h->hl + 1 && !h->ae);
            for (int m0 = 0; m0 <= y.fi; m0 += t)
                for (int n0 = 0; n0 < y.fo; n0 += t) {
                    DwRec r{};
                    r.A = from_x ? nullptr : (from_z ? h->Zb : h->out[l - 1]);
                    r.lda = from_x ? 0 : (from_z ? rup4(h->Z) : h->lay[l - 1].ld);
                    r.B = h->dpre[l]; r.ldb = y.ld; r.ldc = y.ld; r.off = y.off; r.fi = y.fi; r.fo = y.fo;
                    r.m0 = m0; r.n0 = n0; r.l2 = y.l2;
                    recs.push_back(r);
                }
        }
        if ((int)recs.size() != h->n_rec32) return fail(-4, "internal: weight-gradient record count mismatch");
        HIPC(hipMemcpy(h->dwrec, recs.data(), recs.size() * sizeof(DwRec), hipMemcpyHostToDevice));
    }
    if (h->fused) {
        std::vector<DwTile> tiles;
        for (int l = 0; l < h->NL; ++l) {
            const LayerI& y = h->lay[l];
            const FusedLayer& q = h->flay[l];
            for (int m0 = 0; m0 <= y.fi; m0 += 16)
                for (int n0 = 0; n0 < y.fo; n0 += 16) {
                    DwTile t{};
                    t.A = (l == 0) ? h->XT : (l == h->L + 1 ? h->ZT : h->outT[l - 1]); t.B = h->dpreT[l];
                    t.off = y.off; t.fi = y.fi; t.fo = y.fo; t.ld = y.ld; t.l2 = y.l2; t.m0 = m0; t.n0 = n0;
                    t.img_w = q.lds_w; t.img_b = q.lds_b; t.nn = q.nn;
                    tiles.push_back(t);
                }
        }
        if ((int)tiles.size() != h->tiles16) return fail(-4, "internal: tile table size mismatch");
        HIPC(hipMemcpy(h->tile_tab, tiles.data(), tiles.size() * sizeof(DwTile), hipMemcpyHostToDevice));
        // the persistent kernel's stage records, decoded here once (they name workspace buffers: rebuilt with them)
        FusedStage st[2 * MAX_LAYERS];
        StageU su[2 * MAX_LAYERS + 1];
        const int ns = build_stages(h, h->fplan, st);
        for (int i = 0; i < ns; ++i) {
            const FusedStage& t = st[i];
            StageU& u = su[i];
            u.in_off = t.in_off; u.sin = t.sin; u.nb_all = t.K16 >> 4; u.w_off = t.w_off;
            u.nn = t.nn; u.N = t.N; u.packed = t.kind | (t.act << 8) | (t.trans << 16) | (t.variant << 24); u.N16 = rup16(t.N);
            u.out_off = t.out_off; u.s_out = t.s_out; u.b_off = t.b_off; u.aux_off = t.aux_off;
            u.s_aux = t.s_aux; u.NF16 = t.NF * 16;
            const unsigned long long gp = (unsigned long long)(uintptr_t)t.gout;
            u.gout_lo = (int)(unsigned)(gp & 0xffffffffull); u.gout_hi = (int)(unsigned)(gp >> 32);
        }
        su[ns] = su[ns - 1];
        h->n_stage_u = ns;
        h->fkernel = fused_pick_kernel(su, ns);
        if (h->fkernel != fused_pick_kernel(nullptr, 0) && h->fplan.lds_total * (int)sizeof(float) > 48 * 1024 &&
            fused_set_max_lds(h->fkernel, h->fplan.lds_total * (int)sizeof(float)) != 0) {
            (void)hipGetLastError();
            h->fkernel = fused_pick_kernel(nullptr, 0);
        }
        h->pkernel = persist_pick_kernel(su, ns);
        if (h->pkernel != persist_pick_kernel(nullptr, 0) && h->fplan.lds_total * (int)sizeof(float) > 48 * 1024 &&
            persist_set_max_lds(h->pkernel, h->fplan.lds_total * (int)sizeof(float)) != 0) {
            (void)hipGetLastError();
            h->pkernel = persist_pick_kernel(nullptr, 0);
        }
        HIPC(hipMemcpy(h->stage_u, su, (size_t)(ns + 1) * sizeof(StageU), hipMemcpyHostToDevice));
    }
    return 0;
}

// Inference buffers: the latent chunk `izb` (generate) and, for the per-layer path only, two ping-pong activation chunks.
static int ensure_infer(smrt_vae* h, long rows, bool need_ibuf = true) {
    if (rows > h->izcap) {
        HIPC(hipStreamSynchronize(h->stream));
        TRY(dalloc(&h->izb, (size_t)rows * rup4(h->Z), h->stream));
        h->izcap = rows;
    }
    if (need_ibuf && rows > h->icap) {
        HIPC(hipStreamSynchronize(h->stream));
        TRY(dalloc(&h->ibuf[0], (size_t)rows * h->maxld, h->stream));
        TRY(dalloc(&h->ibuf[1], (size_t)rows * h->maxld, h->stream));
        h->icap = rows;
    }
    return 0;
}

static int check_mat(const void* p, int64_t ld, int cols, const char* what) {
    if (!p) return fail(-1, std::string(what) + ": null pointer");
    if (((uintptr_t)p & 15) != 0) return fail(-1, std::string(what) + ": pointer must be 16-byte aligned");
    if (ld % 4 != 0 || ld < cols) return fail(-1, std::string(what) + ": leading dimension must be a multiple of 4 and >= columns");
    return 0;
}

// Stage table of the LDS-resident kernels: forward through every layer, then backward through the activations.
static int build_stages(const smrt_vae* h, const FusedArgs& f, FusedStage* stages) {
    const int L = h->L, NL = h->NL;
    int ns = 0;
    for (int l = 0; l < NL; ++l) {
        const FusedLayer& q = h->flay[l];
        FusedStage& t = stages[ns++];
        t = FusedStage{};
        t.in_off = (l == 0) ? f.lds_x : (l == L + 1 ? f.lds_z : h->flay[l - 1].lds_out);
        t.sin = (l == 0) ? f.s_x : (l == L + 1 ? f.s_z : h->flay[l - 1].s_out);
        t.K16 = rup16(q.fi); t.w_off = q.lds_w; t.nn = q.nn; t.N = q.fo; t.act = q.act; t.b_off = q.lds_b;
        t.trans = 0; t.NF = rup16(q.fo);
        if (l < NL - 1) {
            t.kind = (l == L) ? FK_FWD_HEADS : FK_FWD_ACT;
            t.out_off = q.lds_out; t.s_out = q.s_out; t.gout = h->outT[l];
        } else {
            t.kind = FK_FWD_BCE; t.out_off = f.lds_d0; t.s_out = f.s_d; t.aux_off = f.lds_x; t.s_aux = f.s_x;
            t.gout = h->dpreT[l];
        }
    }
    int cur = f.lds_d0, nxt = f.lds_d1;
    for (int l = NL - 1; l >= 1; --l) {
        const FusedLayer& q = h->flay[l];
        const FusedLayer& pq = h->flay[l - 1];
        FusedStage& t = stages[ns++];
        t = FusedStage{};
        t.in_off = cur; t.sin = f.s_d; t.K16 = rup16(q.fo); t.w_off = q.lds_w; t.nn = q.nn; t.trans = 1;
        t.N = q.fi; t.act = pq.act; t.out_off = nxt; t.s_out = f.s_d;
        t.aux_off = pq.lds_out; t.s_aux = pq.s_out; t.NF = rup16(pq.fo); t.gout = h->dpreT[l - 1];
        t.kind = (l == L + 1) ? FK_BWD_LATENT : FK_BWD_ACT;
        std::swap(cur, nxt);
    }
    for (int i = 0; i < ns; ++i) stages[i].variant = stage_variant(stages[i].kind, stages[i].K16, stages[i].act);
    return ns;
}

// One whole epoch (every contiguous mini-batch, ragged tail last) in ONE persistent launch + the cost of the last batch.
// Returns the number of launches enqueued.
static int enqueue_persist_epoch(smrt_vae* h, const float* X, long ldx, long n_rows, int batch, int G, float* cost_slot,
                                 bool inject_fault = false) {
    hipStream_t s = h->stream;
    const FusedArgs& f = h->fplan;
    PersistArgs a{};
    a.n_stages = h->n_stage_u; a.stage_u = h->stage_u;
    const int n_owner = persist_owner_workgroups(h->tiles16);
    a.G = G;
    a.NL = h->NL; a.L = h->L; a.Z = h->Z; a.D = h->D;
    a.X = X; a.ldx = ldx; a.n_rows = n_rows; a.batch = batch; a.n_steps = (int)((n_rows + batch - 1) / batch);
    a.Wimg = h->Wimg; a.img_total = f.img_total; a.W = h->W; a.s0 = h->s0; a.s1 = h->s1;
    a.tiles = h->tile_tab; a.total_tiles = h->tiles16; a.XT = h->XT; a.Zb_g = h->ZT; a.cap_blocks = h->cap_rows / 16;
    a.XT_all = h->xt_all_on ? h->xt_all : nullptr;
    a.cnt = h->pcnt;
    a.lds_x = f.lds_x; a.s_x = f.s_x; a.lds_z = f.lds_z; a.lds_e = f.lds_e; a.s_z = f.s_z; a.lds_total = f.lds_total;
    a.seed = h->cfg.seed; a.st = h->st; a.lo = h->cfg.bce_lo; a.hi = h->cfg.bce_hi;
    a.bce_p = h->bce_p; a.kl_p = h->kl_p; a.l2_p = h->l2_p;
    a.opt = h->cfg.optimizer; a.clip = h->cfg.clip; a.has_l2 = h->has_l2 ? 1 : 0;
    a.lr = h->cfg.learning_rate; a.l2 = h->has_l2 ? h->cfg.l2_penalty : 0.f;
    a.gmin = h->cfg.gclip_min; a.gmax = h->cfg.gclip_max;
    // The producer flags (words [0, 192)) start at zero for every launch (stream-ordered, no host wait).  The error word
    // (192) is STICKY across the launches of one smrt_vae_train_epochs call: it is cleared once, at the start of the call,
    // every workgroup reads it in its prologue and leaves at once when it is set -- an epoch that follows a timed-out one
    // must not train on the half-updated image twin, and the host must still see the error after the last epoch.
    // (With the epoch ring the launch that closes the previous epoch has cleared them, and the call's first launch follows
    // train_epochs' own clear of all 256 words.)
    if (!h->ring_on && hipMemsetAsync(h->pcnt, 0, 192 * sizeof(unsigned), s) != hipSuccess) return -1;
    // test hook (SMRT_HIP_PERSIST_FAULT_EPOCH): raise the error word in front of this launch, as a peer's time-out would
    if (inject_fault && hipMemsetAsync(h->pcnt + 192, 1, sizeof(unsigned), s) != hipSuccess) return -1;
#ifdef SMRT_STAMPS
    static unsigned long long* pdbg = nullptr;
    if (!pdbg) { (void)hipMalloc((void**)&pdbg, 160 * 8); (void)hipMemset(pdbg, 0, 160 * 8); }
    a.dbg = pdbg;
#endif
    { ProfScope ps(h, PK_PERSIST); launch_persist_train(s, h->pkernel ? h->pkernel : persist_pick_kernel(nullptr, 0), a, G + n_owner); }
#ifdef SMRT_STAMPS
    {
        (void)hipStreamSynchronize(s);
        unsigned long long hb[160];
        (void)hipMemcpy(hb, pdbg, 160 * 8, hipMemcpyDeviceToHost);
        static int once = 0;
        if (once++ == 2) {
            for (int st = 0; st < 12 && hb[64 + 8 * st]; ++st) {
                const unsigned long long* t = hb + 64 + 8 * st;
                fprintf(stderr, "stage %d wave 0, shader cycles: decode + addresses %lld | fragment addresses %lld | reads + MFMAs + sums %lld | epilogue %lld | barrier %lld\n",
                        st, (long long)(t[0] - (st ? t[-4] : t[0])), (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]));
            }
            const char* nm[64] = {nullptr};
            nm[1] = "weights published (poll)"; nm[2] = "image -> LDS"; nm[3] = "x tile, barrier";
            nm[40] = "stages done"; nm[41] = "stores drained + barrier";
            unsigned long long prev = hb[0];
            for (int i = 1; i < 42; ++i) {
                if (!hb[i]) continue;
                char buf[64];
                snprintf(buf, sizeof buf, "stage %d + barrier", i - 4);
                fprintf(stderr, "persist stamp %2d %-40s +%.2f us\n", i, nm[i] ? nm[i] : buf, (double)(hb[i] - prev) * 0.01);
                prev = hb[i];
            }
            fprintf(stderr, "owner: wait for rows %+.2f us after compute's signal | loads + MFMA %.2f | optimizer, stores, drain %.2f\n",
                    ((double)hb[46] - (double)hb[41]) * 0.01, (double)(hb[47] - hb[46]) * 0.01, (double)(hb[48] - hb[47]) * 0.01);
            fprintf(stderr, "persist step (compute workgroup 0, start -> signal) %.2f us\n", (double)(hb[41] - hb[0]) * 0.01);
        }
    }
#endif
    const long rows_last = n_rows - (long)(a.n_steps - 1) * batch;
    FinalizeArgs fa{};
    fa.bce = h->bce_p; fa.n_bce = (int)((rows_last + 15) / 16); fa.kl = h->kl_p; fa.n_kl = fa.n_bce;
    fa.l2p = h->l2_p; fa.n_l2 = n_owner; fa.inv_batch = 1.0f / (float)rows_last; fa.l2 = a.l2;
    fa.cost_out = cost_slot; fa.st = h->st; fa.advance = 0;
    if (h->ring_on) {
        fa.ring_cost = h->cost_pin; fa.ring_time = h->time_pin; fa.ring_ctr = h->ep_ctr;
        fa.err_src = h->pcnt + 192; fa.ring_err = h->err_pin; fa.zero192 = h->pcnt;
    }
    { ProfScope ps(h, PK_FINAL); launch_finalize(s, fa); }
    return 2;
}

// ------------------------------------------------------------------ one training step, enqueued async
struct StepOut { float* cost_slot; float* grad_out; float* dp_out; };

// mode 0: full step (optimizer + finalize).  mode 1: gradients only (data-parallel first half).
static int enqueue_step(smrt_vae* h, const float* X, long ldx, int rows, float inv_batch, long row0_global,
                        const float* eps_in, int mode, const StepOut& o) {
    hipStream_t s = h->stream;
    const int L = h->L, NL = h->NL, Z = h->Z;
    const float* Wuse = h->W;
    int launches = 0;
    OptSeg segs[MAX_LAYERS];
    fill_segs(h, segs);

    if (h->has_drop) {
        DropArgs d{};
        std::memcpy(d.seg, segs, sizeof(OptSeg) * NL); d.n_seg = NL; d.P = h->P_int;
        d.W = h->W; d.Wd = h->Wd; d.mask = h->mask; d.st = h->st; d.seed = h->cfg.seed;
        d.keep = h->cfg.dropout_keep; d.injected = h->masks_injected ? 1 : 0;
        { ProfScope ps(h, PK_DROPOUT); launch_dropout(s, d); } ++launches;
        Wuse = h->Wd;
    }
    // the launch that closes an epoch of a train_epochs call reports cost and time through the epoch ring (FinalizeArgs)
    auto ring = [&](FinalizeArgs& f) {
        if (h->ring_on && o.cost_slot == h->cost_dev) { f.ring_cost = h->cost_pin; f.ring_time = h->time_pin; f.ring_ctr = h->ep_ctr; }
    };
    const int HL = h->hl;               // fused latent heads layer (-2 for the plain auto-encoder: never matches)
    auto in_of = [&](int l) -> const float* { return l == 0 ? X : (l == HL + 1 ? h->Zb : h->out[l - 1]); };
    auto ldin_of = [&](int l) -> long { return l == 0 ? ldx : (l == HL + 1 ? (long)rup4(Z) : (long)h->lay[l - 1].ld); };

    int n_bce = 0;
    int n_kl = h->ae ? 0 : reparam_blocks(rows, Z);
    const bool fused_big = h->fused && mode == 0 && rows > 512 && rows <= FUSED_ROWS_MAX &&
                           fused_slab_splits(rows) <= h->g_splits_cap;    // fused fwd/bwd + slab dW + optimizer sweep
    const bool fused_small = h->fused && mode == 0 && rows <= 512;      // two-kernel step on the transposed workspace
    const bool fused_step = fused_small || fused_big;
    if (fused_step) {
        if (h->has_drop) {           // the image follows the post-dropout weights of this step
            launch_build_image(s, h->Wd, h->stage_tab, h->Wimg, (int)(h->P_int / 4)); ++launches;
        } else if (!h->img_valid || fused_big) {     // (the optimizer sweep of the large-batch path does not write the image)
            launch_build_image(s, h->W, h->stage_tab, h->Wimg, (int)(h->P_int / 4)); ++launches;
            h->img_valid = true;
        }
        FusedArgs f = h->fplan;
        f.n_stages = h->n_stage_u; f.stage_u = h->stage_u;
        f.X = X; f.ldx = ldx; f.rows = rows; f.eps_in = eps_in; f.Wimg = h->Wimg; f.Zb_g = h->ZT;
        f.XT = h->XT;
        f.seed = h->cfg.seed; f.st = h->st; f.row0_global = row0_global;
        f.lo = h->cfg.bce_lo; f.hi = h->cfg.bce_hi; f.inv_batch = inv_batch; f.bce_p = h->bce_p; f.kl_p = h->kl_p;
#ifdef SMRT_STAMPS
        static unsigned long long* dbgbuf = nullptr;
        if (!dbgbuf) { (void)hipMalloc((void**)&dbgbuf, 64 * 8); (void)hipMemset(dbgbuf, 0, 64 * 8); }
        f.dbg = dbgbuf;
#endif
        { ProfScope ps(h, PK_FUSED); launch_fused_fwd_bwd(s, h->fkernel ? h->fkernel : fused_pick_kernel(nullptr, 0), f); } ++launches;
#ifdef SMRT_STAMPS
        {   // diagnostic build: print block 0's phase durations of the 4th launch (wall_clock64 = 100 MHz)
            (void)hipStreamSynchronize(s);
            unsigned long long hb[64];
            (void)hipMemcpy(hb, dbgbuf, 64 * 8, hipMemcpyDeviceToHost);
            static int once = 0;
            if (once++ == 3) for (int i = 1; i < 32 && hb[i]; ++i)
                fprintf(stderr, "stamp %d: +%.2f us  (%.0f shader cycles, %.2f GHz)\n", i, (double)(hb[i] - hb[i - 1]) * 0.01,
                        (double)(hb[32 + i] - hb[32 + i - 1]), (double)(hb[32 + i] - hb[32 + i - 1]) / ((double)(hb[i] - hb[i - 1]) * 10.0));
        }
#endif
        n_bce = n_kl = (rows + 15) / 16;
    }
    // The middle of the VAE at small mini-batches: heads + reparameterisation + first decoder layer in ONE launch forward,
    // first decoder layer + latent + heads in ONE launch backward (small_gemm.hip: mid_fwd / mid_bwd; 2 Z <= 32).
    static const bool mid_env = []() { const char* e = std::getenv("SMRT_HIP_MID"); return e ? std::atoi(e) != 0 : true; }();
    const bool mid = mid_env && !fused_step && !h->ae && rows <= SMALL_M_MAX && 2 * Z <= 32 && L >= 1;
    MidArgs ma{};
    if (mid) {
        const LayerI& yh = h->lay[HL];
        const LayerI& yd = h->lay[HL + 1];
        ma.H = in_of(HL); ma.ldh = ldin_of(HL);
        ma.Wh = Wuse + yh.off; ma.ldwh = yh.ld; ma.bh = h->W + yh.off + (long)yh.fi * yh.ld;
        ma.Wd = Wuse + yd.off; ma.ldwd = yd.ld; ma.bd = h->W + yd.off + (long)yd.fi * yd.ld;
        ma.ML = h->out[HL]; ma.ldml = yh.ld;
        ma.Zb = h->Zb; ma.Eb = h->Eb; ma.ldz = rup4(Z);
        ma.OUT = h->out[HL + 1]; ma.ldo = yd.ld;
        ma.eps_in = eps_in; ma.kl_partials = h->kl_p;
        ma.DP = h->dpre[HL + 1]; ma.lddp = yd.ld;
        ma.DML = h->dpre[HL]; ma.lddml = yh.ld;
        ma.DOUT = h->dpre[HL - 1]; ma.lddo = h->lay[HL - 1].ld;
        ma.M = rows; ma.hL = yh.fi; ma.Z = Z; ma.act = yh.act; ma.inv_batch = inv_batch;
        ma.seed = h->cfg.seed; ma.st = h->st; ma.row0_global = row0_global;
        n_kl = mid_row_blocks(rows, yh.fi);
    }
    // ---- forward
    for (int l = 0; l < NL && !fused_step; ++l) {
        if (mid && l == HL) {
#ifdef SMRT_STAMPS
            static unsigned long long* mdbg = nullptr;
            if (!mdbg) { (void)hipMalloc((void**)&mdbg, 64 * 8); (void)hipMemset(mdbg, 0, 64 * 8); }
            ma.dbg = mdbg;
#endif
            { ProfScope ps(h, PK_FWD); launch_mid_fwd(s, ma); } ++launches;
#ifdef SMRT_STAMPS
            {
                (void)hipStreamSynchronize(s);
                unsigned long long hb[64];
                (void)hipMemcpy(hb, mdbg, 64 * 8, hipMemcpyDeviceToHost);
                static int once = 0;
                if (once++ == 4) for (int b = 0; b < 3; ++b)
                    fprintf(stderr, "mid_fwd wg %s: start +%.2f us after wg 0 | heads tile %.2f | reparam %.2f | dec1 tile %.2f\n",
                            b == 0 ? "first" : (b == 1 ? "middle" : "last"), ((double)hb[16 * b] - (double)hb[0]) * 0.01,
                            (double)(hb[16 * b + 1] - hb[16 * b]) * 0.01, (double)(hb[16 * b + 2] - hb[16 * b + 1]) * 0.01,
                            (double)(hb[16 * b + 3] - hb[16 * b + 2]) * 0.01);
            }
#endif
            continue;
        }
        if (mid && l == HL + 1) continue;
        const LayerI& y = h->lay[l];
        GemmArgs a{};
        a.A = in_of(l); a.lda = ldin_of(l);
        a.B = Wuse + y.off; a.ldb = y.ld; a.bias = h->W + y.off + (long)y.fi * y.ld;
        a.M = rows; a.N = y.fo; a.K = y.fi; a.act = y.act; a.Z = Z;
        if (l < NL - 1) {
            a.C = h->out[l]; a.ldc = y.ld;
#ifdef SMRT_STAMPS
            static unsigned long long* gdbg = nullptr;
            if (!gdbg) { (void)hipMalloc((void**)&gdbg, 64); (void)hipMemset(gdbg, 0, 64); }
            a.dbg = gdbg;
#endif
            { ProfScope ps(h, PK_FWD); launch_gemm(s, GEMM_NN, EPI_BIAS_ACT, a); } ++launches;
#ifdef SMRT_STAMPS
            if (l == 1 && rows >= 4096) {
                (void)hipStreamSynchronize(s);
                unsigned long long hb[4];
                (void)hipMemcpy(hb, gdbg, 32, hipMemcpyDeviceToHost);
                static int once = 0;
                if (once++ < 3) fprintf(stderr, "gemm l=1 main loop: %llu shader cycles in %.2f us -> %.3f GHz (M=%d N=%d K=%d)\n",
                                        hb[0], (double)hb[1] * 0.01, (double)hb[0] / ((double)hb[1] * 10.0), a.M, a.N, a.K);
                if (once <= 3) fprintf(stderr, "   whole workgroup incl. epilogue: %.2f us\n", (double)hb[3] * 0.01);
            }
#endif
            if (l == HL) {
                ProfScope ps(h, PK_REPARAM);
                launch_reparam(s, h->out[L], y.ld, eps_in, h->Zb, h->Eb, rup4(Z), rows, Z, h->cfg.seed, h->st,
                               row0_global, h->kl_p);
                ++launches;
            }
        } else {
            a.C = h->dpre[l]; a.ldc = y.ld; a.aux = X; a.ldaux = ldx;
            a.lo = h->cfg.bce_lo; a.hi = h->cfg.bce_hi; a.partials = h->bce_p;
            // the auto-encoder's loss is the mean over ALL B*D elements (autoencoder.py:260)
            a.inv_batch = h->ae ? inv_batch / (float)h->D : inv_batch;
            n_bce = gemm_grid_blocks(GEMM_NN, a);
            { ProfScope ps(h, PK_FWD_BCE); launch_gemm(s, GEMM_NN, h->ae ? EPI_MSE : EPI_BCE, a); } ++launches;
        }
    }
    // ---- backward through the activations
    for (int l = NL - 1; l >= 1 && !fused_step; --l) {
        if (mid && l == HL + 1) { ProfScope ps(h, PK_BWD_LATENT); launch_mid_bwd(s, ma); ++launches; continue; }
        if (mid && l == HL) continue;
        const LayerI& y = h->lay[l];
        GemmArgs a{};
        a.A = h->dpre[l]; a.lda = y.ld; a.M = rows; a.K = y.fo;
        a.B = Wuse + y.off; a.ldb = y.ld; a.N = y.fi;
        a.Z = Z; a.inv_batch = inv_batch;
        if (l == HL + 1) {
            a.C = h->dpre[L]; a.ldc = h->lay[L].ld; a.aux = h->out[L]; a.ldaux = h->lay[L].ld;
            a.aux2 = h->Eb; a.ldaux2 = rup4(Z); a.act = h->lay[L].act;
            { ProfScope ps(h, PK_BWD_LATENT); launch_gemm(s, GEMM_NT, EPI_LATENT, a); }
        } else {
            a.C = h->dpre[l - 1]; a.ldc = h->lay[l - 1].ld; a.aux = h->out[l - 1]; a.ldaux = h->lay[l - 1].ld;
            a.act = h->lay[l - 1].act;
            { ProfScope ps(h, PK_BWD); launch_gemm(s, GEMM_NT, EPI_ACTGRAD, a); }
        }
        ++launches;
    }
    // ---- small nets, small batches: dW + optimizer in one launch, cost only when somebody reads it
    if (fused_small) {
        DwOptArgs d{};
        const int tb = h->tiles16;
        d.tiles = h->tile_tab; d.total_tiles = tb; d.rows = rows;
        d.W = h->W; d.Wd = h->Wd; d.mask = h->mask; d.s0 = h->s0; d.s1 = h->s1; d.grad_out = o.grad_out;
        d.Wimg = h->has_drop ? nullptr : h->Wimg;
        d.l2_partials = h->l2_p; d.st = h->st; d.ticket = h->ticket;
        d.opt = h->cfg.optimizer; d.clip = h->cfg.clip; d.has_drop = h->has_drop; d.has_l2 = h->has_l2;
        d.lr = h->cfg.learning_rate; d.l2 = h->has_l2 ? h->cfg.l2_penalty : 0.f;
        d.gmin = h->cfg.gclip_min; d.gmax = h->cfg.gclip_max; d.keep = h->cfg.dropout_keep;
#ifdef SMRT_STAMPS
        static unsigned long long* dbgb = nullptr;
        if (!dbgb) { (void)hipMalloc((void**)&dbgb, 64 * 8); (void)hipMemset(dbgb, 0, 64 * 8); }
        d.dbg = dbgb;
#endif
        { ProfScope ps(h, PK_FUSED_DW_OPT); launch_fused_dw_opt(s, d); } ++launches;
#ifdef SMRT_STAMPS
        {
            (void)hipStreamSynchronize(s);
            unsigned long long hb[64];
            (void)hipMemcpy(hb, dbgb, 64 * 8, hipMemcpyDeviceToHost);
            static int onceb = 0;
            if (onceb++ == 3) for (int i = 21; i < 25; ++i) fprintf(stderr, "dw_opt stamp %d: +%.2f us\n", i, (double)(hb[i] - hb[i - 1]) * 0.01);
        }
#endif
        if (o.cost_slot) {
            FinalizeArgs f{};
            f.bce = h->bce_p; f.n_bce = n_bce; f.kl = h->kl_p; f.n_kl = n_kl; f.l2p = h->l2_p;
            f.n_l2 = dw_opt_blocks(tb); f.inv_batch = inv_batch; f.l2 = d.l2; f.cost_out = o.cost_slot;
            f.st = h->st; f.advance = 0;
            ring(f);
            { ProfScope ps(h, PK_FINAL); launch_finalize(s, f); } ++launches;
        }
        h->last_epoch_launches = launches;
        return 0;
    }
    // ---- all weight/bias gradients in one launch
    const bool small = rows <= SMALL_M_MAX;              // register-fed 32x32 tiles, one slab, bias row via a ones index
    const int splits = fused_big ? fused_slab_splits(rows)
                                 : (small ? 1 : std::min(choose_splits(h, rows), h->g_splits_cap));
    const bool big = dw_big(h, rows);
    if (fused_big) {
        DwSlabArgs d{};
        d.tiles = h->tile_tab; d.total_tiles = h->tiles16; d.rows = rows;
        d.blocks_per_split = (((rows + 15) / 16) + splits - 1) / splits;
        d.G = h->G; d.slab_stride = h->P_int;
        { ProfScope ps(h, PK_DW); launch_fused_dw_slab(s, d, splits); } ++launches;
    } else {
        // Small mini-batches, a full step, no dropout: the weight-gradient launch applies the update itself (small_gemm.hip:
        // small_dw_opt) -- one launch fewer per mini-batch, no gradient slab, no optimizer sweep.
        static const bool dwopt_env = []() { const char* e = std::getenv("SMRT_HIP_DW_OPT"); return e ? std::atoi(e) != 0 : true; }();
        if (small && mode == 0 && !h->has_drop && dwopt_env && h->n_rec32 <= h->l2_cap) {
            DwFusedArgs d{};
            d.tiles = h->dwrec; d.total_tiles = h->n_rec32; d.rows = rows;
            d.X = X; d.ldx = ldx;
            d.W = h->W; d.s0 = h->s0; d.s1 = h->s1; d.grad_out = o.grad_out; d.l2_partials = h->l2_p; d.st = h->st;
            d.opt = h->cfg.optimizer; d.clip = h->cfg.clip; d.has_l2 = h->has_l2 ? 1 : 0;
            d.lr = h->cfg.learning_rate; d.l2 = h->has_l2 ? h->cfg.l2_penalty : 0.f;
            d.gmin = h->cfg.gclip_min; d.gmax = h->cfg.gclip_max;
            d.advance = h->cfg.optimizer != 2;          // (as the sweep: under Adam every block reads the powers, finalize advances)
            d.want_l2 = (o.cost_slot || !d.advance) ? 1 : 0;
#ifdef SMRT_STAMPS
            static unsigned long long* ddbg = nullptr;
            if (!ddbg) { (void)hipMalloc((void**)&ddbg, 64 * 8); (void)hipMemset(ddbg, 0, 64 * 8); }
            d.dbg = ddbg;
#endif
            { ProfScope ps(h, PK_DW); launch_small_dw_opt(s, d); } ++launches;
#ifdef SMRT_STAMPS
            {
                (void)hipStreamSynchronize(s);
                unsigned long long hb[64];
                (void)hipMemcpy(hb, ddbg, 64 * 8, hipMemcpyDeviceToHost);
                static int once = 0;
                if (once++ == 4) for (int b = 0; b < 3; ++b)
                    fprintf(stderr, "dw_opt wg %s: start +%.2f us after wg 0 | set-up + slot requests %.2f | tile %.2f | update + stores issued %.2f | stores drained %.2f | l2 partial %.2f\n",
                            b == 0 ? "first" : (b == 1 ? "middle" : "last"), ((double)hb[16 * b] - (double)hb[0]) * 0.01,
                            (double)(hb[16 * b + 1] - hb[16 * b]) * 0.01, (double)(hb[16 * b + 2] - hb[16 * b + 1]) * 0.01,
                            (double)(hb[16 * b + 3] - hb[16 * b + 2]) * 0.01, (double)(hb[16 * b + 4] - hb[16 * b + 3]) * 0.01,
                            (double)(hb[16 * b + 5] - hb[16 * b + 4]) * 0.01);
            }
#endif
            h->img_valid = false;
            if (o.cost_slot || !d.advance) {
                FinalizeArgs f{};
                f.bce = h->bce_p; f.n_bce = n_bce; f.kl = h->kl_p; f.n_kl = n_kl; f.l2p = h->l2_p; f.n_l2 = d.total_tiles;
                f.inv_batch = h->ae ? inv_batch / (float)h->D : inv_batch; f.l2 = d.l2;
                f.cost_out = o.cost_slot ? o.cost_slot : h->cost_dev + 1;
                f.st = h->st; f.advance = d.advance ? 0 : 1;
                ring(f);
                { ProfScope ps(h, PK_FINAL); launch_finalize(s, f); } ++launches;
            }
            h->last_epoch_launches = launches;
            return 0;
        }
        DwArgs d{};
        const int t = small ? 32 : (big ? 128 : 64);
        int tb = 0;
        for (int l = 0; l < NL; ++l) {
            const LayerI& y = h->lay[l];
            DwLayer& q = d.L[l];
            q.A = in_of(l); q.lda = ldin_of(l); q.B = h->dpre[l]; q.ldb = y.ld;
            q.C = h->G + y.off; q.ldc = y.ld; q.fi = y.fi; q.fo = y.fo;
            q.tiles_n = (y.fo + t - 1) / t; q.tile_begin = tb;
            tb += ((y.fi + (small ? 1 : 0) + t - 1) / t) * q.tiles_n;
        }
        d.n_layers = NL; d.total_tiles = tb; d.rows = rows; d.splits = splits;
        d.k_per_split = (((rows + splits - 1) / splits) + 31) & ~31;
        d.slab_stride = h->P_int;
        if (!small) d.total_tiles = plan_dw_units(h, big, splits, &d, nullptr);      // (tile, K range) work units
        { ProfScope ps(h, PK_DW); if (small) launch_small_dw(s, d); else launch_dw(s, d, big); } ++launches;
    }
    if (mode == 1) {
        { ProfScope ps(h, PK_REDUCE); launch_reduce_slabs(s, h->G, splits, h->P_int, h->P_int, o.dp_out); } ++launches;
        FinalizeArgs f{};
        f.bce = h->bce_p; f.n_bce = n_bce; f.kl = h->kl_p; f.n_kl = n_kl; f.l2p = nullptr; f.n_l2 = 0;
        f.inv_batch = h->ae ? inv_batch / (float)h->D : inv_batch; f.l2 = 0.f; f.loss_sum_out = o.dp_out + h->P_int; f.st = h->st; f.advance = 0;
        { ProfScope ps(h, PK_FINAL); launch_finalize(s, f); } ++launches;
        h->last_epoch_launches = launches;
        return 0;
    }
    // ---- fused reduce + L2 + clip + update, then the scalar cost
    OptArgs p{};
    std::memcpy(p.seg, segs, sizeof(OptSeg) * NL); p.n_seg = NL; p.P = h->P_int;
    p.W = h->W; p.Wd = h->Wd; p.mask = h->mask; p.G = h->G; p.splits = splits; p.slab_stride = h->P_int;
    p.s0 = h->s0; p.s1 = h->s1; p.grad_out = o.grad_out; p.l2_partials = h->l2_p; p.st = h->st;
    p.opt = h->cfg.optimizer; p.clip = h->cfg.clip; p.has_drop = h->has_drop; p.has_l2 = h->has_l2;
    p.lr = h->cfg.learning_rate; p.l2 = h->has_l2 ? h->cfg.l2_penalty : 0.f;
    p.gmin = h->cfg.gclip_min; p.gmax = h->cfg.gclip_max; p.keep = h->cfg.dropout_keep;
    // Step counter / beta powers: advanced by the optimizer's first block, except under Adam (every block reads the powers
    // there).  The scalar cost is then only assembled when somebody reads it (the last batch of an epoch, base.py:222).
    p.advance = h->cfg.optimizer != 2;
    { ProfScope ps(h, PK_OPT); launch_optimizer(s, p, nullptr); } ++launches;
    h->img_valid = false;
    if (o.cost_slot || !p.advance) {
        FinalizeArgs f{};
        f.bce = h->bce_p; f.n_bce = n_bce; f.kl = h->kl_p; f.n_kl = n_kl; f.l2p = h->l2_p; f.n_l2 = h->n_l2;
        f.inv_batch = h->ae ? inv_batch / (float)h->D : inv_batch; f.l2 = p.l2; f.cost_out = o.cost_slot ? o.cost_slot : h->cost_dev + 1;
        f.st = h->st; f.advance = p.advance ? 0 : 1;
        ring(f);
        { ProfScope ps(h, PK_FINAL); launch_finalize(s, f); } ++launches;
    }
    h->last_epoch_launches = launches;
    return 0;
}

// ------------------------------------------------------------------ ABI
extern "C" {

const char* smrt_last_error(void) { return g_err.c_str(); }
const char* smrt_version(void) { return "smrt-hip 0.1 (gfx950)"; }

int smrt_vae_create(const smrt_vae_cfg* cfg, int device, smrt_vae** out) {
    if (!cfg || !out) return fail(-1, "smrt_vae_create: null argument");
    if (cfg->model != SMRT_MODEL_VAE && cfg->model != SMRT_MODEL_AE) return fail(-1, "smrt_vae_create: bad model id");
    if (cfg->n_features < 1 || (cfg->n_latent < 1 && cfg->model == SMRT_MODEL_VAE))
        return fail(-1, "smrt_vae_create: n_features and n_latent must be >= 1");
    if (cfg->n_hidden_layers < 1 || cfg->n_hidden_layers > SMRT_MAX_HIDDEN)
        return fail(-1, "smrt_vae_create: n_hidden_layers must be in 1..SMRT_MAX_HIDDEN");
    for (int i = 0; i < cfg->n_hidden_layers; ++i)
        if (cfg->hidden[i] < 1) return fail(-1, "smrt_vae_create: hidden widths must be >= 1");
    if (cfg->activation < 0 || cfg->activation > 4) return fail(-1, "smrt_vae_create: bad activation id");
    if (cfg->optimizer < 0 || cfg->optimizer > 6) return fail(-1, "smrt_vae_create: bad optimizer id");
    if (!(cfg->dropout_keep > 0.f && cfg->dropout_keep <= 1.f)) return fail(-1, "smrt_vae_create: dropout_keep must be in (0,1]");
    if (cfg->max_batch < 1) return fail(-1, "smrt_vae_create: max_batch must be >= 1");
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(-1, "smrt_vae_create: no such device");
    HIPC(hipSetDevice(device));
    smrt_vae* h = new smrt_vae();
    h->cfg = *cfg; h->device = device;
    // Leading dimensions (weights, and with them every activation / gradient buffer): a multiple of 4 floats for nets on the
    // LDS-resident path (their global buffers are blocked transposed, the flat one only feeds the image twin), a multiple of
    // 32 floats = 128 bytes otherwise.  With rows of 452 floats (the reference's 450-wide layer) every 128-byte row piece a
    // wave loads or stores straddles two cache lines; measured on the reference's own net at batch 128 against the same
    // net with 448 / 768-wide layers: weight-gradient launch 10.0 -> 7.5 us, every other launch 0.5-1 us.
    static const int ld_env = []() { const char* e = std::getenv("SMRT_HIP_LD_ALIGN"); return e ? std::atoi(e) : 32; }();
    build_layers(h, 4);
    plan_fused(h);
    if (!h->fused && ld_env > 4 && ld_env % 4 == 0) { build_layers(h, ld_env); plan_fused(h); }
    h->has_drop = cfg->dropout_keep < 1.0f;
    h->has_l2 = cfg->l2_penalty >= 0.f;
    const char* g = std::getenv("SMRT_HIP_GRAPH");
    h->use_graph = g ? std::atoi(g) : 1;
    auto bail = [&](int rc) { smrt_vae_destroy(h); return rc; };
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(-2, "hipStreamCreate failed"));
    h->stream = h->own_stream;
    int rc;
    h->n_l2 = optimizer_blocks(h->P_int);
    {
        int dw32 = 0;
        for (int l = 0; l < h->NL; ++l) dw32 += ((h->lay[l].fi + 1 + 31) / 32) * ((h->lay[l].fo + 31) / 32);
        h->l2_cap = std::max(std::max(h->n_l2, dw_opt_blocks(h->tiles16)), dw32);
        auto layout = [&](Arena& a) {
            a.off = 0;
            carve(a, &h->W, (size_t)h->P_int);
            carve(a, &h->s0, (size_t)h->P_int);
            carve(a, &h->s1, (size_t)h->P_int);
            carve(a, &h->Wimg, h->fused ? (size_t)h->fplan.img_total + 16 : 0);
            carve(a, &h->st, (size_t)1);
            carve(a, &h->ticket, (size_t)4);
            carve(a, &h->pcnt, (size_t)256);
            carve(a, &h->ep_ctr, (size_t)4);
            carve(a, &h->cost_dev, (size_t)4);
            carve(a, &h->l2_p, (size_t)h->l2_cap);     // (>= persist_owner_workgroups, >= the 32x32 tiles of small_dw_opt)
            carve(a, &h->stage_tab, h->fused ? (size_t)h->P_int / 4 : 0);
            carve(a, &h->gtmp, (size_t)h->P_int + 4);
            carve(a, &h->Wd, h->has_drop ? (size_t)h->P_int : 0);
            carve(a, &h->mask, h->has_drop ? (size_t)h->P_int : 0);
        };
        Arena dry;
        layout(dry);
        if ((rc = arena_alloc(h->pa, dry.off, h->stream))) return bail(rc);
        layout(h->pa);
        h->cost_cap = 2;
    }
    if (h->fused) {
        // where each 16-byte piece of the flat parameter buffer lands in the image twin (float offsets)
        std::vector<int> tab((size_t)h->P_int / 4);
        for (int l = 0; l < h->NL; ++l) {
            const LayerI& y = h->lay[l];
            const FusedLayer& q = h->flay[l];
            for (int r = 0; r <= y.fi; ++r)
                for (int c4 = 0; c4 < y.ld; c4 += 4)
                    tab[(size_t)(y.off + (long)r * y.ld + c4) / 4] =
                        (r < y.fi) ? q.lds_w + (((r >> 2) * q.nn + c4) << 2) + (r & 3) : q.lds_b + 4 * c4;
        }
        if (hipMemcpy(h->stage_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(-2, "stage table upload failed"));
    }
    if ((rc = smrt_vae_reset_opt_state(h))) return bail(rc);
    if ((rc = ensure_train(h, cfg->max_batch))) return bail(rc);
    *out = h;
    return 0;
}

void smrt_vae_destroy(smrt_vae* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream || h->own_stream) (void)hipStreamSynchronize(h->stream);
    if (h->pa.base) (void)hipFree(h->pa.base);
    if (h->ta.base) (void)hipFree(h->ta.base);
    float* ibufs[] = {h->ibuf[0], h->ibuf[1], h->izb};
    for (float* b : ibufs) if (b) (void)hipFree(b);
    if (h->cost_pin) (void)hipHostFree(h->cost_pin);
    if (h->time_pin) (void)hipHostFree(h->time_pin);
    if (h->xt_all) (void)hipFree(h->xt_all);
    if (h->err_pin) (void)hipHostFree(h->err_pin);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof_pool) (void)hipEventDestroy(e);
    drop_graph(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int64_t smrt_vae_param_count(const smrt_vae* h) { return h ? h->P_abi : -1; }
int64_t smrt_vae_dp_count(const smrt_vae* h) { return h ? h->P_int + 4 : -1; }
int64_t smrt_vae_step_count(const smrt_vae* h) { return h ? h->steps : -1; }
int smrt_vae_last_epoch_launches(const smrt_vae* h) { return h ? h->last_epoch_launches : -1; }

int32_t smrt_persistent_budget(int32_t device) { return persist_budget(device); }
int32_t smrt_vae_persistent_form(const smrt_vae* h) {
    if (!h || !h->persist_ok || !h->pkernel) return 0;
    return h->pkernel == persist_pick_kernel(nullptr, 0) ? 1 : 2;
}
int32_t smrt_vae_persistent_workgroups(const smrt_vae* h, int32_t batch_rows) {
    if (!h) return -1;
    if (!h->persist_ok || batch_rows > PERSIST_ROWS_MAX || batch_rows < 1) return 0;
    const int owners = persist_owner_workgroups(h->tiles16);
    if (owners > persist_max_owner_workgroups()) return 0;
    return (batch_rows + 15) / 16 + owners;
}
int smrt_vae_set_persistent(smrt_vae* h, int32_t mode) {
    if (!h) return fail(-1, "smrt_vae_set_persistent: null handle");
    if (mode < -1 || mode > 1) return fail(-1, "smrt_vae_set_persistent: mode must be -1 (automatic), 0 (never) or 1 (planned by the caller)");
    h->persist_mode = mode;
    return 0;
}

int smrt_vae_set_params(smrt_vae* h, const float* host_flat, int64_t n) {
    if (!h || !host_flat) return fail(-1, "smrt_vae_set_params: null argument");
    if (n != h->P_abi) return fail(-1, "smrt_vae_set_params: expected " + std::to_string(h->P_abi) + " floats");
    HIPC(hipSetDevice(h->device));
    std::vector<float> in(h->P_int, 0.f);
    convert(h, const_cast<float*>(host_flat), in.data(), true);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipMemcpy(h->W, in.data(), sizeof(float) * h->P_int, hipMemcpyHostToDevice));
    h->img_valid = false;
    return 0;
}

// Bring the LDS-image twin up to date before fused steps run (a cached epoch graph holds no rebuild node).
static void ensure_image(smrt_vae* h) {
    if (h->fused && !h->has_drop && !h->img_valid) {
        launch_build_image(h->stream, h->W, h->stage_tab, h->Wimg, (int)(h->P_int / 4));
        h->img_valid = true;
    }
}

static int download_abi(smrt_vae* h, const float* dev, float* host_flat, int64_t n, const char* who) {
    if (!h || !host_flat) return fail(-1, std::string(who) + ": null argument");
    if (n != h->P_abi) return fail(-1, std::string(who) + ": expected " + std::to_string(h->P_abi) + " floats");
    HIPC(hipSetDevice(h->device));
    std::vector<float> in(h->P_int);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipMemcpy(in.data(), dev, sizeof(float) * h->P_int, hipMemcpyDeviceToHost));
    convert(h, host_flat, in.data(), false);
    return 0;
}

int smrt_vae_get_params(smrt_vae* h, float* host_flat, int64_t n) {
    return download_abi(h, h ? h->W : nullptr, host_flat, n, "smrt_vae_get_params");
}

int smrt_vae_get_opt_state(smrt_vae* h, int slot, float* host_flat, int64_t n) {
    if (slot != 0 && slot != 1) return fail(-1, "smrt_vae_get_opt_state: slot must be 0 or 1");
    return download_abi(h, h ? (slot ? h->s1 : h->s0) : nullptr, host_flat, n, "smrt_vae_get_opt_state");
}

int smrt_vae_reset_opt_state(smrt_vae* h) {
    if (!h) return fail(-1, "smrt_vae_reset_opt_state: null handle");
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamSynchronize(h->stream));
    float v0 = 0.f;
    if (h->cfg.optimizer == SMRT_OPT_RMSPROP) v0 = 1.0f;                                  // rms slot starts at one
    if (h->cfg.optimizer == SMRT_OPT_ADAGRAD || h->cfg.optimizer == SMRT_OPT_PROXIMAL_ADAGRAD) v0 = 0.1f;
    std::vector<float> a(h->P_int, v0);
    HIPC(hipMemcpy(h->s0, a.data(), sizeof(float) * h->P_int, hipMemcpyHostToDevice));
    HIPC(hipMemset(h->s1, 0, sizeof(float) * h->P_int));
    DevState st{0.9f, 0.999f, 0u, 0u};
    HIPC(hipMemcpy(h->st, &st, sizeof(st), hipMemcpyHostToDevice));
    h->steps = 0;
    return 0;
}

int smrt_vae_set_dropout_masks(smrt_vae* h, const float* host_flat, int64_t n) {
    if (!h) return fail(-1, "smrt_vae_set_dropout_masks: null handle");
    drop_graph(h);
    if (!host_flat) { h->masks_injected = false; return 0; }
    if (!h->has_drop) return fail(-1, "smrt_vae_set_dropout_masks: handle was created with dropout_keep == 1");
    if (n != h->P_abi) return fail(-1, "smrt_vae_set_dropout_masks: expected " + std::to_string(h->P_abi) + " floats");
    HIPC(hipSetDevice(h->device));
    std::vector<float> in(h->P_int, 0.f);
    convert(h, const_cast<float*>(host_flat), in.data(), true);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipMemcpy(h->mask, in.data(), sizeof(float) * h->P_int, hipMemcpyHostToDevice));
    h->masks_injected = true;
    return 0;
}

// cost_dev is two floats for the handle's lifetime (slot 0: the cost that is read back, slot 1: scratch for
// steps whose cost nobody reads) so epoch graphs never hold a stale pointer; only the pinned mirror grows.
static int ensure_cost(smrt_vae* h, long n) {
    if (n > h->cost_pin_cap) {
        if (h->cost_pin) (void)hipHostFree(h->cost_pin);
        if (h->err_pin) (void)hipHostFree(h->err_pin);
        if (h->time_pin) (void)hipHostFree(h->time_pin);
        h->cost_pin = nullptr; h->err_pin = nullptr; h->time_pin = nullptr; h->cost_pin_cap = 0;
        drop_graph(h);                                   // (a cached epoch graph writes the ring through the old pointers)
        HIPC(hipHostMalloc((void**)&h->cost_pin, sizeof(float) * n, hipHostMallocDefault));
        HIPC(hipHostMalloc((void**)&h->time_pin, sizeof(unsigned long long) * (n + 1), hipHostMallocDefault));
        HIPC(hipHostMalloc((void**)&h->err_pin, sizeof(unsigned) * n, hipHostMallocDefault));
        h->cost_pin_cap = n;
    }
    return 0;
}

int smrt_vae_train_step(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, const float* eps_dev,
                        float* cost_host, float* grads_host) {
    if (!h) return fail(-1, "smrt_vae_train_step: null handle");
    if (rows < 1) return fail(-1, "smrt_vae_train_step: rows must be >= 1");
    TRY(check_mat(X_dev, ldx, h->D, "smrt_vae_train_step: X"));
    HIPC(hipSetDevice(h->device));
    TRY(ensure_train(h, rows));
    TRY(ensure_cost(h, 1));
    StepOut o{h->cost_dev, grads_host ? h->gtmp : nullptr, nullptr};
    ensure_image(h);
    TRY(enqueue_step(h, X_dev, ldx, rows, 1.0f / (float)rows, 0, eps_dev, 0, o));
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(h->cost_pin, h->cost_dev, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    h->steps += 1;
    if (cost_host) *cost_host = h->cost_pin[0];
    if (grads_host) {
        std::vector<float> in(h->P_int);
        HIPC(hipMemcpy(in.data(), h->gtmp, sizeof(float) * h->P_int, hipMemcpyDeviceToHost));
        convert(h, grads_host, in.data(), false);
    }
    return 0;
}

int smrt_vae_train_epochs(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n_rows, int32_t batch_size,
                          int32_t n_epochs, int32_t early_stopping, float min_change,
                          float* epoch_cost_host, float* epoch_ms_host, int32_t* epochs_done) {
    if (!h) return fail(-1, "smrt_vae_train_epochs: null handle");
    if (n_rows < 1 || batch_size < 1 || n_epochs < 0) return fail(-1, "smrt_vae_train_epochs: bad sizes");
    TRY(check_mat(X_dev, ldx, h->D, "smrt_vae_train_epochs: X"));
    HIPC(hipSetDevice(h->device));
    const int bmax = (int)std::min<int64_t>(batch_size, n_rows);
    TRY(ensure_train(h, bmax));
    const long nb = (n_rows + batch_size - 1) / batch_size;
    TRY(ensure_cost(h, std::max<long>(n_epochs, 1) + 1));
    // a bounded ring of event pairs: at most EV_RING epochs are in flight before the host waits and reads their times
    constexpr int EV_RING = 64;
    while ((long)h->events.size() < 2L * std::min(std::max(n_epochs, 1), EV_RING)) {
        hipEvent_t e; HIPC(hipEventCreate(&e)); h->events.push_back(e);
    }
    std::vector<float> ep_ms((size_t)std::max(n_epochs, 1), 0.f);
    int harvested = 0;
    bool ring = false;                              // the epoch ring instead of an event pair + a copy per epoch (set below)
    auto harvest = [&](int upto) {                  // epochs [harvested, upto) have completed
        for (; harvested < upto; ++harvested) {
            const int k = harvested % EV_RING;
            if (ring) ep_ms[harvested] = (float)((double)(h->time_pin[harvested + 1] - h->time_pin[harvested]) * 1e-5);   // 100 MHz
            else (void)hipEventElapsedTime(&ep_ms[harvested], h->events[2 * k], h->events[2 * k + 1]);
        }
    };
    if (epochs_done) *epochs_done = 0;
    if (n_epochs == 0) return 0;
    ensure_image(h);
    // The persistent whole-epoch kernel when the net and the schedule qualify and its workgroups fit the device's budget of
    // co-resident persistent workgroups (released when this call returns, whichever way).  A caller that has planned the
    // budget over several handles itself (smrt_vae_set_persistent(h, 1): fit_group, in class order, so that which classes
    // take this path does not depend on thread timing) is always granted.
    int PG = persist_grid(h, bmax);
    if (PG && persist_owner_workgroups(h->tiles16) > persist_max_owner_workgroups()) PG = 0;
    const int PW = PG ? PG + persist_owner_workgroups(h->tiles16) : 0;          // workgroups of the launch
    std::atomic<int>& held = g_persist_wgs[std::min(std::max(h->device, 0), SMRT_MAX_DEVICES - 1)];
    if (PG) {
        const int prev = held.fetch_add(PW);
        if (h->persist_mode != 1 && prev + PW > persist_budget(h->device)) { held.fetch_sub(PW); PG = 0; }
    }
    struct Release { std::atomic<int>* c; int n; void drop() { if (n) c->fetch_sub(n); n = 0; } ~Release() { drop(); } } release{&held, PG ? PW : 0};
    // test hooks: SMRT_HIP_PERSIST_FAULT_EPOCH = e raises the error word in front of epoch e's launch (what a peer's time-out
    // leaves behind); SMRT_HIP_PERSIST_NO_FALLBACK = 1 reports it (-6) instead of re-running on the two-launch step
    const char* fe_env = std::getenv("SMRT_HIP_PERSIST_FAULT_EPOCH");
    const int fault_epoch = (fe_env && *fe_env) ? std::atoi(fe_env) : -1;
    const char* nf_env = std::getenv("SMRT_HIP_PERSIST_NO_FALLBACK");
    const bool no_fallback = nf_env && std::atoi(nf_env) != 0;
    if (PG) HIPC(hipMemsetAsync(h->pcnt, 0, 256 * sizeof(unsigned), h->stream));   // producer flags + the sticky error word: once per call
    // X does not change during the call: its blocked-transposed form (the first layer's weight-gradient operand) is built once,
    // in front of the epochs, instead of being published by the compute workgroups every step -- when it fits the cap
    // (SMRT_HIP_XT_ALL_MB, default 1024; 0 keeps the per-step form)
    h->xt_all_on = false;
    if (PG) {
        const char* xe = std::getenv("SMRT_HIP_XT_ALL_MB");
        const size_t cap_mb = xe ? (size_t)std::max(0, std::atoi(xe)) : 1024;
        const size_t need = (size_t)nb * (size_t)PG * (size_t)rup16(h->D) * 16;           // floats
        if (need * sizeof(float) <= (cap_mb << 20) && need * sizeof(float) < (size_t(1) << 32)) {
            if (need > h->xt_all_cap) {
                HIPC(hipStreamSynchronize(h->stream));
                h->xt_all_cap = 0;
                TRY(dalloc(&h->xt_all, need, h->stream));
                h->xt_all_cap = need;
            }
            launch_xt_all(h->stream, X_dev, ldx, n_rows, batch_size, PG, h->D, h->xt_all);
            h->xt_all_on = true;
        }
    }
    int cur_epoch = 0;
    // Without the persistent kernel an epoch is a graph of a few launches (two mini-batches of the reference's own 500-row
    // example: 13 nodes, 115 us): an event pair and a 4-byte copy per epoch around it were 12 of its 134 us.  The launch that
    // closes the epoch writes cost and time into pinned memory itself (FinalizeArgs::ring_*); SMRT_HIP_EPOCH_RING=0 keeps
    // the events and copies.  A persistent epoch (one launch + the cost launch) gets rid of five more stream operations
    // that way: the flag clear in front of every launch, two event records, the cost copy and the error-word copy.
    const char* ring_e = std::getenv("SMRT_HIP_EPOCH_RING");          // (read per call: tests switch it)
    ring = !(ring_e && std::atoi(ring_e) == 0);
    struct RingOff { smrt_vae* h; ~RingOff() { h->ring_on = false; } } ring_off{h};
    h->ring_on = ring;
    if (ring) launch_ring_start(h->stream, h->time_pin, h->ep_ctr, 0u);

    // Enqueue one epoch (every contiguous batch, ragged tail last: sklearn gen_batches, base.py:198-199).
    // The cost of the LAST batch lands in `slot` (base.py:222); other steps write a scratch slot.
    auto enqueue_epoch = [&](float* slot) -> int {
        int launches = 0;
        if (PG) {
            launches = enqueue_persist_epoch(h, X_dev, ldx, n_rows, batch_size, PG, slot, cur_epoch == fault_epoch);
            if (launches < 0) return fail(-2, "persistent epoch: counter reset failed");
            h->last_epoch_launches = launches;
            return 0;
        }
        for (long b = 0; b < nb; ++b) {
            const long r0 = b * batch_size;
            const int rows = (int)std::min<long>(batch_size, n_rows - r0);
            StepOut o{(b == nb - 1) ? slot : nullptr, nullptr, nullptr};
            TRY(enqueue_step(h, X_dev + r0 * ldx, ldx, rows, 1.0f / (float)rows, 0, nullptr, 0, o));
            launches += h->last_epoch_launches;
        }
        h->last_epoch_launches = launches;
        return 0;
    };

    // The epoch graph replays the identical epoch: every step-dependent scalar (Philox step, Adam powers)
    // lives in device memory, so only the cost slot differs -> the graph writes slot 0 and a copy node moves it.
    bool graphed = false;
    if (h->use_graph && !PG && !h->prof && n_epochs > 1 && h->stream == h->own_stream) {
        const bool hit = h->gexec && h->g_X == X_dev && h->g_ldx == ldx && h->g_n == n_rows && h->g_batch == batch_size &&
                         h->g_ring == ring;
        if (hit) { graphed = true; h->last_epoch_launches = h->g_launches; }
        else {
            drop_graph(h);
            // Build the epoch graph node by node (vae_kernels.hpp: GraphBuild) -- no stream is ever put into capture.
            if (hipGraphCreate(&h->graph, 0) == hipSuccess) {
                g_build = GraphBuild{};
                g_build.graph = h->graph;
                const int rc = enqueue_epoch(h->cost_dev);
                const hipError_t e = g_build.err;
                const long nodes = g_build.nodes;
                g_build = GraphBuild{};
                if (rc == 0 && e == hipSuccess && nodes > 0 &&
                    hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0) == hipSuccess) {
                    graphed = true;
                    h->g_X = X_dev; h->g_ldx = ldx; h->g_n = n_rows; h->g_batch = batch_size;
                    h->g_launches = h->last_epoch_launches; h->g_ring = ring;
                } else {
                    (void)hipGetLastError(); drop_graph(h);
                    if (rc) return rc;
                }
            } else (void)hipGetLastError();
        }
    }
    int done = 0; float last_cost = 0.f; bool have_last = false; int rc = 0;
    auto run_epoch = [&](int ep) -> int {
        cur_epoch = ep;
        if (ring) {
            if (graphed) HIPC(hipGraphLaunch(h->gexec, h->stream));
            else TRY(enqueue_epoch(h->cost_dev));
            return 0;
        }
        if (ep - harvested >= EV_RING) {            // the ring of events is full: wait for what is queued, read its times
            HIPC(hipStreamSynchronize(h->stream));
            harvest(ep);
        }
        const int k = ep % EV_RING;
        HIPC(hipEventRecord(h->events[2 * k], h->stream));
        if (graphed) HIPC(hipGraphLaunch(h->gexec, h->stream));
        else TRY(enqueue_epoch(h->cost_dev));
        HIPC(hipEventRecord(h->events[2 * k + 1], h->stream));
        HIPC(hipMemcpyAsync(h->cost_pin + ep, h->cost_dev, sizeof(float), hipMemcpyDeviceToHost, h->stream));
        if (PG) HIPC(hipMemcpyAsync(h->err_pin + ep, h->pcnt + 192, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        return 0;
    };
    // A persistent launch whose workgroups gave up waiting for a peer (2 s: the grid was not co-resident -- another process
    // on the GPU, a CU mask) leaves the flat W / slot buffers and the step scalars exactly as they were when ITS epoch
    // began: owners write them back only at the end of a complete launch, and every later launch of this call leaves in
    // its prologue (sticky error word).  So the first epoch whose error word is set, and all after it, are simply run
    // again on the two-launch step; only the image twin has to be rebuilt from W.
    auto first_failed = [&](int from, int upto) -> int {
        if (!PG) return -1;
        for (int ep = from; ep < upto; ++ep) if (h->err_pin[ep] != 0u) return ep;
        return -1;
    };
    auto leave_persistent = [&](int from_epoch) -> int {
        PG = 0; release.drop();
        h->img_valid = false;
        ensure_image(h);
        HIPC(hipMemsetAsync(h->pcnt + 192, 0, 64 * sizeof(unsigned), h->stream));
        if (ring) launch_ring_start(h->stream, h->time_pin, h->ep_ctr, (unsigned)from_epoch);   // the ring goes on at that epoch
        return 0;
    };
    const char* give_up = "smrt_vae_train_epochs: the persistent training kernel gave up waiting for a peer workgroup (2 s) and "
                          "SMRT_HIP_PERSIST_NO_FALLBACK is set; weights, slots and step count are those at the start of the failed epoch";
    if (!early_stopping) {
        int ep0 = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (int ep = ep0; ep < n_epochs && rc == 0; ++ep) rc = run_epoch(ep);
            if (rc) return rc;
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return fail(-2, std::string("train_epochs sync: ") + hipGetErrorString(e));
            const int bad = first_failed(ep0, n_epochs);
            if (bad < 0) break;
            if (no_fallback) { h->steps += (int64_t)bad * nb; h->img_valid = false; return fail(-6, give_up); }
            harvest(bad);
            harvested = std::min(harvested, bad);          // the epochs from `bad` on are timed again
            TRY(leave_persistent(bad));
            ep0 = bad;
        }
        done = n_epochs;
    } else {
        for (int ep = 0; ep < n_epochs; ++ep) {
            rc = run_epoch(ep);
            if (rc) return rc;
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return fail(-2, std::string("train_epochs sync: ") + hipGetErrorString(e));
            if (first_failed(ep, ep + 1) >= 0) {
                if (no_fallback) { h->steps += (int64_t)ep * nb; h->img_valid = false; return fail(-6, give_up); }
                harvest(ep);
                TRY(leave_persistent(ep));
                rc = run_epoch(ep);
                if (rc) return rc;
                e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) return fail(-2, std::string("train_epochs sync: ") + hipGetErrorString(e));
            }
            done = ep + 1;
            const float c = h->cost_pin[ep];
            if (!have_last) { last_cost = c; have_last = true; }
            else if (std::fabs(last_cost - c) <= min_change) break;       // base.py:234-238
            else last_cost = c;
        }
    }
    harvest(done);
    for (int ep = 0; ep < done; ++ep) {
        if (epoch_cost_host) epoch_cost_host[ep] = h->cost_pin[ep];
        if (epoch_ms_host) epoch_ms_host[ep] = ep_ms[ep];
    }
    h->steps += (int64_t)done * nb;
    if (epochs_done) *epochs_done = done;
    return 0;
}

// ---- several handles trained side by side with grouped launches (vae_kernels.hpp: MultiSlot / Recorder) ------------
// Returns 1 ("declined", nothing done) when grouping does not apply: a handle on the LDS-resident paths (its steps are one
// or two launches already), mini-batches beyond the small-batch kernels, handles on different devices.
int smrt_vae_train_epochs_multi(smrt_vae** hs, int32_t n_handles, const float* const* X_dev, const int64_t* ldx,
                                const int64_t* n_rows, int32_t batch_size, int32_t n_epochs,
                                float* epoch_cost_host, float* epoch_ms_host) {
    if (!hs || !X_dev || !ldx || !n_rows || n_handles < 1) return fail(-1, "smrt_vae_train_epochs_multi: null argument");
    if (batch_size < 1 || n_epochs < 1) return fail(-1, "smrt_vae_train_epochs_multi: bad sizes");
    for (int k = 0; k < n_handles; ++k) {
        smrt_vae* h = hs[k];
        if (!h) return fail(-1, "smrt_vae_train_epochs_multi: null handle");
        if (n_rows[k] < 1) return fail(-1, "smrt_vae_train_epochs_multi: bad sizes");
        TRY(check_mat(X_dev[k], ldx[k], h->D, "smrt_vae_train_epochs_multi: X"));
        if (h->device != hs[0]->device || h->fused || h->prof || std::min<int64_t>(batch_size, n_rows[k]) > SMALL_M_MAX ||
            h->stream != h->own_stream)             // (a lent stream orders this handle against its lender's work, not ours)
            return 1;
        for (int j = 0; j < k; ++j) if (hs[j] == h) return fail(-1, "smrt_vae_train_epochs_multi: a handle appears twice");
    }
    if (n_handles < 2) return 1;
    HIPC(hipSetDevice(hs[0]->device));
    hipStream_t stream = hs[0]->stream;
    for (int k = 0; k < n_handles; ++k) {
        HIPC(hipStreamSynchronize(hs[k]->stream));
        TRY(ensure_train(hs[k], (int)std::min<int64_t>(batch_size, n_rows[k])));
        TRY(ensure_cost(hs[k], n_epochs + 1));
    }
    // every handle reports its epochs' costs and end times through its own epoch ring (FinalizeArgs::ring_*): no copy per
    // handle and epoch, no event pairs (SMRT_HIP_EPOCH_RING=0 keeps them)
    const char* ring_e = std::getenv("SMRT_HIP_EPOCH_RING");
    const bool ring = !(ring_e && std::atoi(ring_e) == 0);
    struct RingOff { smrt_vae** hs; int n; ~RingOff() { for (int k = 0; k < n; ++k) hs[k]->ring_on = false; } } ring_off{hs, n_handles};
    for (int k = 0; k < n_handles; ++k) {
        hs[k]->ring_on = ring;
        if (ring) launch_ring_start(stream, hs[k]->time_pin, hs[k]->ep_ctr, 0u);
    }
    // 1. record every handle's epoch, step by step (nothing is launched)
    std::vector<std::vector<std::vector<RecordedLaunch>>> rec(n_handles);       // [handle][step][launch]
    size_t max_steps = 0;
    for (int k = 0; k < n_handles; ++k) {
        smrt_vae* h = hs[k];
        const long nb = (n_rows[k] + batch_size - 1) / batch_size;
        rec[k].resize(nb);
        for (long b = 0; b < nb; ++b) {
            const long r0 = b * batch_size;
            const int rows = (int)std::min<long>(batch_size, n_rows[k] - r0);
            StepOut o{(b == nb - 1) ? h->cost_dev : nullptr, nullptr, nullptr};
            g_record.out = &rec[k][b];
            g_record.group = n_handles;                  // (tile shapes are picked for the grouped grid)
            const int rc = enqueue_step(h, X_dev[k] + r0 * ldx[k], ldx[k], rows, 1.0f / (float)rows, 0, nullptr, 0, o);
            g_record.out = nullptr; g_record.group = 1;
            if (rc) return rc;
        }
        max_steps = std::max(max_steps, (size_t)nb);
    }
    // 2. merge position by position: launches of one kernel at the same (step, position) become one grouped launch
    struct Merged { const MultiInfo* info; dim3 grid, block; unsigned shmem; size_t table_off; std::function<void(hipStream_t)> plain; };
    std::vector<Merged> plan;
    std::vector<char> table;
    for (size_t sidx = 0; sidx < max_steps; ++sidx) {
        size_t max_len = 0;
        for (int k = 0; k < n_handles; ++k) if (sidx < rec[k].size()) max_len = std::max(max_len, rec[k][sidx].size());
        for (size_t j = 0; j < max_len; ++j) {
            std::vector<RecordedLaunch*> here;
            for (int k = 0; k < n_handles; ++k)
                if (sidx < rec[k].size() && j < rec[k][sidx].size()) here.push_back(&rec[k][sidx][j]);
            std::vector<bool> done(here.size(), false);
            for (size_t i = 0; i < here.size(); ++i) {
                if (done[i]) continue;
                std::vector<RecordedLaunch*> grp{here[i]};
                done[i] = true;
                const MultiInfo* info = here[i]->args.empty() ? nullptr : smrt_multi_of(here[i]->kernel);
                if (info)
                    for (size_t i2 = i + 1; i2 < here.size(); ++i2)
                        if (!done[i2] && here[i2]->kernel == here[i]->kernel && !here[i2]->args.empty() &&
                            here[i2]->shmem == here[i]->shmem && here[i2]->block.x == here[i]->block.x &&
                            here[i2]->grid.z == 1 && here[i]->grid.z == 1) { grp.push_back(here[i2]); done[i2] = true; }
                if (!info || grp.size() < 2) {
                    for (RecordedLaunch* r : grp) plan.push_back(Merged{nullptr, r->grid, r->block, r->shmem, 0, r->plain});
                    continue;
                }
                Merged m{info, dim3(0, 0, (unsigned)grp.size()), grp[0]->block, grp[0]->shmem, table.size(), nullptr};
                for (RecordedLaunch* r : grp) {
                    m.grid.x = std::max(m.grid.x, r->grid.x); m.grid.y = std::max(m.grid.y, r->grid.y);
                    const size_t at = table.size();
                    table.resize(at + info->slot_size, 0);
                    std::memcpy(table.data() + at, r->args.data(), info->arg_size);
                    const unsigned g2[2] = {r->grid.x, r->grid.y};
                    std::memcpy(table.data() + at + info->gx_off, g2, sizeof(g2));
                }
                plan.push_back(std::move(m));
            }
        }
    }
    // 3. the tables on the device, the epoch as ONE graph, n_epochs replays
    char* table_dev = nullptr;
    if (!table.empty()) {
        HIPC(hipMalloc((void**)&table_dev, table.size()));
        if (hipMemcpy(table_dev, table.data(), table.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(table_dev);
            return fail(-2, "smrt_vae_train_epochs_multi: table upload failed");
        }
    }
    auto issue_all = [&](hipStream_t s) {
        for (Merged& m : plan) {
            if (!m.info) { m.plain(s); continue; }
            std::tuple<const void*> arg((const void*)(table_dev + m.table_off));
            smrt_issue((void (*)(const void*))m.info->multi, m.grid, m.block, m.shmem, s, arg);
        }
    };
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
    bool graphed = false;
    if (hs[0]->use_graph && n_epochs > 1 && hipGraphCreate(&graph, 0) == hipSuccess) {
        g_build = GraphBuild{};
        g_build.graph = graph;
        issue_all(stream);
        const hipError_t e = g_build.err;
        const long nodes = g_build.nodes;
        g_build = GraphBuild{};
        graphed = e == hipSuccess && nodes > 0 && hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (!graphed) (void)hipGetLastError();
    }
    std::vector<hipEvent_t> ev(ring ? 0 : 2 * (size_t)n_epochs, nullptr);
    int rc = 0;
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) rc = fail(-2, "hipEventCreate failed");
    for (int ep = 0; ep < n_epochs && rc == 0; ++ep) {
        if (ring) {
            if (graphed) { if (hipGraphLaunch(gexec, stream) != hipSuccess) rc = fail(-2, "grouped epoch graph launch failed"); }
            else issue_all(stream);
            continue;
        }
        (void)hipEventRecord(ev[2 * ep], stream);
        if (graphed) { if (hipGraphLaunch(gexec, stream) != hipSuccess) rc = fail(-2, "grouped epoch graph launch failed"); }
        else issue_all(stream);
        (void)hipEventRecord(ev[2 * ep + 1], stream);
        for (int k = 0; k < n_handles && rc == 0; ++k)
            if (hipMemcpyAsync(hs[k]->cost_pin + ep, hs[k]->cost_dev, sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess)
                rc = fail(-2, "cost read-back failed");
    }
    const hipError_t es = hipStreamSynchronize(stream);
    if (rc == 0 && es != hipSuccess) rc = fail(-2, std::string("train_epochs_multi sync: ") + hipGetErrorString(es));
    if (rc == 0) {
        for (int ep = 0; ep < n_epochs; ++ep) {
            float ms = 0.f;
            if (ring) {                                  // an epoch ends when its last handle's closing launch has stamped
                unsigned long long t0 = 0, t1 = 0;
                for (int k = 0; k < n_handles; ++k) {
                    t0 = std::max(t0, hs[k]->time_pin[ep]); t1 = std::max(t1, hs[k]->time_pin[ep + 1]);
                }
                ms = (float)((double)(t1 - t0) * 1e-5);   // 100 MHz
            } else (void)hipEventElapsedTime(&ms, ev[2 * ep], ev[2 * ep + 1]);
            if (epoch_ms_host) epoch_ms_host[ep] = ms;
            for (int k = 0; k < n_handles; ++k)
                if (epoch_cost_host) epoch_cost_host[(size_t)k * n_epochs + ep] = hs[k]->cost_pin[ep];
        }
        for (int k = 0; k < n_handles; ++k) {
            hs[k]->steps += (int64_t)n_epochs * ((n_rows[k] + batch_size - 1) / batch_size);
            hs[k]->last_epoch_launches = (int)plan.size();
        }
    }
    if (rc != 0) {
        // launches may have run: the device-side step counters are the truth (an epoch advances them once per mini-batch)
        (void)hipStreamSynchronize(stream);
        for (int k = 0; k < n_handles; ++k) {
            DevState st{};
            if (hipMemcpy(&st, hs[k]->st, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) hs[k]->steps = (int64_t)st.step;
            hs[k]->img_valid = false;
        }
        (void)hipGetLastError();
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    if (table_dev) (void)hipFree(table_dev);
    return rc;
}

int smrt_vae_set_stream(smrt_vae* h, void* hip_stream, int32_t own) {
    if (!h) return fail(-1, "smrt_vae_set_stream: null handle");
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamSynchronize(h->stream));
    drop_graph(h);
    h->stream = own ? h->own_stream : (hipStream_t)hip_stream;
    return 0;
}

static int grad_step_impl(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, int64_t global_rows,
                          int64_t row0_global, const float* eps_dev, float* grads_dev, bool sync, const char* who) {
    if (!h || !grads_dev) return fail(-1, std::string(who) + ": null argument");
    if (rows < 0 || global_rows < rows || global_rows < 1) return fail(-1, std::string(who) + ": bad row counts");
    if (h->has_drop) return fail(-1, std::string(who) + ": dropout < 1 is not supported on the data-parallel path");
    if (((uintptr_t)grads_dev & 15) != 0) return fail(-1, std::string(who) + ": grads_dev must be 16-byte aligned");
    HIPC(hipSetDevice(h->device));
    if (rows == 0) {
        // a rank whose stripe of a ragged tail batch is empty contributes zeros to the all-reduce
        HIPC(hipMemsetAsync(grads_dev, 0, sizeof(float) * (size_t)(h->P_int + 4), h->stream));
    } else {
        TRY(check_mat(X_dev, ldx, h->D, who));
        TRY(ensure_train(h, rows));
        StepOut o{nullptr, nullptr, grads_dev};
        TRY(enqueue_step(h, X_dev, ldx, rows, 1.0f / (float)global_rows, row0_global, eps_dev, 1, o));
        HIPC(hipGetLastError());
    }
    if (sync) HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

int smrt_vae_grad_step(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, int64_t global_rows,
                       int64_t row0_global, const float* eps_dev, float* grads_dev) {
    return grad_step_impl(h, X_dev, ldx, rows, global_rows, row0_global, eps_dev, grads_dev, true, "smrt_vae_grad_step");
}

int smrt_vae_grad_step_async(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, int64_t global_rows,
                             int64_t row0_global, const float* eps_dev, float* grads_dev) {
    return grad_step_impl(h, X_dev, ldx, rows, global_rows, row0_global, eps_dev, grads_dev, false,
                          "smrt_vae_grad_step_async");
}

// L2 + clip + optimizer on an already reduced gradient; the step's cost goes to cost_dst (device memory).
static int enqueue_apply(smrt_vae* h, const float* grads_dev, float* cost_dst) {
    OptSeg segs[MAX_LAYERS];
    fill_segs(h, segs);
    OptArgs p{};
    std::memcpy(p.seg, segs, sizeof(OptSeg) * h->NL); p.n_seg = h->NL; p.P = h->P_int;
    p.W = h->W; p.G = grads_dev; p.splits = 1; p.slab_stride = h->P_int;
    p.s0 = h->s0; p.s1 = h->s1; p.l2_partials = h->l2_p; p.st = h->st;
    p.opt = h->cfg.optimizer; p.clip = h->cfg.clip; p.has_drop = 0; p.has_l2 = h->has_l2;
    p.lr = h->cfg.learning_rate; p.l2 = h->has_l2 ? h->cfg.l2_penalty : 0.f;
    p.gmin = h->cfg.gclip_min; p.gmax = h->cfg.gclip_max; p.keep = 1.f;
    launch_optimizer(h->stream, p, nullptr);
    h->img_valid = false;
    FinalizeArgs f{};
    f.l2p = h->l2_p; f.n_l2 = h->n_l2; f.presummed = grads_dev + h->P_int; f.l2 = p.l2;
    f.cost_out = cost_dst; f.st = h->st; f.advance = 1;
    launch_finalize(h->stream, f);
    HIPC(hipGetLastError());
    return 0;
}

int smrt_vae_apply_grads(smrt_vae* h, const float* grads_dev, float* cost_host) {
    if (!h || !grads_dev) return fail(-1, "smrt_vae_apply_grads: null argument");
    if (((uintptr_t)grads_dev & 15) != 0) return fail(-1, "smrt_vae_apply_grads: grads_dev must be 16-byte aligned");
    HIPC(hipSetDevice(h->device));
    TRY(ensure_cost(h, 1));
    TRY(enqueue_apply(h, grads_dev, h->cost_dev));
    HIPC(hipMemcpyAsync(h->cost_pin, h->cost_dev, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    h->steps += 1;
    if (cost_host) *cost_host = h->cost_pin[0];
    return 0;
}

int smrt_vae_apply_grads_async(smrt_vae* h, const float* grads_dev, float* cost_dev) {
    if (!h || !grads_dev) return fail(-1, "smrt_vae_apply_grads_async: null argument");
    if (((uintptr_t)grads_dev & 15) != 0) return fail(-1, "smrt_vae_apply_grads_async: grads_dev must be 16-byte aligned");
    HIPC(hipSetDevice(h->device));
    TRY(ensure_cost(h, 1));
    TRY(enqueue_apply(h, grads_dev, cost_dev ? cost_dev : h->cost_dev + 1));
    h->steps += 1;
    return 0;
}

int smrt_vae_profile_epochs(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n_rows, int32_t batch_size,
                            int32_t n_epochs, double* ms_by_kind, int64_t* launches_by_kind) {
    if (!h || !ms_by_kind || !launches_by_kind) return fail(-1, "smrt_vae_profile_epochs: null argument");
    HIPC(hipSetDevice(h->device));
    h->prof = true; h->prof_used = 0; h->prof_kind.clear();
    std::vector<float> cost(std::max(n_epochs, 1)), ms(std::max(n_epochs, 1));
    int32_t done = 0;
    int rc = smrt_vae_train_epochs(h, X_dev, ldx, n_rows, batch_size, n_epochs, 0, 0.f, cost.data(), ms.data(), &done);
    h->prof = false;
    if (rc) return rc;
    for (int k = 0; k < PK_COUNT; ++k) { ms_by_kind[k] = 0.0; launches_by_kind[k] = 0; }
    for (size_t i = 0; i < h->prof_kind.size(); ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, h->prof_pool[2 * i], h->prof_pool[2 * i + 1]);
        ms_by_kind[h->prof_kind[i]] += t;
        launches_by_kind[h->prof_kind[i]] += 1;
    }
    return 0;
}

static const long INFER_CHUNK = 32768;
static const long INFER_CHUNK_FUSED = 1L << 20;      // rows of z drawn per launch on the fused path

// Whether infer_fused would take layers [l0, l1] (the same tests it applies, without enqueuing anything).
static bool infer_fused_fits(const smrt_vae* h, int l0, int l1) {
    if (!h->fused || !h->Wimg) return false;
    const char* e = std::getenv("SMRT_HIP_FUSED_INFER");
    if (e && std::atoi(e) == 0) return false;
    int maxw = rup16(h->flay[l0].fi);
    for (int l = l0; l <= l1; ++l) maxw = std::max(maxw, rup16(h->flay[l].fo));
    if (maxw / 16 > 8) return false;
    const int start = h->flay[l0].lds_w, end = h->flay[l1].lds_b + 4 * rup16(h->flay[l1].fo);
    const int ts = std::max(rup16(h->flay[l0].fi), rup16(h->flay[l1].fo)) + 4;
    return (end - start) * 4 + 8 * 16 * ts * 4 <= 160 * 1024;
}

// Layers [l0, l1] of a net on the fused path in ONE persistent launch (fused_infer_kernel).  Returns 1 when the launch was
// enqueued, 0 when this net / call does not qualify (the caller then runs the per-layer GEMM chain), < 0 on error.
struct GenDraw { uint64_t elem0; float loc, scale; };
static int infer_fused(smrt_vae* h, int l0, int l1, const float* in, long ldin, long n,
                       float* out0, long ld0, int split, float* out1, long ld1, const GenDraw* gen = nullptr) {
    if (!h->fused || !h->Wimg || n <= 0) return 0;
    // in-kernel draw: whole Philox blocks only, at most two k-blocks of latent factors
    if (!in && !(gen && (h->flay[l0].fi & 3) == 0 && h->flay[l0].fi <= 32 && (gen->elem0 & 3) == 0)) return 0;
    const char* e = std::getenv("SMRT_HIP_FUSED_INFER");
    if (e && std::atoi(e) == 0) return 0;
    InferArgs a{};
    const int start = h->flay[l0].lds_w;
    const int end = h->flay[l1].lds_b + 4 * rup16(h->flay[l1].fo);
    int maxw = rup16(h->flay[l0].fi);
    for (int l = l0; l <= l1; ++l) {
        const FusedLayer& q = h->flay[l];
        InferStage& t = a.st[a.n_stages++];
        t.K16 = rup16(q.fi); t.N = q.fo; t.N16 = rup16(q.fo); t.w_off = q.lds_w - start; t.nn = q.nn;
        t.b_off = q.lds_b - start; t.act = q.act;
        maxw = std::max(maxw, t.N16);
    }
    a.maxt = maxw / 16;
    if (a.maxt > 8) return 0;                                        // wider layers keep the per-layer chain
    a.tile_stride = std::max(rup16(h->flay[l0].fi), rup16(h->flay[l1].fo)) + 4;
    a.img = h->Wimg + start; a.img_n4 = (end - start) / 4;
    a.in = in; a.ldin = ldin; a.K = h->flay[l0].fi; a.n = n;
    if (!in) { a.gen_seed = h->cfg.seed; a.gen_elem0 = gen->elem0; a.gen_loc = gen->loc; a.gen_scale = gen->scale; }
    a.out0 = out0; a.ld0 = ld0; a.split = split; a.out1 = out1; a.ld1 = ld1;
    const int N = h->flay[l1].fo;
    a.vec_in = (in && a.K % 4 == 0 && ldin % 4 == 0 && ((uintptr_t)in & 15) == 0) ? 1 : 0;
    a.vec_out = (split >= N && N % 4 == 0 && ld0 % 4 == 0 && ((uintptr_t)out0 & 15) == 0) ? 1 : 0;
    // the image twin must hold W itself: under dropout it follows the post-dropout weights of the last step
    // (inference runs with keep = 1, autoencoder.py:448) and training rebuilds it every step anyway
    if (h->has_drop) {
        launch_build_image(h->stream, h->W, h->stage_tab, h->Wimg, (int)(h->P_int / 4));
        h->img_valid = false;
    } else if (!h->img_valid) {
        launch_build_image(h->stream, h->W, h->stage_tab, h->Wimg, (int)(h->P_int / 4));
        h->img_valid = true;
    }
#ifdef SMRT_STAMPS
    static unsigned long long* idbg = nullptr;
    if (!idbg) { (void)hipMalloc((void**)&idbg, 64 * 8); (void)hipMemset(idbg, 0, 64 * 8); }
    a.dbg = idbg;
#endif
    if (launch_fused_infer(h->stream, a) != 0) return 0;
#ifdef SMRT_STAMPS
    {   // diagnostic build: wave 0 of workgroup 0, its third block of rows (wall_clock64 = 100 MHz; clock64 = shader clock)
        (void)hipStreamSynchronize(h->stream);
        unsigned long long hb[64];
        (void)hipMemcpy(hb, idbg, 64 * 8, hipMemcpyDeviceToHost);
        static int once = 0;
        if (once++ < 2 && n > 100000) {
            fprintf(stderr, "infer wg0: staged +%.2f us, wave 0 done +%.2f us; last wg: entry +%.2f, staged +%.2f, wave 0 done +%.2f us\n",
                    (double)(hb[21] - hb[20]) * 0.01, (double)(hb[22] - hb[20]) * 0.01, (double)((long long)hb[24] - (long long)hb[20]) * 0.01,
                    (double)((long long)hb[25] - (long long)hb[20]) * 0.01, (double)((long long)hb[26] - (long long)hb[20]) * 0.01);
            const int ids[] = {0, 1, 2, 3, 4, 5, 6, 10, 11};
            int prev = 0;
            for (int k = 1; k < 9; ++k) {
                const int i = ids[k];
                if (!hb[i]) continue;
                fprintf(stderr, "infer stamp %d: +%.2f us  (%.0f shader cycles)\n", i, (double)(hb[i] - hb[prev]) * 0.01,
                        (double)(hb[32 + i] - hb[32 + prev]));
                prev = i;
            }
        }
    }
#endif
    return 1;
}

int smrt_vae_encode(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n, float* mu_dev, float* logsig_dev) {
    if (!h || !mu_dev || (!logsig_dev && !h->ae)) return fail(-1, "smrt_vae_encode: null argument");
    if (n < 0) return fail(-1, "smrt_vae_encode: n must be >= 0");
    if (n == 0) return 0;
    TRY(check_mat(X_dev, ldx, h->D, "smrt_vae_encode: X"));
    HIPC(hipSetDevice(h->device));
    if (!h->ae) {
        const int f = infer_fused(h, 0, h->L, X_dev, ldx, n, mu_dev, h->Z, h->Z, logsig_dev, h->Z);
        if (f < 0) return f;
        if (f) { HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(h->stream)); return 0; }
    }
    const long chunk = std::min<long>(n, INFER_CHUNK);
    TRY(ensure_infer(h, chunk));
    for (long r0 = 0; r0 < n; r0 += chunk) {
        const int rows = (int)std::min<long>(chunk, n - r0);
        const float* in = X_dev + r0 * ldx; long ldin = ldx;
        const int last = h->ae ? h->L - 1 : h->L;
        for (int l = 0; l <= last; ++l) {
            const LayerI& y = h->lay[l];
            GemmArgs a{};
            a.A = in; a.lda = ldin; a.B = h->W + y.off; a.ldb = y.ld; a.bias = h->W + y.off + (long)y.fi * y.ld;
            a.M = rows; a.N = y.fo; a.K = y.fi; a.act = y.act; a.Z = h->Z;
            if (l < last) {
                a.C = h->ibuf[l & 1]; a.ldc = y.ld;
                launch_gemm(h->stream, GEMM_NN, EPI_BIAS_ACT, a);
                in = a.C; ldin = y.ld;
            } else if (h->ae) {
                a.C = mu_dev + r0 * h->Zc; a.ldc = h->Zc;              // the code, [n][h_L] contiguous
                launch_gemm(h->stream, GEMM_NN, EPI_BIAS_ACT, a);
            } else {
                a.C = mu_dev + r0 * h->Z; a.out2 = logsig_dev + r0 * h->Z;
                launch_gemm(h->stream, GEMM_NN, EPI_HEADS_SPLIT, a);
            }
        }
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

static int decode_chunk(smrt_vae* h, const float* Zin, long ldz, int rows, float* out, long ldo,
                        float* ib0 = nullptr, float* ib1 = nullptr) {
    float* ib[2] = {ib0 ? ib0 : h->ibuf[0], ib1 ? ib1 : h->ibuf[1]};
    const float* in = Zin; long ldin = ldz;
    for (int l = h->dec0; l < h->NL; ++l) {
        const LayerI& y = h->lay[l];
        GemmArgs a{};
        a.A = in; a.lda = ldin; a.B = h->W + y.off; a.ldb = y.ld; a.bias = h->W + y.off + (long)y.fi * y.ld;
        a.M = rows; a.N = y.fo; a.K = y.fi; a.act = y.act; a.Z = h->Z;
        if (l < h->NL - 1) { a.C = ib[l & 1]; a.ldc = y.ld; }
        else               { a.C = out; a.ldc = ldo; }
        launch_gemm(h->stream, GEMM_NN, EPI_BIAS_ACT, a);
        in = a.C; ldin = y.ld;
    }
    return 0;
}

int smrt_vae_decode(smrt_vae* h, const float* Z_dev, int64_t ldz, int64_t n, float* out_dev, int64_t ldo) {
    if (!h || !out_dev) return fail(-1, "smrt_vae_decode: null argument");
    if (n < 0) return fail(-1, "smrt_vae_decode: n must be >= 0");
    if (n == 0) return 0;
    TRY(check_mat(Z_dev, ldz, h->Zc, "smrt_vae_decode: Z"));
    if (ldo < h->D) return fail(-1, "smrt_vae_decode: ldo must be >= n_features");
    HIPC(hipSetDevice(h->device));
    {
        const int f = infer_fused(h, h->dec0, h->NL - 1, Z_dev, ldz, n, out_dev, ldo, h->D, nullptr, 0);
        if (f < 0) return f;
        if (f) { HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(h->stream)); return 0; }
    }
    const long chunk = std::min<long>(n, INFER_CHUNK);
    TRY(ensure_infer(h, chunk));
    for (long r0 = 0; r0 < n; r0 += chunk) {
        const int rows = (int)std::min<long>(chunk, n - r0);
        TRY(decode_chunk(h, Z_dev + r0 * ldz, ldz, rows, out_dev + r0 * ldo, ldo));
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

int smrt_vae_generate(smrt_vae* h, int64_t n, uint64_t philox_offset, float loc, float scale,
                      float* out_dev, int64_t ldo) {
    if (!h || !out_dev) return fail(-1, "smrt_vae_generate: null argument");
    if (n < 0) return fail(-1, "smrt_vae_generate: n must be >= 0");
    if (n == 0) return 0;
    if (h->ae) return fail(-1, "smrt_vae_generate: a plain auto-encoder has no generative prior");
    if (ldo < h->D) return fail(-1, "smrt_vae_generate: ldo must be >= n_features");
    HIPC(hipSetDevice(h->device));
    const long ldz = rup4(h->Z);
    const bool fused_ok = h->fused && h->Wimg;
    const long chunk = std::min<long>(n, fused_ok ? INFER_CHUNK_FUSED : INFER_CHUNK);
    if (fused_ok) {
        // z drawn inside the persistent decode launch (one launch for the whole call, no z buffer)
        const GenDraw gd{philox_offset, loc, scale};
        const int f = infer_fused(h, h->dec0, h->NL - 1, nullptr, 0, n, out_dev, ldo, h->D, nullptr, 0, &gd);
        if (f < 0) return f;
        if (f) { HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(h->stream)); return 0; }
    }
    TRY(ensure_infer(h, chunk, !fused_ok));
    for (long r0 = 0; r0 < n; r0 += chunk) {
        const long rows = std::min<long>(chunk, n - r0);
        launch_sample_normal(h->stream, h->izb, ldz, rows, h->Z, h->cfg.seed, 3u, 0u,
                             philox_offset + (uint64_t)r0 * h->Z, loc, scale);
        int f = 0;
        if (fused_ok) f = infer_fused(h, h->dec0, h->NL - 1, h->izb, ldz, rows, out_dev + r0 * ldo, ldo, h->D, nullptr, 0);
        if (f < 0) return f;
        if (!f) {
            // (only reached when the fused launch was refused: the z chunk is cut to the per-layer path's buffers)
            TRY(ensure_infer(h, std::min<long>(rows, INFER_CHUNK), true));
            for (long c0 = 0; c0 < rows; c0 += INFER_CHUNK) {
                const int cr = (int)std::min<long>(INFER_CHUNK, rows - c0);
                TRY(decode_chunk(h, h->izb + c0 * ldz, ldz, cr, out_dev + (r0 + c0) * ldo, ldo));
            }
        }
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- the entry point that allocates nothing (include/smrt_hip.h: smrt_vae_generate_noalloc) ----------------------
// smrt_vae_generate draws its latent chunk into a library-allocated buffer (ensure_infer -> hipMalloc) and, for nets off
// the fused path, ping-pongs the decoder's activations through two more.  Here the same kernels take those buffers from
// ONE file-scope device array: each GPU the library runs on holds its own copy (SMRT_NOALLOC_SCRATCH_BYTES) at its own
// address until the process ends.  `scratch` below is only a host-side stand-in for that array: host code never passes
// it to a kernel or a memset; it asks the runtime for the array's address on the current device, once per device, and
// uses that pointer.
#define SMRT_NOALLOC_SCRATCH_FLOATS (SMRT_NOALLOC_SCRATCH_BYTES / 4)
static __device__ __align__(256) __attribute__((used)) float scratch[SMRT_NOALLOC_SCRATCH_FLOATS];
static float* g_scratch_ptr[SMRT_MAX_DEVICES] = {nullptr};
static std::mutex g_scratch_mu;
static std::mutex g_noalloc_mu[SMRT_MAX_DEVICES];     // one call at a time per GPU: every call on a GPU shares that GPU's array

// Address of this device's copy of `scratch` (resolved on first use, with `device` current).
static int scratch_on(int device, float** out) {
    if (device < 0 || device >= SMRT_MAX_DEVICES) return fail(-1, "smrt_vae_generate_noalloc: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(-2, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

int64_t smrt_vae_generate_noalloc_max_rows(const smrt_vae* h) {
    if (!h || h->ae) return -1;
    const long ldz = rup4(h->Z);
    const bool fused_ok = h->fused && h->Wimg && h->fplan.img_total > 0 && infer_fused_fits(h, h->dec0, h->NL - 1);
    const long per_row = fused_ok ? ldz : ldz + 2L * h->maxld;      // z (+ two activation rows off the fused path)
    return (long)SMRT_NOALLOC_SCRATCH_FLOATS / per_row;
}

int smrt_vae_generate_noalloc(smrt_vae* h, int64_t n, uint64_t philox_offset, float loc, float scale,
                              float* out_dev, int64_t ldo) {
    if (!h || !out_dev) return fail(-1, "smrt_vae_generate_noalloc: null argument");
    if (n < 0) return fail(-1, "smrt_vae_generate_noalloc: n must be >= 0");
    if (h->ae) return fail(-1, "smrt_vae_generate_noalloc: a plain auto-encoder has no generative prior");
    if (ldo < h->D) return fail(-1, "smrt_vae_generate_noalloc: ldo must be >= n_features");
    const int64_t cap = smrt_vae_generate_noalloc_max_rows(h);
    if (n > cap)
        return fail(-5, "smrt_vae_generate_noalloc: " + std::to_string(n) + " rows exceed the " + std::to_string(cap) +
                        " rows this net can generate from the " + std::to_string(SMRT_NOALLOC_SCRATCH_BYTES >> 20) +
                        " MiB static scratch; use smrt_vae_generate or split the call");
    if (n == 0) return 0;
    if (h->device < 0 || h->device >= SMRT_MAX_DEVICES) return fail(-1, "smrt_vae_generate_noalloc: device index out of range");
    HIPC(hipSetDevice(h->device));
    const bool fused_ok = h->fused && h->Wimg && infer_fused_fits(h, h->dec0, h->NL - 1);
    if (fused_ok) {
        // where smrt_vae_generate draws z inside the persistent decode launch, so does this call: no latent buffer at all
        // (infer_fused enqueues nothing and returns 0 when the in-kernel draw does not apply: ragged Z, unaligned offset)
        const GenDraw gd{philox_offset, loc, scale};
        const int f = infer_fused(h, h->dec0, h->NL - 1, nullptr, 0, n, out_dev, ldo, h->D, nullptr, 0, &gd);
        if (f < 0) return f;
        if (f) { HIPC(hipGetLastError()); HIPC(hipStreamSynchronize(h->stream)); return 0; }
    }
    // the calls of all handles and host threads on this GPU take turns on its array (held until the stream has drained)
    std::lock_guard<std::mutex> turn(g_noalloc_mu[h->device]);
    float* sp = nullptr;
    TRY(scratch_on(h->device, &sp));
    const long ldz = rup4(h->Z);
    // carve: [z : n x ldz] [activations A : n x maxld] [activations B : n x maxld], each piece 256-byte aligned by
    // construction (ldz and maxld are multiples of 4 floats; round the piece sizes up to 64 floats)
    const size_t zf = ((size_t)n * ldz + 63) & ~size_t(63);
    const size_t af = fused_ok ? 0 : (((size_t)n * h->maxld + 63) & ~size_t(63));
    if (zf + 2 * af > (size_t)SMRT_NOALLOC_SCRATCH_FLOATS) return fail(-5, "smrt_vae_generate_noalloc: internal: carve exceeds the scratch");
    float* zbuf = sp;
    float* ib0 = af ? sp + zf : nullptr;
    float* ib1 = af ? sp + zf + af : nullptr;
    // the array keeps the last call's contents: pad columns of the pieces must read as zero
    HIPC(hipMemsetAsync(sp, 0, (zf + 2 * af) * sizeof(float), h->stream));
    launch_sample_normal(h->stream, zbuf, ldz, n, h->Z, h->cfg.seed, 3u, 0u, philox_offset, loc, scale);
    int f = 0;
    if (fused_ok) f = infer_fused(h, h->dec0, h->NL - 1, zbuf, ldz, n, out_dev, ldo, h->D, nullptr, 0);
    if (f < 0) return f;
    if (!f) {
        if (!af) return fail(-4, "smrt_vae_generate_noalloc: the fused launch was refused and no activation scratch was carved");
        for (long c0 = 0; c0 < n; c0 += INFER_CHUNK) {
            const int cr = (int)std::min<long>(INFER_CHUNK, n - c0);
            TRY(decode_chunk(h, zbuf + c0 * ldz, ldz, cr, out_dev + c0 * ldo, ldo, ib0, ib1));
        }
    }
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

int smrt_vae_sample_eps(smrt_vae* h, int64_t step, int64_t row0, int32_t rows, float* eps_dev) {
    if (!h || !eps_dev) return fail(-1, "smrt_vae_sample_eps: null argument");
    HIPC(hipSetDevice(h->device));
    launch_sample_normal(h->stream, eps_dev, h->Z, rows, h->Z, h->cfg.seed, 1u, (uint32_t)step,
                         (uint64_t)row0 * h->Z, 0.f, 1.f);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

}  // extern "C"
