// The persistent training kernel's entry points: the generic kernel (any LDS-resident net), the registry of the
// sequence-specialised instantiations (persist_seq.hip) and the launch.  The kernel itself: persist_body.hpp.
#include "persist_body.hpp"

#include <cstdlib>

namespace smrt {

__global__ __launch_bounds__(FTHREADS) void persist_train_kernel(const PersistArgs a) { persist_train_body<SeqGeneric<2>>(a); }

// persist_seq.hip, compiled once per hidden width (Makefile): nullptr when the instantiation does not exist
persist_kernel_t persist_seq_kernel_h2(int db, int zb, int z2b);
persist_kernel_t persist_seq_kernel_h4(int db, int zb, int z2b);
persist_kernel_t persist_seq_kernel_h8(int db, int zb, int z2b);

// Whether this stage table is the pattern the written-out kernels exist for: a VAE with one hidden layer, sigmoid everywhere,
// every padded width 16 / 32 / 64 / 128 -- checked against the records themselves, so a table of any other shape gets the
// generic kernels.
bool seq_vae1_shape(const StageU* st, int n_stages, int& db, int& hb, int& zb, int& z2b) {
    const char* e = std::getenv("SMRT_HIP_PERSIST_SEQ");             // (read per call: tests compare the two builds)
    if ((e && std::atoi(e) == 0) || n_stages != 7 || !st) return false;
    auto nb = [&](int i) { return st[i].nb_all; };
    auto nt = [&](int i) { return st[i].N16 >> 4; };
    static const int want[7] = {FK_FWD_ACT, FK_FWD_HEADS, FK_FWD_ACT, FK_FWD_BCE, FK_BWD_ACT, FK_BWD_LATENT, FK_BWD_ACT};
    for (int i = 0; i < 7; ++i)
        if ((st[i].packed & 255) != want[i] || ((st[i].packed >> 8) & 255) != STAGE_ACT_SIGMOID ||
            st[i].nn != (want[i] >= FK_BWD_ACT ? 16 * nb(i) : st[i].N16) + 1) return false;
    db = nb(0); hb = nt(0); zb = nb(2); z2b = nt(1);
    return nb(1) == hb && nt(2) == hb && nb(3) == hb && nt(3) == db && nb(4) == db && nt(4) == hb && nb(5) == hb &&
           nt(5) == zb && nb(6) == z2b && nt(6) == hb;
}
persist_kernel_t persist_pick_kernel(const StageU* st, int n_stages) {
    int db, hb, zb, z2b;
    if (!seq_vae1_shape(st, n_stages, db, hb, zb, z2b)) return persist_train_kernel;
#ifdef SMRT_SEQ_ONLY8              // (diagnostic and A/B builds carry the 128-wide instantiations only)
    persist_kernel_t k = hb == 8 ? persist_seq_kernel_h8(db, zb, z2b) : nullptr;
#else
    persist_kernel_t k = hb == 2 ? persist_seq_kernel_h2(db, zb, z2b) : hb == 4 ? persist_seq_kernel_h4(db, zb, z2b)
                       : hb == 8 ? persist_seq_kernel_h8(db, zb, z2b) : nullptr;
#endif
    return k ? k : persist_train_kernel;
}

// X of a whole epoch -> blocked transposed [step][16-row block g][rup16(D) features][16 rows], rows past a step's mini-batch and
// features past D zero: what the compute workgroups used to publish per step as 4-byte write-through stores (X never changes
// during a fit).  One workgroup per (step, block): the 16 x D tile goes through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void xt_all_kernel(const float* __restrict__ X, long ldx, long n_rows, int batch, int G, int D,
                                                     float* __restrict__ out) {
    __shared__ float tile[16 * 129];
    const int NF = (D + 15) & ~15;
    const long step = blockIdx.x / G;
    const int g = blockIdx.x % G;
    const long rows = min((long)batch, n_rows - step * batch);
    const long r0 = step * batch + (long)g * 16;
    float* dst = out + ((long)blockIdx.x * NF) * 16;
    for (int f0 = 0; f0 < NF; f0 += 128) {
        const int nf = min(128, NF - f0);
        for (int i = threadIdx.x; i < 16 * nf; i += 256) {
            const int r = i / nf, f = i - r * nf;
            const bool in = (long)g * 16 + r < rows && f0 + f < D;
            tile[r * 129 + f] = in ? X[(r0 + r) * ldx + f0 + f] : 0.0f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * nf; i += 256) {
            const int f = i >> 4, r = i & 15;
            dst[(long)(f0 + f) * 16 + r] = tile[r * 129 + f];
        }
        __syncthreads();
    }
}
void launch_xt_all(hipStream_t s, const float* X, long ldx, long n_rows, int batch, int G, int D, float* XT_all) {
    const long n_steps = (n_rows + batch - 1) / batch;
    SMRT_LAUNCH(xt_all_kernel, dim3((unsigned)(n_steps * G)), dim3(256), 0, s, X, ldx, n_rows, batch, G, D, XT_all);
}

void launch_persist_train(hipStream_t s, persist_kernel_t k, const PersistArgs& a, int n_workgroups) {
    SMRT_LAUNCH(k, dim3(n_workgroups), dim3(FTHREADS), (size_t)a.lds_total * sizeof(float), s, a);
}

int persist_set_max_lds(persist_kernel_t k, int bytes) {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int persist_owner_workgroups(int total_tiles) { return (total_tiles + OWNER_TILES - 1) / OWNER_TILES; }
int persist_max_owner_workgroups() { return PCNT_FW_MAX; }

}  // namespace smrt
