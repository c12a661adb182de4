// The persistent training kernel's entry points: the generic kernel (any LDS-resident net), the registry of the
// sequence-specialised instantiations (persist_seq.hip) and the launch.  The kernel itself: persist_body.hpp.
#include "persist_body.hpp"

#include <cstdlib>

namespace smrt {

__global__ __launch_bounds__(FTHREADS) void persist_train_kernel(const PersistArgs a) { persist_train_body<SeqGeneric>(a); }

// persist_seq.hip, compiled once per hidden width (Makefile): nullptr when the instantiation does not exist
persist_kernel_t persist_seq_kernel_h2(int db, int zb, int z2b);
persist_kernel_t persist_seq_kernel_h4(int db, int zb, int z2b);
persist_kernel_t persist_seq_kernel_h8(int db, int zb, int z2b);

// The instantiation written out for this stage table, if there is one: a VAE with one hidden layer, sigmoid everywhere, every
// padded width 16 / 32 / 64 / 128 -- checked against the records themselves, so a table of any other shape gets the generic kernel.
persist_kernel_t persist_pick_kernel(const StageU* st, int n_stages) {
    const char* e = std::getenv("SMRT_HIP_PERSIST_SEQ");             // (read per call: tests compare the two builds)
    if ((e && std::atoi(e) == 0) || n_stages != 7 || !st) return persist_train_kernel;
    auto kind = [&](int i) { return st[i].packed & 255; };
    auto act = [&](int i) { return (st[i].packed >> 8) & 255; };
    auto nb = [&](int i) { return st[i].nb_all; };
    auto nt = [&](int i) { return st[i].N16 >> 4; };
    static const int want[7] = {FK_FWD_ACT, FK_FWD_HEADS, FK_FWD_ACT, FK_FWD_BCE, FK_BWD_ACT, FK_BWD_LATENT, FK_BWD_ACT};
    for (int i = 0; i < 7; ++i)
        if (kind(i) != want[i] || act(i) != STAGE_ACT_SIGMOID ||
            st[i].nn != (want[i] >= FK_BWD_ACT ? 16 * nb(i) : st[i].N16) + 1) return persist_train_kernel;
    const int db = nb(0), hb = nt(0), zb = nb(2), z2b = nt(1);
    const bool same = nb(1) == hb && nt(2) == hb && nb(3) == hb && nt(3) == db && nb(4) == db && nt(4) == hb && nb(5) == hb &&
                      nt(5) == zb && nb(6) == z2b && nt(6) == hb;
    if (!same) return persist_train_kernel;
#ifdef SMRT_SEQ_ONLY8              // (diagnostic and A/B builds carry the 128-wide instantiations only)
    persist_kernel_t k = hb == 8 ? persist_seq_kernel_h8(db, zb, z2b) : nullptr;
#else
    persist_kernel_t k = hb == 2 ? persist_seq_kernel_h2(db, zb, z2b) : hb == 4 ? persist_seq_kernel_h4(db, zb, z2b)
                       : hb == 8 ? persist_seq_kernel_h8(db, zb, z2b) : nullptr;
#endif
    return k ? k : persist_train_kernel;
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
