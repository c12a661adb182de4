// Sequence-specialised instantiations of the persistent training kernel (persist_body.hpp: SeqVae1), one translation unit
// per hidden width so that they build in parallel: compiled with -DSEQ_HB=2|4|8 (Makefile).
// Feature widths 16 / 32 / 64 / 128 (DB = 1, 2, 4, 8) x latent classes Z <= 8 (ZB 1, Z2B 1), Z <= 16 (1, 2), Z <= 32 (2, 4).
#include "persist_body.hpp"
#include "fused_body.hpp"

#ifndef SEQ_HB
#error "compile with -DSEQ_HB=2, 4 or 8"
#endif
#define SEQ_CAT2(a, b) a##b
#define SEQ_CAT(a, b) SEQ_CAT2(a, b)

namespace smrt {

template <int DB, int ZB, int Z2B>
__global__ __launch_bounds__(FTHREADS) void SEQ_CAT(persist_train_seq_hb, SEQ_HB)(const PersistArgs a) {
    persist_train_body<SeqVae1<2, DB, SEQ_HB, ZB, Z2B>>(a);
}

persist_kernel_t SEQ_CAT(persist_seq_kernel_h, SEQ_HB)(int db, int zb, int z2b) {
#define SEQ_ONE(DB_, ZB_, Z2B_) if (db == DB_ && zb == ZB_ && z2b == Z2B_) return SEQ_CAT(persist_train_seq_hb, SEQ_HB)<DB_, ZB_, Z2B_>;
#define SEQ_D(DB_) SEQ_ONE(DB_, 1, 1) SEQ_ONE(DB_, 1, 2) SEQ_ONE(DB_, 2, 4)
    SEQ_D(1) SEQ_D(2) SEQ_D(4) SEQ_D(8)
#undef SEQ_D
#undef SEQ_ONE
    return nullptr;
}


// the same for the forward / loss / backward launch of the two-kernel step (fused_body.hpp)
template <int DB, int ZB, int Z2B>
__global__ __launch_bounds__(FTHREADS) void SEQ_CAT(fused_fwd_bwd_seq_hb, SEQ_HB)(const FusedArgs a) {
    fused_fwd_bwd_body<SeqVae1<1, DB, SEQ_HB, ZB, Z2B>>(a);
}
fused_kernel_t SEQ_CAT(fused_seq_kernel_h, SEQ_HB)(int db, int zb, int z2b) {
#define SEQ_ONE(DB_, ZB_, Z2B_) if (db == DB_ && zb == ZB_ && z2b == Z2B_) return SEQ_CAT(fused_fwd_bwd_seq_hb, SEQ_HB)<DB_, ZB_, Z2B_>;
#define SEQ_D(DB_) SEQ_ONE(DB_, 1, 1) SEQ_ONE(DB_, 1, 2) SEQ_ONE(DB_, 2, 4)
    SEQ_D(1) SEQ_D(2) SEQ_D(4) SEQ_D(8)
#undef SEQ_D
#undef SEQ_ONE
    return nullptr;
}

}  // namespace smrt
