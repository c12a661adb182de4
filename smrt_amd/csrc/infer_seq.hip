// The inference kernel (infer_body.hpp) written out per two-layer chain shape, one translation unit per hidden width so that they
// build in parallel: compiled with -DSEQ_HB=2|4|8 (Makefile).  Input / output widths 16 / 32 / 64 / 128 (decode and generate:
// latent -> hidden -> features; encode: features -> hidden -> both heads).
#include "infer_body.hpp"

#ifndef SEQ_HB
#error "compile with -DSEQ_HB=2, 4 or 8"
#endif
#define SEQ_CAT2(a, b) a##b
#define SEQ_CAT(a, b) SEQ_CAT2(a, b)

namespace smrt {

template <int NK0, int NN1>
__global__ __launch_bounds__(1024) void SEQ_CAT(fused_infer_seq_hb, SEQ_HB)(const InferArgs a) {
    fused_infer_body<1024, NK0, SEQ_HB, NN1>(a);
}

infer_kernel_t SEQ_CAT(infer_seq_kernel_h, SEQ_HB)(int nk0, int nn1) {
#define SEQ_ONE(K_, N_) if (nk0 == K_ && nn1 == N_) return SEQ_CAT(fused_infer_seq_hb, SEQ_HB)<K_, N_>;
#define SEQ_K(K_) SEQ_ONE(K_, 1) SEQ_ONE(K_, 2) SEQ_ONE(K_, 4) SEQ_ONE(K_, 8)
    SEQ_K(1) SEQ_K(2) SEQ_K(4) SEQ_K(8)
#undef SEQ_K
#undef SEQ_ONE
    return nullptr;
}

}  // namespace smrt
