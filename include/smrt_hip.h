/*
 * smrt_hip.h -- C ABI of the MI355X (gfx950) VAE train-and-generate library.
 *
 * The reference (tgsmith61591/smrt) has no FFI: its arithmetic leaves Python at four
 * tf.Session.run call sites.  Each entry point below names the call site it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - return 0 on success, negative on error; smrt_last_error() gives the message (thread-local).
 *   - "dev" pointers are device memory on the handle's GPU, 16-byte aligned, fp32 row-major.
 *     Input matrices carry a leading dimension (row stride, in floats) that must be a
 *     multiple of 4; pad columns are never read as data and never written.
 *   - "host" pointers are ordinary host memory.
 *   - a handle is bound to one device and one stream; not thread-safe per handle, independent
 *     handles may be driven from different host threads (one per-class VAE job per GPU).
 *   - every call returns after its work on the handle's stream has completed unless it says
 *     "async".
 *
 * Flat parameter order ("ABI order", used by set/get_params, get_opt_state and gradients):
 *   for each dense layer in [enc_1 .. enc_L, z_mean head, z_log_sigma head, dec_1 .. dec_{L+1}]
 *   (smrt/autoencode/layer.py:292-340): W[fan_in][fan_out] row-major, then b[fan_out].
 */
#ifndef SMRT_HIP_H
#define SMRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smrt_vae smrt_vae;

/* smrt/autoencode/autoencoder.py:28-34 (PERMITTED_ACTIVATIONS) */
enum { SMRT_ACT_ELU = 0, SMRT_ACT_IDENTITY = 1, SMRT_ACT_RELU = 2, SMRT_ACT_SIGMOID = 3, SMRT_ACT_TANH = 4 };
/* smrt/autoencode/autoencoder.py:38-48 (PERMITTED_OPTIMIZERS); 'momentum' and 'adagrad-da' cannot be
 * constructed the way the reference calls them (base.py:138,141) and have no code here. */
enum { SMRT_OPT_SGD = 0, SMRT_OPT_RMSPROP = 1, SMRT_OPT_ADAM = 2, SMRT_OPT_ADAGRAD = 3,
       SMRT_OPT_ADADELTA = 4, SMRT_OPT_PROXIMAL_SGD = 5, SMRT_OPT_PROXIMAL_ADAGRAD = 6 };

/* smrt/autoencode/autoencoder.py:285 (VariationalAutoEncoder) / :91 (AutoEncoder: symmetric encoder/decoder, the
 * user's activation on every layer, loss = mean over all B*D elements of (x - x_hat)^2, autoencoder.py:260). */
enum { SMRT_MODEL_VAE = 0, SMRT_MODEL_AE = 1 };

#define SMRT_MAX_HIDDEN 6

typedef struct smrt_vae_cfg {
    int32_t  n_features;                 /* D */
    int32_t  n_hidden_layers;            /* L, 1..SMRT_MAX_HIDDEN */
    int32_t  hidden[SMRT_MAX_HIDDEN];    /* h_1..h_L */
    int32_t  n_latent;                   /* Z (already resolved from a float, layer.py:254-257) */
    int32_t  activation;                 /* SMRT_ACT_*; last decoder layer is always sigmoid (layer.py:333-338) */
    int32_t  optimizer;                  /* SMRT_OPT_* */
    float    learning_rate;
    float    l2_penalty;                 /* < 0 : no regularisation (l2_penalty=None, base.py:128) */
    int32_t  clip;                       /* base.py:137 */
    float    gclip_min, gclip_max;       /* base.py:144 */
    float    bce_lo, bce_hi;             /* float32(eps), float32(1-eps): _ae_utils.py:31 */
    float    dropout_keep;               /* (0,1]; dropout acts on the WEIGHTS, layer.py:365 */
    uint64_t seed;                       /* Philox key for epsilon / dropout / fast generate */
    int32_t  max_batch;                  /* rows per training step the workspace is sized for */
    int32_t  model;                      /* SMRT_MODEL_*; for SMRT_MODEL_AE n_latent, bce_lo/hi are ignored, the flat
                                          * parameter order is enc_1..enc_L, dec_1..dec_L, smrt_vae_encode writes the code
                                          * [n][h_L] to mu_dev (logsig_dev may be NULL), smrt_vae_decode takes the code */
} smrt_vae_cfg;

/* Replaces graph construction + tf.global_variables_initializer (autoencoder.py:439-486, base.py:185-190).
 * Weights start at zero; the host initialiser uploads them with smrt_vae_set_params. */
int  smrt_vae_create(const smrt_vae_cfg* cfg, int device, smrt_vae** out);
/* Replaces Session.close (base.py:115-121). */
void smrt_vae_destroy(smrt_vae* h);

/* Number of floats in the flat ABI-order parameter vector. */
int64_t smrt_vae_param_count(const smrt_vae* h);
/* tf.Variable initial values / a checkpoint (layer.py:417,464); resets nothing else. host -> device. */
int  smrt_vae_set_params(smrt_vae* h, const float* host_flat, int64_t n);
int  smrt_vae_get_params(smrt_vae* h, float* host_flat, int64_t n);
/* Optimizer slots in ABI order; slot 0/1 = (rms, momentum) | (m, v) | (accumulator, -) | (accum, accum_update).
 * Also resets/reads the step counter (Adam's beta powers follow it). */
int  smrt_vae_get_opt_state(smrt_vae* h, int slot, float* host_flat, int64_t n);
int  smrt_vae_reset_opt_state(smrt_vae* h);
int64_t smrt_vae_step_count(const smrt_vae* h);

/* Inject dropout keep-masks (0/1 floats, ABI order, bias entries ignored) for the following steps;
 * NULL returns to the Philox stream.  Parity hook for tf.nn.dropout (layer.py:365). */
int  smrt_vae_set_dropout_masks(smrt_vae* h, const float* host_flat_or_null, int64_t n);

/* One sess.run([optimizer, cost]) (base.py:215-217): forward, loss, backward, L2, clip, update.
 *   X_dev       [rows][ldx] training rows (ldx % 4 == 0), rows <= max_batch
 *   eps_dev     [rows][n_latent] contiguous epsilon to inject, or NULL for the Philox stream
 *   cost_host   the step's cost (pre-update weights), autoencoder.py:479-480
 *   grads_host  optional: d cost / d variables BEFORE clipping, ABI order (tf.gradients, base.py:142) */
int  smrt_vae_train_step(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows,
                         const float* eps_dev, float* cost_host, float* grads_host);

/* The whole _train loop (base.py:198-245) on device: contiguous unshuffled batches with a ragged
 * tail, n_epochs passes, cost of the LAST batch per epoch, optional early stopping
 * |last - c| <= min_change.  One host sync per call (one per epoch under early stopping).
 *   epoch_cost_host[n_epochs], epoch_ms_host[n_epochs], *epochs_done.  Epoch times are device times: the launch that
 *   closes an epoch writes its cost and a 100 MHz time stamp into pinned host memory (an epoch's time is the difference to
 *   the previous stamp); with SMRT_HIP_EPOCH_RING=0 they are hipEvent pairs around each epoch and the cost is copied back. */
int  smrt_vae_train_epochs(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n_rows, int32_t batch_size,
                           int32_t n_epochs, int32_t early_stopping, float min_change,
                           float* epoch_cost_host, float* epoch_ms_host, int32_t* epochs_done);

/* The _train loops of SEVERAL handles side by side (no counterpart in the reference, which fits its per-class VAEs one
 * after the other, smrt/balance/smrt.py:197-223): handle k trains on X_dev[k] ([n_rows[k]][ldx[k]]) with the reference's
 * schedule, exactly as n smrt_vae_train_epochs calls would -- same kernels, same numbers -- but the launches of one kernel
 * at the same position of the handles' steps are issued as ONE grouped launch (gridDim.z = handle), so the small grids of
 * a default-batch step fill the GPU together.  No early stopping.  epoch_cost_host [n_handles][n_epochs];
 * epoch_ms_host [n_epochs] (the epochs are shared).  Returns 0, < 0 on error, or 1 when grouping does not apply to these
 * handles (LDS-resident nets, mini-batches beyond 512 rows, different devices, a single handle): nothing was done and the
 * caller trains them one by one. */
int  smrt_vae_train_epochs_multi(smrt_vae** hs, int32_t n_handles, const float* const* X_dev, const int64_t* ldx,
                                 const int64_t* n_rows, int32_t batch_size, int32_t n_epochs,
                                 float* epoch_cost_host, float* epoch_ms_host);

/* Which training path a handle takes (no counterpart in the reference: its one path is Session.run per mini-batch,
 * base.py:215-217).  Nets whose weights fit one CU's LDS train an epoch in ONE persistent launch whose workgroups must all
 * be resident at once; a device holds smrt_persistent_budget(device) of them per process (5/8 of its CUs, divided by
 * LOCAL_WORLD_SIZE / visible devices when several ranks share a GPU; SMRT_HIP_PERSIST_BUDGET overrides).
 *   smrt_vae_persistent_workgroups: workgroups such a launch of h would hold at mini-batches of batch_rows rows; 0 when h
 *                                   or the schedule does not qualify.
 *   smrt_vae_set_persistent       : mode -1 (default) -- taken when the workgroups fit what is left of the budget at the
 *                                   moment of the call (first come, first served between concurrent calls);
 *                                   0 -- never (always the two-launch step); 1 -- always: the caller has planned the
 *                                   budget over its handles itself, so that which of several concurrent fits take this
 *                                   path does not depend on thread timing (smrt_amd/autoencode/vae.py: fit_group).
 * A persistent launch that cannot become co-resident gives up after 2 s; smrt_vae_train_epochs then runs that epoch and
 * the ones after it again on the two-launch step (weights and slots are intact), unless SMRT_HIP_PERSIST_NO_FALLBACK=1,
 * in which case it returns -6. */
int32_t smrt_persistent_budget(int32_t device);
int32_t smrt_vae_persistent_workgroups(const smrt_vae* h, int32_t batch_rows);
int  smrt_vae_set_persistent(smrt_vae* h, int32_t mode);
/* Which build of the persistent kernel h's launches use (known once its training workspace exists; 0 before, and for nets
 * off this path): 1 -- the generic one, a loop over the net's stage records; 2 -- one written out for the net's shape (a VAE with one
 * hidden layer and sigmoid activations whose padded widths are 16 / 32 / 64 / 128 features, 32 / 64 / 128 hidden, <= 32 latent:
 * no loop, each stage its structure-specialised body).  Both run the same arithmetic in the same order: results are
 * bit-identical (tests/test_gpu_parity.py).  SMRT_HIP_PERSIST_SEQ=0 keeps every net on the generic one. */
int32_t smrt_vae_persistent_form(const smrt_vae* h);

/* Data-parallel halves of a step (no reference counterpart): grad_step leaves
 * [internal-layout gradient sums | sum of row losses] scaled by 1/global_rows in grads_dev
 * (smrt_vae_dp_count floats) so ranks can all-reduce(SUM) it; apply_grads adds L2, clips, updates.
 *   row0_global: index of this shard's first row inside the global batch (epsilon is drawn per global row). */
int64_t smrt_vae_dp_count(const smrt_vae* h);
int  smrt_vae_grad_step(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, int64_t global_rows,
                        int64_t row0_global, const float* eps_dev, float* grads_dev);
int  smrt_vae_apply_grads(smrt_vae* h, const float* grads_dev, float* cost_host);
/* The same two halves, enqueued only ("async": no host synchronisation, no read-back), so that a
 * host loop of grad_step -> all-reduce -> apply_grads never stalls the GPU.  rows == 0 is allowed (a rank
 * whose stripe of a ragged tail batch is empty): grads_dev is zero-filled.  cost_dev: one device float that
 * receives the step's cost, or NULL. */
int  smrt_vae_grad_step_async(smrt_vae* h, const float* X_dev, int64_t ldx, int32_t rows, int64_t global_rows,
                              int64_t row0_global, const float* eps_dev, float* grads_dev);
int  smrt_vae_apply_grads_async(smrt_vae* h, const float* grads_dev, float* cost_dev);
/* Lend the handle a stream (no reference counterpart).  own == 0: every later call enqueues on hip_stream
 * (a hipStream_t; NULL is the legacy default stream), e.g. the stream the caller's RCCL collectives are ordered
 * against; own != 0: back to the handle's private stream.  Synchronises the stream in use first. */
int  smrt_vae_set_stream(smrt_vae* h, void* hip_stream, int32_t own);

/* sess.run([z_mean_, z_log_sigma_], {X}) (autoencoder.py:500).  mu/ls are [n][n_latent] contiguous. */
int  smrt_vae_encode(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n, float* mu_dev, float* logsig_dev);
/* sess.run(decode, {z_: Z}) (autoencoder.py:517).  Z_dev [n][ldz], out_dev [n][ldo]. */
int  smrt_vae_decode(smrt_vae* h, const float* Z_dev, int64_t ldz, int64_t n, float* out_dev, int64_t ldo);
/* Fast generate (no reference counterpart for the RNG): z = loc + scale * N(0,1) from the handle's Philox
 * stream at element offset philox_offset, then decode.  The exact-compat path of generate()
 * (autoencoder.py:571, numpy RandomState draws) is host z + smrt_vae_decode. */
int  smrt_vae_generate(smrt_vae* h, int64_t n, uint64_t philox_offset, float loc, float scale,
                       float* out_dev, int64_t ldo);

/* The ONE entry point that allocates nothing.  It has no counterpart in the reference; it is for callers that cannot
 * allocate device memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * result as smrt_vae_generate (same kernels, same Philox stream), but the latent chunk -- and, for nets off the fused
 * path, the decoder's two ping-pong activation buffers -- live in a static device array of
 * SMRT_NOALLOC_SCRATCH_BYTES per GPU instead of library-allocated buffers.  No hipMalloc / hipFree / allocating
 * routine is called.  n may not exceed smrt_vae_generate_noalloc_max_rows(h) (= scratch floats / (rup4(Z) [+ 2 x widest
 * layer])); a larger n returns -5 with a message and writes nothing.  Every call on a GPU shares that GPU's array: the
 * calls of one PROCESS take turns on it (a per-device mutex, held until the call's stream has drained); calls from
 * different processes on one GPU each have their own copy.  The array is cleared at the start of each call that uses it;
 * where smrt_vae_generate draws z inside its decode launch (LDS-resident nets, Z % 4 == 0, Z <= 32, philox_offset % 4 ==
 * 0) so does this call, and the array is not touched. */
#define SMRT_NOALLOC_SCRATCH_BYTES (8 << 20)
int64_t smrt_vae_generate_noalloc_max_rows(const smrt_vae* h);
int  smrt_vae_generate_noalloc(smrt_vae* h, int64_t n, uint64_t philox_offset, float loc, float scale,
                               float* out_dev, int64_t ldo);

/* Debug/parity: the epsilon the kernels would draw for training step `step`, rows [row0, row0+rows). */
int  smrt_vae_sample_eps(smrt_vae* h, int64_t step, int64_t row0, int32_t rows, float* eps_dev);

/* Kernel launches enqueued by the most recent epoch. */
int  smrt_vae_last_epoch_launches(const smrt_vae* h);

/* Measurement aid (no reference counterpart): run n_epochs of the _train loop eagerly with a hipEvent pair around
 * every kernel launch on the handle's stream; returns total ms and launch count per kernel kind
 * (SMRT_PK_*).  It trains exactly like smrt_vae_train_epochs (weights advance). */
enum { SMRT_PK_DROPOUT = 0, SMRT_PK_FWD, SMRT_PK_FWD_BCE, SMRT_PK_REPARAM, SMRT_PK_BWD, SMRT_PK_BWD_LATENT,
       SMRT_PK_DW, SMRT_PK_OPT, SMRT_PK_FINAL, SMRT_PK_REDUCE, SMRT_PK_FUSED, SMRT_PK_FUSED_DW_OPT, SMRT_PK_PERSIST,
       SMRT_PK_COUNT };
int  smrt_vae_profile_epochs(smrt_vae* h, const float* X_dev, int64_t ldx, int64_t n_rows, int32_t batch_size,
                             int32_t n_epochs, double* ms_by_kind, int64_t* launches_by_kind);

/* Host-side draws on numpy's LEGACY RandomState stream, bit for bit (smrt_amd/csrc/host_rng.cpp).  They replace the two
 * sequential numpy loops that are the wall-clock of smrt_balance once the VAE trains on the GPU behind them:
 *   smrt_rng_normal      = random_state.normal(loc, scale, size)   the latent draws of generate / generate_from_sample
 *                          (smrt/autoencode/autoencoder.py:549-572, called per class from smrt/balance/smrt.py:226-238)
 *   smrt_rng_permutation = random_state.permutation(np.arange(n))  the output shuffle (smrt/balance/smrt.py:244)
 * State in / out in RandomState.get_state(legacy=True) form: key[624] (uint32), pos in [0, 624], has_gauss, cached_gauss.
 * All pointers are HOST memory.  Returns 0, -1 on a bad argument or -2 when the work arrays cannot be allocated (state
 * untouched either way).  The work arrays are kept per calling thread between calls (up to 256 MB). */
int  smrt_rng_normal(uint32_t* key, int32_t* pos, int32_t* has_gauss, double* cached_gauss, double loc, double scale,
                     double* out, int64_t n, int32_t n_threads);
int  smrt_rng_permutation(uint32_t* key, int32_t* pos, int64_t* out, int64_t n);

/* Host-side output assembly of smrt_balance: replaces the reference's X.copy() + per-class np.vstack + shuffle gather
 * (smrt/balance/smrt.py:184,237,244-250) with ONE multi-threaded pass.  All pointers are HOST memory.
 *   out[i][:] = fp64( row order[i] of the virtual concatenation [X (n_orig rows) ; synth (n_synth fp32 rows)] )
 *   X: fp32 (x_is_f32 != 0) or fp64, row stride ldx elements; synth row stride ld_synth; out [n_out][d] contiguous;
 *   order: n_out row numbers in [0, n_orig + n_synth), or NULL for the identity (then n_out == n_orig + n_synth).
 * Returns 0, or -1 on a bad argument (nothing is written). */
int  smrt_assemble_rows(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                        const float* synth, int64_t n_synth, int64_t ld_synth,
                        const int64_t* order, int64_t n_out, double* out, int32_t n_threads);

/* The same pass in two halves (no counterpart in the reference): which = 1 writes only the output rows whose source is X
 * (synth may be NULL; smrt_balance runs this while the GPU still trains), which = 2 only those whose source is a
 * synthetic row (X may be NULL), 0 everything.  Rows of the other kind are left untouched. */
int  smrt_assemble_rows_part(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                             const float* synth, int64_t n_synth, int64_t ld_synth,
                             const int64_t* order, int64_t n_out, double* out, int32_t n_threads, int32_t which);

/* The synthetic half in pieces (no counterpart in the reference): writes only the output rows whose source is one of the
 * synthetic rows [s_lo, s_hi) of synth [n_synth][ld_synth]; order / n_out / out as above (order entries are trusted: run
 * smrt_assemble_rows_part with which = 1 over the same order first, it validates them).  smrt_balance calls it per class,
 * while the next class's rows are still being decoded and downloaded. */
int  smrt_assemble_rows_synth_range(const float* synth, int64_t n_orig, int64_t n_synth, int64_t ld_synth, int64_t d,
                                    const int64_t* order, int64_t n_out, double* out, int32_t n_threads,
                                    int64_t s_lo, int64_t s_hi);

/* The finiteness sweep of sklearn's check_array (smrt/balance/base.py:24) on n_threads cores: 1 when all n elements of
 * the fp32 / fp64 HOST array are finite, 0 otherwise (the caller then lets sklearn raise its own error), -1 on a bad
 * argument. */
int  smrt_all_finite(const void* X, int32_t x_is_f32, int64_t n, int32_t n_threads);

/* Same pass with fp64 appended rows (smote_balance, smrt/balance/smote.py:93,254-260); with n_synth == 0 it is the row
 * gather of over_sample_balance (over.py:77-81) and under_sample_balance (under.py:69-78). */
int  smrt_assemble_rows_f64(const void* X, int32_t x_is_f32, int64_t n_orig, int64_t d, int64_t ldx,
                            const double* synth, int64_t n_synth, int64_t ld_synth,
                            const int64_t* order, int64_t n_out, double* out, int32_t n_threads);

/* ---- smote_balance (smrt/balance/smote.py).  All pointers are DEVICE memory; work is queued on hip_stream.
 * smrt_knn_f64 replaces NearestNeighbors(...).fit(X_sub).kneighbors(X_sub, k, return_distance=False)
 * (smote.py:53-69): for every query row the k candidate rows with the smallest sum_j (q_j - c_j)^2, accumulated in
 * fp64 in feature order without fused multiply-adds (what sklearn's tree metric computes), sorted ascending, equal
 * sums to the lower candidate row.  idx_out [nq][k]; rdist_out [nq][k] (the squared distances) may be NULL.
 * Only the Euclidean metric (p = 2) is provided; 1 <= k <= min(64, nc). */
int  smrt_knn_f64(const double* Q_dev, int64_t nq, int64_t ldq, const double* C_dev, int64_t nc, int64_t ldc,
                  int64_t d, int32_t k, int64_t* idx_out_dev, double* rdist_out_dev, void* hip_stream);
/* smrt_smote_rows_f64 replaces the strategy functions applied per neighbour (smote.py:22-29,84-89):
 *   mode 0 'perturb'    : out[i] = (X[base[i]] - X[nbr[i]]) * r[i] + X[base[i]]      (each operation rounded, no FMA)
 *   mode 1 'interpolate': out[i] = (X[base[i]] + X[nbr[i]]) / 2
 * A row whose indices fall outside [0, n_x) is written as NaN, never read. */
int  smrt_smote_rows_f64(const double* X_dev, int64_t n_x, int64_t ldx, int64_t d, const int64_t* base_idx_dev,
                         const int64_t* nbr_idx_dev, const double* r_dev, int64_t n_rows, int32_t mode,
                         double* out_dev, int64_t ldo, void* hip_stream);

/* ---- diagnostics: achievable peaks and the latency floor the roofline figures are quoted against.
 * No counterpart in the reference; nothing on the product path calls them.
 *   smrt_diag_mfma_peak     : register-resident v_mfma_f32_32x32x2_f32 loop on every CU -> TFLOP/s actually reachable.
 *   smrt_diag_hbm_bandwidth : float4 copy (read+write) and read-only sweep over `bytes` of device memory -> GB/s.
 *   smrt_diag_step_floor    : a captured graph of steps x kernels_per_step dependent one-word kernels of
 *                             `workgroups` x `threads` -> microseconds per step with no arithmetic at all. */
int  smrt_diag_mfma_peak(int32_t device, double* tflops);
int  smrt_diag_mfma_peak_16x16x4(int32_t device, double* tflops);   /* the same loop on v_mfma_f32_16x16x4_f32 */
int  smrt_diag_hbm_bandwidth(int32_t device, int64_t bytes, double* copy_gbs, double* read_gbs);
int  smrt_diag_step_floor(int32_t device, int32_t kernels_per_step, int32_t steps, int32_t workgroups,
                          int32_t threads, double* us_per_step);

const char* smrt_last_error(void);
const char* smrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SMRT_HIP_H */
