/*
 * sdrm_hip.h — C ABI of libsdrm_hip.so, the MI355X (gfx950) denoising engine for SDRM.
 *
 * The reference (Multi-resolution-diffusion-recommender/SDRM) is pure Python on PyTorch and has
 * no FFI layer; the boundary this library replaces is the Python call surface of
 * /root/reference/train_SDRM.py as consumed by main.py:148-175 and
 * hyperparameter_search.py:147,386,691.  Each entry point below cites the
 * reference lines whose device work it replaces.  The Python shim that keeps the
 * reference's names (`train_SDRM`, `sample_ddpm`, `SDRM.forward`, ...) lives in
 * sdrm_amd/train_SDRM.py and binds these symbols with ctypes.
 *
 * Conventions
 *  - Every `const float*` / `float*` / `const int64_t*` / `const uint8_t*` argument whose name does
 *    not end in `_host` is a DEVICE pointer owned by the caller (e.g. `tensor.data_ptr()` of a
 *    contiguous ROCm torch tensor).  The library owns only handle-internal buffers and never
 *    allocates in a step call.
 *  - `stream` is a `hipStream_t` passed as `void*` (NULL = the default stream).  All work is
 *    enqueued on it and nothing synchronises unless stated.
 *  - All arithmetic is fp32; `t` / `Tj` are int64 as in the reference (`torch.randint`, :327).
 *  - Return value: 0 on success, negative `sdrm_status` otherwise; never throws.  The message of the
 *    last failure on a handle is returned by sdrm_last_error().
 *  - A handle is not thread-safe; distinct handles are independent: all tile / path selection state lives in the
 *    handle (read from the SDRM_* environment variables once, in sdrm_create), none of it is process-global.
 *  - Test and tuning hooks (sdrm_debug_*) are declared in sdrm_hip_debug.h, not here.
 *
 * Parameter order of every "flat" vector (P floats) = SDRM.named_parameters() of the reference
 * (train_SDRM.py:86-95):
 *   emb_layer.weight[T,T] emb_layer.bias[T] dnn.0.weight[W,L+T] dnn.0.bias[W] dnn.1.weight[1]
 *   (H>=1: dnn.2.weight[W,W] dnn.2.bias[W] dnn.3.weight[1])  dnn.{2+2H}.weight[L,W] dnn.{2+2H}.bias[L]
 * The H hidden layers share ONE weight/bias/slope (train_SDRM.py:94, quirk Q1).
 */
#ifndef SDRM_HIP_H
#define SDRM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdrm_engine sdrm_engine;

enum sdrm_status {
  SDRM_OK = 0,
  SDRM_ERR_ARG = -1,    /* null pointer / bad enum */
  SDRM_ERR_SHAPE = -2,  /* rows > max_rows, L/W/T/H outside the supported envelope */
  SDRM_ERR_HIP = -3,    /* a HIP runtime call failed */
  SDRM_ERR_STATE = -4,  /* call order violated (e.g. backward before forward) */
  SDRM_ERR_NOMEM = -5,
  SDRM_ERR_RCCL = -6,   /* librccl could not be loaded, or an RCCL call failed */
  SDRM_ERR_DEVICE = -7  /* the device is not the chip this library is built for (gfx950, 256 compute units) */
};

/* Where the randomness of a call comes from (SURVEY.md §8b "two RNG modes"). */
enum sdrm_rng_mode {
  SDRM_RNG_EXPLICIT = 0, /* caller supplies noise / t / dropout keep-masks / z (parity mode) */
  SDRM_RNG_PHILOX = 1    /* counter-based Philox4x32-10 keyed by (seed, step, GLOBAL row, column) */
};

/* Explicit randoms of one train step.  Replaces the draws at train_SDRM.py:326-327 and the three
 * F.dropout masks of :100 (one per forward pass, order P, S, Q — :331, :193, :195). */
typedef struct sdrm_train_randoms {
  const float* noise;    /* [B,L] eps, ALREADY multiplied by noise_divider (:326) */
  const int64_t* t;      /* [B] timesteps in 1..T (:327) */
  const uint8_t* keep;   /* [3,B,L] 1 = element survives dropout (then scaled by 2) */
} sdrm_train_randoms;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Builds an engine for an eps-predictor SDRM(N_ITEMS=L, EMB_DIM=T, LATENT_DIM=W, n_hidden_layers=H)
 * (train_SDRM.py:86-95, :305) able to process up to max_rows rows per call, with Adam state
 * (train_SDRM.py:309) and the DDPM schedule for beta1=1e-4, beta2=0.02 (train_SDRM.py:275-276,
 * 300-303).  Parameters start at zero: call sdrm_set_params.  Envelope: 1<=L,W<=4096, 2<=T<=1020,
 * 0<=H<=16 (the reference's search space, hyperparameter_search.py:103-113, is L=W<=1000, T<=198, H<=5).
 * The device must be a gfx950 with 256 compute units (MI355X in SPX mode): the one-round launches (one work-group per CU) and
 * the work-group -> XCD mapping are sized for it, anything else is refused with SDRM_ERR_DEVICE rather than run mis-balanced. */
int sdrm_create(int L, int W, int T, int H, int max_rows, int device_id, sdrm_engine** out);
int sdrm_destroy(sdrm_engine* e);
const char* sdrm_last_error(const sdrm_engine* e);
int64_t sdrm_param_count(const sdrm_engine* e);

/* ---- schedule (train_SDRM.py:296-303; module globals b_t/a_t/ab_t, quirk Q10) ---------------- */
int sdrm_set_schedule(sdrm_engine* e, float beta1, float beta2);
/* Copies beta, alpha, alpha-bar ([T+1] each) to HOST arrays. */
int sdrm_get_schedule(const sdrm_engine* e, float* beta_host, float* alpha_host, float* alphabar_host);

/* ---- parameters / optimiser state (nn.Module.state_dict / torch.optim.Adam state) ------------ */
int sdrm_set_params(sdrm_engine* e, const float* flat, void* stream);
int sdrm_get_params(const sdrm_engine* e, float* flat, void* stream);
/* The live parameters in place: DEVICE pointer to the library's flat master vector [P] (what nn.Module.parameters() hands out in
 * the reference, hyperparameter_search.py:53: views, not copies).  Valid until sdrm_destroy.  READ-ONLY for the caller: the
 * kernels read padded / transposed / fragment-packed compute copies that only sdrm_set_params and the train step refresh. */
const float* sdrm_params_ptr(const sdrm_engine* e);
/* Gradient of the last backward, flat [P] (what autograd leaves in p.grad after :336).  If that backward was
 * given a caller buffer (`grad`), this reads from it: the buffer must still be alive. */
int sdrm_get_grads(const sdrm_engine* e, float* flat, void* stream);
/* Adam first/second moments, flat [P] each, and the global step counter (Q8). */
int sdrm_get_adam_state(const sdrm_engine* e, float* exp_avg, float* exp_avg_sq, int64_t* step_host, void* stream);
int sdrm_set_adam_state(sdrm_engine* e, const float* exp_avg, const float* exp_avg_sq, int64_t step, void* stream);
int sdrm_adam_reset(sdrm_engine* e, void* stream);

/* ---- training step, split in the three phases the user-sharded (multi-GPU) step needs --------- *
 * Replaces train_SDRM.py:326-337 for one batch of B rows (this rank's shard).
 *
 * (1) sdrm_train_forward: q_sample (:203,:328) + the three eps-net forwards P=f(x_pert,t),
 *     S=f(x0,t), Q=f(x0+0.1*eps,t) (:331,:193-195) + the rank-local partial sums of the loss
 *     (:196-198): sums[0..4] = { sum D^2, sum (R-S)^2, sum R, sum R^2, count } in float64, where
 *     R = P-x0, D = (Q-S)/mu^2 - R.  Multi-GPU callers all-reduce(sum) these 5 doubles.
 *     row0 = global index of this shard's first row (keys the Philox streams; 0 on one GPU),
 *     seed/step key the PHILOX mode; `rnd` is used in EXPLICIT mode (may be NULL in PHILOX mode).
 * (2) sdrm_train_backward: loss value (:198) and closed-form gradient seeds from the GLOBAL sums,
 *     backward through the three passes, gradient accumulation over passes and over the H shared
 *     hidden applications; writes this rank's flat gradient [P] to `grad` (device; may be NULL to
 *     keep it internal) and the global loss to `loss` (device float; may be NULL).  Multi-GPU
 *     callers all-reduce(sum) `grad`.
 * (3) sdrm_adam_step: torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-4) with
 *     coupled L2 (:309, :337), on `grad` (device, flat [P]; NULL = the internal gradient).
 *     lr is the caller's per-epoch value DIFF_LR*(1-ep/EPOCHS) (:316).
 */
int sdrm_train_forward(sdrm_engine* e, const float* x0, int B, int64_t row0, int mode,
                       const sdrm_train_randoms* rnd, uint64_t seed, uint64_t step, float noise_divider,
                       double* sums, void* stream);
int sdrm_train_backward(sdrm_engine* e, const double* sums, float* grad, float* loss, void* stream);
/* Phase 2 in two calls, for callers that overlap the gradient exchange with the backward (no reference
 * counterpart: the reference is single-device).  `begin` runs the loss seeds, the dgrad chain, the layer-0
 * weight gradient and the embedding backward and finalises the FIRST bucket of `grad` in `stream` order; `finish`
 * runs the weight gradients of the upper layers and finalises the SECOND bucket.  sdrm_grad_buckets returns the two [offset, length) ranges of the flat gradient (first: emb_layer.*,
 * dnn.0.weight, dnn.0.bias; second: everything from dnn.1 (the first PReLU slope) on).  Same `grad` in both calls.
 * sdrm_train_backward == begin + finish. */
int sdrm_train_backward_begin(sdrm_engine* e, const double* sums, float* grad, float* loss, void* stream);
int sdrm_train_backward_finish(sdrm_engine* e, float* grad, void* stream);
int sdrm_grad_buckets(const sdrm_engine* e, int64_t* first_off, int64_t* first_len, int64_t* second_off,
                      int64_t* second_len);
int sdrm_adam_step(sdrm_engine* e, const float* grad, float lr, void* stream);
/* Single-GPU convenience: (1)+(2)+(3) back to back on `stream`. */
int sdrm_train_step(sdrm_engine* e, const float* x0, int B, float lr, int mode, const sdrm_train_randoms* rnd,
                    uint64_t seed, uint64_t step, float noise_divider, float* loss, void* stream);
/* ---- multi-GPU: the user-sharded train step with its exchange inside the library ---------- *
 * No reference counterpart (the reference is single-device, train_SDRM.py:18): rows (users) of the global batch are
 * partitioned over one process per GPU; parameters, Adam state and schedule are replicated.  Per step two things cross
 * GPUs, both as RCCL all-reduce(sum) issued by the library itself: the 5 float64 loss sums (var(R) and both mse means of
 * train_SDRM.py:196-198 are over the GLOBAL batch) and the flat gradient [P] - by default in one all-reduce after the
 * backward; optionally (sdrm_hip_debug.h) in the two buckets of sdrm_grad_buckets, the first one overlapped with the upper
 * layers' weight gradients on an auxiliary stream.  Sampling needs no exchange.
 *
 * librccl is loaded at run time (dlopen "librccl.so.1", or $SDRM_RCCL_LIB); a process that already holds one - e.g. a
 * PyTorch process - shares that instance.  Either
 *   sdrm_comm_unique_id (rank 0; ship the 128 bytes to the other ranks by any channel) + sdrm_comm_init_rank (all ranks;
 *   the library owns the communicator), or
 *   sdrm_allreduce_init (adopts a ncclComm_t the caller made with the same librccl; not destroyed by the library).
 * aux_stream: a hipStream_t for the overlapped bucket (NULL = the library creates one). */
enum { SDRM_COMM_ID_BYTES = 128 };
int sdrm_comm_unique_id(void* id_host);
/* 1 when librccl can be resolved in this process (what sdrm_comm_unique_id / sdrm_comm_init_rank need), else 0: a purely local
 * check, so that the ranks of a job can agree on the exchange BEFORE any of them enters the collective ncclCommInitRank. */
int sdrm_comm_available(void);
int sdrm_comm_init_rank(sdrm_engine* e, int nranks, int rank, const void* id_host);
int sdrm_allreduce_init(sdrm_engine* e, void* rccl_comm, void* aux_stream);
int sdrm_comm_info(const sdrm_engine* e, int* nranks, int* rank);   /* nranks 0 / rank -1: no communicator */
int sdrm_comm_destroy(sdrm_engine* e);                             /* also done by sdrm_destroy */
/* sdrm_train_forward (this rank's rows, first global row row0) -> all-reduce of the loss sums -> sdrm_train_backward ->
 * all-reduce of the flat gradient -> sdrm_adam_step (two-bucket form: sdrm_train_backward_begin -> all-reduce of the first
 * bucket beside sdrm_train_backward_finish -> all-reduce of the second bucket -> sdrm_adam_step).
 * `loss` (device float, may be NULL) receives the GLOBAL loss.  With one rank it equals sdrm_train_step bit for bit. */
int sdrm_train_step_sharded(sdrm_engine* e, const float* x0, int B, int64_t row0, float lr, int mode,
                            const sdrm_train_randoms* rnd, uint64_t seed, uint64_t step, float noise_divider, float* loss,
                            void* stream);
/* Outputs of the last sdrm_train_forward: P,S,Q as [3,B,L] (parity tests). */
int sdrm_get_train_outputs(const sdrm_engine* e, float* psq, void* stream);

/* ---- plain forward: SDRM.forward(x, t) (train_SDRM.py:97-103), dropout always on (Q2) ---------- *
 * keep [n,L] (EXPLICIT) or Philox(seed, step, row0+row, col).  Used by callers that run their own
 * sampler (hyperparameter_search.py:67,78). */
int sdrm_forward(sdrm_engine* e, const float* x, const int64_t* t, int n, int mode, const uint8_t* keep,
                 uint64_t seed, uint64_t step, int64_t row0, float* out, void* stream);

/* ---- reverse sampling: sample_ddpm latent loop (train_SDRM.py:37-59) + denoise_add_noise (:20-25) *
 * Runs i = T..1 over n rows and writes the final latents x_0 to out[n,L] (the caller then applies
 * vae.decode, :49/:61).  Tj == NULL -> full resolution (:50-59).  Tj != NULL -> multi-resolution
 * (:37-48): row j runs i = Tj[j]..1; rows are independent, so this equals the reference's per-user
 * batch-1 loop (Q11).  EXPLICIT mode: xT[n,L] start noise (:38/:51), z[T+1,n,L] with z[i] the noise
 * injected at step i ALREADY multiplied by noise_divider (z[1] is ignored: no noise at i==1, Q9),
 * keep[T+1,n,L] dropout keep-masks per step.  PHILOX mode: all of those (and Tj when
 * `multires` != 0) are drawn on device; Tj_out (device int64 [n], may be NULL) receives the drawn
 * start steps. */
int sdrm_sample(sdrm_engine* e, int n, float noise_divider, int multires, int mode, const float* xT,
                const float* z, const uint8_t* keep, const int64_t* Tj, uint64_t seed, uint64_t call_id,
                int64_t row0, float* out, int64_t* Tj_out, void* stream);

/* The same loop in resumable form (sdrm_sample == begin + steps(all) + end).  A sampling call uses the parameters as
 * they are at sdrm_sample_begin (the sampler reads its own snapshot of the net; a narrow net's whole reverse loop is one launch
 * that sdrm_sample_begin itself issues): sdrm_set_params or train steps issued after the begin, between its sdrm_sample_steps
 * calls, do not change its result.  The streams of the begin / steps / end calls of one sampling call must be the same or
 * ordered by the caller.  The EXPLICIT-mode pointers must stay valid until sdrm_sample_end.  sdrm_sample_steps runs at most `count` reverse
 * steps; sdrm_sample_remaining returns the next step index i (0 = loop finished).
 * The steps are queued, not necessarily on `stream`: the engine runs row ranges of a call as chains of launches on streams of its own
 * (ordered after everything queued on `stream` at the first sdrm_sample_steps; folded back into `stream` by sdrm_sample_end and by every
 * other entry point that touches the sampler's state), so a train step queued between two sdrm_sample_steps calls may execute beside
 * the call's launches.  Neither sees the other: the call's result and the train steps' are those of the sequential order, bit for bit. */
int sdrm_sample_begin(sdrm_engine* e, int n, float noise_divider, int multires, int mode, const float* xT,
                      const float* z, const uint8_t* keep, const int64_t* Tj, uint64_t seed, uint64_t call_id,
                      int64_t row0, int64_t* Tj_out, void* stream);
int sdrm_sample_steps(sdrm_engine* e, int count, void* stream);
int sdrm_sample_remaining(const sdrm_engine* e);
int sdrm_sample_end(sdrm_engine* e, float* out, void* stream);

/* One reverse step on caller-owned state, for callers that drive the loop themselves:
 * x <- denoise_add_noise(x, i, f(x, i), z) (train_SDRM.py:56-59).  z may be NULL (= 0, the i==1 case). */
int sdrm_reverse_step(sdrm_engine* e, float* x, int n, int i, const float* z, const uint8_t* keep, void* stream);

/* q_sample alone: perturb_input(x, t, noise) (train_SDRM.py:202-203). */
int sdrm_perturb_input(sdrm_engine* e, const float* x, const int64_t* t, const float* noise, int n, float* out,
                       void* stream);

/* Pre-activations of eps-net layer `layer` (0..H) from the last sdrm_train_forward, as [3,B,W]
 * (pass order P,S,Q).  Parity tests use their signs to evaluate the oracle with the same PReLU
 * derivative choice at pre-activations that are zero within fp32 rounding (DESIGN.md "kink flips"). */
int sdrm_get_preacts(const sdrm_engine* e, int layer, float* out, void* stream);

/* ---- introspection for bench.py / profiling ---------------------------------------------------- */
/* Per-kernel timing with HIP events recorded on the launch stream around every GEMM launch.
 * sdrm_profile_begin enables it (capacity = max launches recorded), sdrm_profile_end synchronises the
 * stream and aggregates per kernel class; sdrm_profile_get returns, for class `cls`
 * (0..sdrm_profile_classes()-1), the summed duration [ms], the launch count and the summed
 * ALGORITHMIC flops (2*rows*N*K on the unpadded dims).  Adds two event records per launch, so it is
 * used in a separate pass, never inside the throughput-timed region. */
int sdrm_profile_begin(sdrm_engine* e, int capacity);
/* Restricts the bracketing to ONE kernel class (cls < 0: all classes again).  Every bracketed launch puts two marker packets on
 * the stream, and the markers of neighbouring launches inflate each other's intervals (+3..5 us per launch, +10 us on the
 * largest one): bench.py finds the dominant class with all classes bracketed, then times that class alone. */
int sdrm_profile_only(sdrm_engine* e, int cls);
int sdrm_profile_end(sdrm_engine* e, void* stream);
int sdrm_profile_classes(void);
const char* sdrm_profile_name(int cls);
int sdrm_profile_get(const sdrm_engine* e, int cls, double* total_ms, int64_t* launches, double* flops);

/* Kernel launches issued through this handle since sdrm_create (memcpy / memset nodes not counted): bench.py reports
 * launches per step beside the roofline of the latency-bound configurations. */
int64_t sdrm_launch_count(const sdrm_engine* e);
/* Name of the GEMM kernel variant family in use and tile geometry, as a static string. */
const char* sdrm_build_info(void);
/* Hex SHA-256 of the sources this binary was compiled from (sdrm_amd/csrc/ and include/, as sdrm_amd/_build.py hashes
 * them).  The Python loader refuses, or rebuilds, a library whose hash differs from the sources next to it. */
const char* sdrm_source_hash(void);
/* Sparse batch feed (reference: dataloaders.py:46-79 builds a COO tensor per batch on the host, train_SDRM.py:323
 * densifies it before vae.encode).  The CSR matrix of the whole feed [n_rows, n_items] stays on the device (int64
 * indptr, int32 column indices, float32 data or null for all-ones); out [b, n_items] float32 receives the dense rows
 * rows[0..b) (device int64 array; the caller's epoch permutation) or, when rows is null, rows row0 .. row0+b-1
 * (checked on the host against n_rows).  The caller's CSR is not trusted with the address of a store: on the device every
 * row id is checked against [0, n_rows), every indptr pair for order, every column index against [0, n_items); an offending
 * row / entry is left zero and recorded in the handle's feed status word, which sdrm_feed_status reports. */
int sdrm_csr_rows_to_dense(sdrm_engine* e, const int64_t* indptr, const int32_t* indices, const float* data, int64_t n_rows,
                           const int64_t* rows, int64_t row0, int b, int n_items, float* out, void* stream);
/* SDRM_OK, or SDRM_ERR_ARG (message: what was out of range) if any sdrm_csr_rows_to_dense launch since the last call met an
 * out-of-range row id, indptr pair or column index; clears the record.  Synchronises `stream` (call it once per epoch, not per
 * batch). */
int sdrm_feed_status(sdrm_engine* e, void* stream);

/* Equal-sparsity binarisation of sampled data on the device (reference: main.py:177-180,
 *   threshold = np.quantile(M.flatten(), SPARSITY); M_equal_sparsity = (M >= threshold)):
 * x [n] float32 (the flattened [users, items] matrix, 16-byte aligned), q in [0,1].  threshold (device float*, may
 * be null) receives exactly what np.quantile (numpy 2.x, method "linear": float32 virtual index, float32 lerp
 * between the two neighbouring order statistics) returns for a float32 array; out (device uint8[n], 4-byte
 * aligned, may be null) receives x >= threshold.  Inputs must not contain NaN. */
int sdrm_equal_sparsity(sdrm_engine* e, const float* x, int64_t n, double q, uint8_t* out, float* threshold, void* stream);

/* The VAE decode hook on the device (reference: train_SDRM.py:212-214 `decoder = Linear(latent, hidden) -> Tanh ->
 * Linear(hidden, N_ITEMS)`, :252-254 `VAE.decode`, called by sample_ddpm at :49 / :61 on the sampled latents): two launches of
 * the engine's fp32 MFMA GEMM with the bias + tanh / bias epilogues.  The struct holds DEVICE pointers to the four nn.Linear
 * tensors as PyTorch stores them (weight [out, in] row-major).  z [n, latent] float32 -> out [n, n_items] float32.  Scratch
 * (padded copies of z and of the weights, the hidden activations) is library-owned and grows on demand - this is a
 * once-per-sampling call, not a step call. */
typedef struct sdrm_vae_decoder {
  const float* w1; const float* b1;   /* decoder[0]: [hidden, latent], [hidden] */
  const float* w2; const float* b2;   /* decoder[2]: [n_items, hidden], [n_items] */
  int latent, hidden, n_items;
} sdrm_vae_decoder;
int sdrm_vae_decode(sdrm_engine* e, const sdrm_vae_decoder* dec, const float* z, int n, float* out, void* stream);
/* Recall@k and NDCG@k of a score matrix against held-out interactions (reference: utilities.py:116-171,
 * mask_training_examples + recall_at_k_batch + NDCG_binary_at_k_batch, as svd_benchmark.py:58-66 chains them).
 * scores [U, I] float32 row-major; held_* / train_* are CSR index arrays over the same U rows (int64 indptr [U+1],
 * int32 column indices; every stored entry counts as an interaction); train_* may both be null (no masking,
 * otherwise those items score -inf).  ks_host: nk <= 8 cut-offs (HOST array, each 1 <= k <= min(128, I)).
 * tp [kmax] = 1/log2(r+2) and idcg [kmax+1] = sum(tp[:m]) are float64 DEVICE tables computed by the caller with numpy
 * so that the constants are the reference's own.  recall, ndcg: [nk, U] float64 (nan where a user has nothing held
 * out, like the reference's 0/0).  Ties in the scores rank towards the lower item index (the reference's
 * argpartition leaves them unspecified). */
int sdrm_rank_metrics(sdrm_engine* e, const float* scores, int U, int I, const int64_t* held_indptr,
                      const int32_t* held_indices, const int64_t* train_indptr, const int32_t* train_indices,
                      const int32_t* ks_host, int nk, const double* tp, const double* idcg, double* recall, double* ndcg,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDRM_HIP_H */
