/*
 * sdrm_hip_debug.h — test and tuning hooks of libsdrm_hip.so.  NOT part of the drop-in boundary (that is
 * sdrm_hip.h): nothing here has a counterpart in the reference, a production caller never needs it, and the
 * signatures may change between rounds.  Used by tests/ (to force every kernel variant through the same parity
 * checks, whatever the size thresholds currently are) and by tools/ (tile sweeps).
 *
 * Every setter acts on ONE handle: tile / path selection lives in the engine (sdrm_hip.h "Conventions"), so
 * forcing a variant on one engine never changes what another engine in the same process launches.
 */
#ifndef SDRM_HIP_DEBUG_H
#define SDRM_HIP_DEBUG_H

#include "sdrm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What sdrm_create checks of the device: SDRM_OK for a gfx950 (architecture string as hipDeviceProp_t::gcnArchName gives it,
 * e.g. "gfx950:sramecc+:xnack-") with 256 compute units, SDRM_ERR_DEVICE otherwise.  Pure function: callable without a GPU. */
int sdrm_debug_device_check(const char* gcn_arch, int compute_units);

/* Tile shapes of the MFMA GEMM template, as numbered by sdrm_debug_set_tile / the `cfg` arguments below:
 *   0 = 64x64x16 (default), 1 = 64x64x32, 2 = 64x128x16, 3 = 128x128x16 on v_mfma_f32_32x32x2_f32,
 *   4 = 32x32x32 on v_mfma_f32_16x16x4_f32. */
enum { SDRM_TILE_AUTO = -1, SDRM_TILE_COUNT = 5 };

/* Forces the GEMM tile shape of every launch of this engine (SDRM_TILE_AUTO = automatic: 64x64x16, or 32x32x32 for NT
 * launches of at most `nt32` rows, see sdrm_debug_set_nt32_rows); also env SDRM_TILE, read by sdrm_create.  Refused
 * (SDRM_ERR_STATE) between sdrm_train_backward_begin and _finish and inside a sampling call; a train forward still
 * waiting for its backward is dropped (the slope partial layout of a backward is fixed by the tile of its forward). */
int sdrm_debug_set_tile(sdrm_engine* e, int cfg);
/* Row thresholds of the automatic tile choice: NT launches of at most max_rows rows (sampling, plain forward; default
 * 4096) / max_rows_train stacked rows (train step; default 8192) run on the 32x32x32 tile.  A negative value leaves
 * that threshold alone.  Also env SDRM_NT32_MAX_ROWS / SDRM_NT32_MAX_ROWS_TRAIN. */
int sdrm_debug_set_nt32_rows(sdrm_engine* e, int max_rows, int max_rows_train);
/* The LDS-resident kernels used when the padded widths are <= 64 (csrc/skinny.h, csrc/skinny_step.h, csrc/skinny_fwd4.h):
 * 1 (default) on, the train forward owning 4 users per work-group (4x4x1 MFMAs); 2 on, with the 16-user train forward (what shapes
 * whose LDS image does not fit take anyway); 0 off: narrow nets go through the general per-layer GEMM path. */
int sdrm_debug_set_skinny(sdrm_engine* e, int on);
/* Row-owned train forward (csrc/rowchain.h: staging, every layer and the loss partial sums of a 96-row group of stacked rows in
 * ONE work-group per CU; nets with L == W and a padded width of 128..352): 0 never, 1 (default) when the batch fills whole
 * rounds of the chip (one round: at least 154 groups of 32 users; more: the last round at least five sixths full), 2 whenever the net allows
 * (tests); also env SDRM_ROWCHAIN.  A forced tile (sdrm_debug_set_tile) switches it off.  Results agree with the per-layer path
 * to fp32 summation order; the stacked rows of that step are in the GROUPED order (elementwise.h), which sdrm_get_train_outputs
 * / sdrm_get_preacts undo.  Refused between sdrm_train_backward_begin and _finish; drops a pending train forward. */
int sdrm_debug_set_rowchain(sdrm_engine* e, int mode);
/* The row-owned train step on 48-row work-groups (csrc/rows48.h: the P, S, Q rows of 16 users per work-group through staging,
 * every layer and the loss sums, then through the loss seeds and every layer's input gradient; the nets of the row-owned forward):
 * 0 never, 1 (default) when the batch's 16-user groups fill most of one round of the chip (160..256 groups: 2545..4096 users) and
 * the 96-row kernels do not take the batch, 2 whenever the net allows (tests); also env SDRM_ROWS48.  sdrm_debug_set_rowchain(e, 2)
 * and a forced tile take precedence.  The stacked rows of such a step are grouped by 16 users (elementwise.h).  Refused between
 * sdrm_train_backward_begin and _finish; drops a pending train forward. */
int sdrm_debug_set_rows48(sdrm_engine* e, int mode);
/* Column-split row groups of that step (csrc/rows48.h): G work-groups of ONE XCD share a 48-row group - each owns 1 / G of every
 * layer's output columns and pulls the rest of the next layer's input through that XCD's L2 (work-group-scope atomics + sc1 loads,
 * no device-scope fence) - so that batches of at most 2048 users fill the chip in 6 launches instead of 11: 0 never, 1 (default) by
 * size (1281..2048 users: 2 work-groups per group), 2 / 4 that many whenever groups x G fit the 256 CUs (every
 * work-group of such a launch must be resident at once); also env SDRM_ROWS48_SPLIT.  Needs the chip's block -> XCD mapping
 * (checked by sdrm_create: sdrm_debug_rows48_split_available).  A hand-shake that times out (30 ms) is reported by the next train
 * call on the handle as SDRM_ERR_HIP and switches the path off for the handle. */
int sdrm_debug_set_rows48_split(sdrm_engine* e, int mode);
int sdrm_debug_rows48_split_available(const sdrm_engine* e);
/* The shared-tile form of the 48-row kernels (csrc/rows48.h; nets whose padded width is 4 q + 2 column tiles of 16 - 352 is 22 -, one
 * work-group per row group): the four waves own q tiles each and the waves of a pair split the K-steps of one more, instead of
 * multiplying two tiles of zeros in the last wave: 1 (default) on, 0 the plain form; also env SDRM_ROWS48_SHARE.  The same sums in
 * another order for the shared tiles' columns (two partial sums over alternate K-steps). */
int sdrm_debug_set_rows48_share(sdrm_engine* e, int on);
/* Reverse-sampling steps without kernel boundaries between the layers (csrc/sample_persist.h: one launch per sdrm_sample_steps call
 * runs its `count` steps - every layer on the 32x32 tile, the reverse update in the out layer's epilogue - with the column tiles of
 * a row tile synchronised through ONE XCD's L2): full-resolution PHILOX sampling of a net with L == W; 0 never, 1 (default) for at
 * most 352 rows (one work-group per CU: measured 12.1 against 16.5 us per step at 339 rows; at 679 rows - the 8-GPU shard of the 5429
 * sampled users - the fullest XCDs hold 33 work-groups for 32 CUs and it ties, 16.0 against 16.4), 2 whenever every work-group of the launch is resident at once
 * (at most two 32x32 tiles per CU on the fullest XCD); also env SDRM_SAMPLE_PERSIST.  Results are those of the per-layer path with
 * the fused reverse update (same tile body, same epilogue).  Refused inside a sampling call. */
int sdrm_debug_set_sample_persist(sdrm_engine* e, int mode);
/* Fault injection for that path: the NEXT column-split launch waits for a count its group never reaches (its base is moved up by
 * `skew`), so every work-group of it runs into the 30 ms bound of its first hand-shake, raises the abort word and returns; the
 * next train call on the handle then reports SDRM_ERR_HIP and the handle continues on the per-layer path. */
int sdrm_debug_split_skew(sdrm_engine* e, unsigned skew);
/* Strip-owned weight gradients (csrc/wgrad2.h: every weight gradient of a step in one balanced round of one work-group per CU,
 * bias gradients from the ones column of the layer inputs) behind the row-owned forward: 1 (default) on, 0 the batched 64x64-tile
 * split-K launch; also env SDRM_WGRAD_STRIPS.  Takes effect with the next backward. */
int sdrm_debug_set_wgrad_strips(sdrm_engine* e, int on);
/* Row-owned input gradients (csrc/dgrad_rows.h: one work-group per CU owns 96 stacked rows and every column of a layer's input
 * gradient, operands straight from global memory, PReLU' and the slope partial sums on 16-byte quads) behind the row-owned
 * forward: 1 (default) the loss value, the gradient seeds and every layer's dgrad in ONE launch (k_dgrad_chain: rows never meet, a
 * work-group runs down the chain on its own), 2 k_loss_seed + one k_dgrad_rows launch per layer, 0 k_loss_seed + the 64x64-tile
 * launches; also env SDRM_DGRAD_ROWS.  Drops a pending train forward (a row-owned forward stores no pre-activations when these
 * kernels follow: they read the activations). */
int sdrm_debug_set_dgrad_rows(sdrm_engine* e, int mode);
/* 1 when this engine's shape qualifies for the row-owned forward (its fragment-packed weight copies exist). */
int sdrm_debug_rowchain_available(const sdrm_engine* e);
/* Fusion of the DDPM reverse update into the out-layer GEMM epilogue (full-resolution sampling with on-device Philox;
 * every other case uses the stand-alone k_reverse_update): 0 never, 1 (default) for launches of at most 4096 rows - the
 * shards of a multi-GPU run, which run on the 32x32 tile where one launch less per reverse step is worth more than the
 * epilogue's Philox work - 2 always; also env SDRM_FUSE_REV.  Results are identical up to the rounding of the update
 * arithmetic. */
int sdrm_debug_set_fused_reverse(sdrm_engine* e, int mode);
/* Row chains of a sampling call (csrc/sdrm_hip.hip: independent row ranges run on separate HIP streams so that one
 * chain's launch gaps are filled by another's kernels): -1 = by size (default: sdrm_debug_chains), 1..4 forced; also env SDRM_CHAINS.
 * Rows are independent and randoms are keyed by row: every chain count computes the same call; the tile and the fusion of the reverse
 * update go by a chain's row count, so the last bits may differ between chain counts (never between two runs of one). */
int sdrm_debug_set_chains(sdrm_engine* e, int chains);
/* The number of row chains of the sampling call in progress (or of the last one); by size: two once the call has 2560 x 352 elements
 * per layer (ML-1M: n >= 2560 rows).  While an event profile is recorded (sdrm_profile_begin) the chains run one after the other.
 * How chains and train steps queued between sampling steps share the chip (csrc/sdrm_hip.hip: chains_for, hold_chains, hold_point) has two
 * environment switches for A/B runs, read at sdrm_create: SDRM_DETACH=0 keeps the one chain of a small call on the caller's stream
 * (default: on an auxiliary stream, beside train steps of the per-layer path), SDRM_HOLD_EARLY=0 makes the chains wait for the end of a
 * row-owned train step (default: for its weight gradients; the tail runs beside them). */
int sdrm_debug_chains(const sdrm_engine* e);

/* Gradient all-reduces of sdrm_train_step_sharded: 1 (default) = one all-reduce of the whole flat gradient after the one-call
 * backward; 2 = the two buckets of sdrm_grad_buckets, the first overlapped with the upper layers' weight gradients on the
 * auxiliary stream (pays when the first bucket's all-reduce takes longer than the ~40 us the hand-offs cost); also env
 * SDRM_AR_BUCKETS.  The result is the same bit for bit. */
int sdrm_debug_set_gradient_buckets(sdrm_engine* e, int buckets);

/* The on-device generator's draws as the staging kernels consume them (csrc/philox.h): for rows row0 .. row0 + rows - 1 and column
 * quads 0 .. quads - 1, one Philox4x32-10 call keyed by (seed, step, purpose) each: normals [rows, 4 * quads] float32 (two
 * Box-Muller pairs per call) and, when lowbits is not NULL, [rows, 4 * quads] uint8 holding bits 0..2 of the call's four words
 * (the dropout keep bits of the three passes of a train step).  purpose: 1 = train step elements, 3 = sampling start noise,
 * 4 | (i << 8) = reverse step i.  For the statistical tests of the generator (tests/test_philox_stats.py). */
int sdrm_debug_philox_draws(sdrm_engine* e, uint64_t seed, uint32_t purpose, uint32_t step, int64_t row0, int rows, int quads,
                            float* normals, uint8_t* lowbits, void* stream);

/* The engine's ncclComm_t (NULL without one): lets a test hand a communicator made elsewhere to sdrm_allreduce_init. */
void* sdrm_debug_comm_handle(const sdrm_engine* e);

/* Host-side planning of a split-K weight-gradient launch with the default settings (no device work): for a reduction
 * over `rows` stacked rows into ONE [n_out, k_in] gradient, the number of K-slices (slabs) and the rows per slice the
 * engine would use (a step's gradients are planned together: one slice count for all - one round of the resident
 * work-groups when that gives eight slices or more of at most 1024 rows, else about 2.2 rounds).  Invariants: rows_per_slice is a multiple of 32, slices * rows_per_slice >= rows > (slices - 1) *
 * rows_per_slice, slices <= 64, at least four K-steps of 32 rows per slice when the rows allow, and slices * tiles does not
 * exceed the 2816 work-group target unless a single slice already does. */
int sdrm_debug_plan_wgrad(int rows, int n_out, int k_in, int* slices, int* rows_per_slice);

/* C[M,N] = A[M,K] * B^T (variant 0, B is [N,K]), A * B (variant 1, B is [K,N]), A^T * B (variant 2, A is [K,M], B is
 * [K,N]) through the same MFMA kernel the engine uses, on tile shape `cfg` (0..4).  All dims must be multiples of 32.
 * Stages the operands in zero-padded scratch and synchronises. */
int sdrm_debug_gemm(int variant, int cfg, const float* A, const float* B, float* C, int M, int N, int K, void* stream);
/* Same kernel timed: `reps` launches on zero-filled scratch operands, mean microseconds per launch by HIP events on
 * `stream` (tools/gemm_tune.py). */
int sdrm_debug_gemm_time(int variant, int cfg, int M, int N, int K, int reps, float* us_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDRM_HIP_DEBUG_H */
