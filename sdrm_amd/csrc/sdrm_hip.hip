// libsdrm_hip.so — C ABI (include/sdrm_hip.h) over the gfx950 kernels in gemm.h / elementwise.h.
// Host logic only: buffer ownership, launch orchestration of the train step / sampler, error codes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/sdrm_hip.h"
#include "../../include/sdrm_hip_debug.h"
#include "decode.h"
#include "elementwise.h"
#include "exchange.h"
#include "feed.h"
#include "gemm.h"
#include "rank.h"
#include "rowchain.h"
#include "rows48.h"
#include "sample_persist.h"
#include "select.h"
#include "skinny.h"
#include "skinny_step.h"
#include "skinny_fwd4.h"
#include "tail.h"
#include "dgrad_rows.h"
#include "wgrad2.h"

using namespace sdrm;

namespace {

constexpr int BM = 64;              // row padding granule = rows of the default tile (the 128-row alternates read into
                                    // the slack every buffer carries, and write rows nobody reads)
constexpr int S_MAX = 64;          // max split-K slabs per weight-gradient GEMM
constexpr int LOSS_BLOCKS = 256;

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

// Tile / path selection of ONE engine: read from the environment once, in sdrm_create, changed afterwards only by the
// sdrm_debug_* setters of that handle (include/sdrm_hip_debug.h).  Nothing here is process-global: two engines on two
// threads never see each other's settings.
struct Tuning {
  int force_cfg = -1;            // SDRM_TILE: -1 = automatic, 0..4 forced tile shape
  int chains = -1;               // SDRM_CHAINS: sampler row chains, -1 = by size, 1..4 forced
  int detach = 1;                // SDRM_DETACH: the one chain of a small sampling call runs on an auxiliary stream (chains_for)
  int hold_early = 1;            // SDRM_HOLD_EARLY: the chains of a sampling call wait for the weight gradients of a train step queued
                                 // between two of its steps (1) or for the whole step (0): hold_point
  int fuse_rev = 1;              // SDRM_FUSE_REV: reverse update fused into the out-layer GEMM epilogue (full-resolution PHILOX
                                 // sampling): 0 never, 1 for launches of at most FUSE_REV_MAX_ROWS rows, 2 always
  int skinny = 1;                // (2: with the 16-user train forward instead of the 4-user one) LDS-resident kernels for nets with padded widths <= 64 (persistent sampler, fused train
                                 // forward and dgrad chain)
  int nt32_max_rows = 4096;      // SDRM_NT32_MAX_ROWS: sampling / plain-forward NT launches of at most this many rows use the 32x32 tile
  int nt32_max_rows_train = 8192;  // SDRM_NT32_MAX_ROWS_TRAIN: the same for the train step's launches (stacked rows = 3 x batch):
                                 // 6144 stacked rows (a 4-GPU shard of the 8192 batch) 234 -> 226 us per step on the 32x32 tile,
                                 // 12288 a tie; the sampling launch of 5429 rows is faster on 64x64 (18.5 k vs 17.5 k steps/s)
  int wgrad_blocks = 2816;       // SDRM_WGRAD_BLOCKS: work-groups the batched weight-gradient launch of a step aims for (2.2 rounds of 5 per CU) ...
  int wgrad_round = 1280;        // SDRM_WGRAD_ROUND: ... unless ONE round (this many work-groups) already gives eight slices or more
  int wgrad_slices = 0;          // SDRM_WGRAD_SLICES: > 0 forces the K-slice count of every weight-gradient problem (tuning aid)
  int ar_buckets = 1;            // SDRM_AR_BUCKETS: gradient all-reduces of sdrm_train_step_sharded: 1 (after the whole backward) or 2 (overlapped)
  int dgrad_rows = 1;            // SDRM_DGRAD_ROWS: row-owned input gradients (csrc/dgrad_rows.h) behind the row-owned forward: 0 off,
                                 // 1 the whole chain (loss seeds + every layer) in one launch, 2 k_loss_seed + one launch per layer
  int strips = 1;                // SDRM_WGRAD_STRIPS: strip-owned weight gradients (csrc/wgrad2.h) behind the row-owned forward: 0 off
  int rows48 = 1;                // SDRM_ROWS48: the row-owned train step on 48-row work-groups (csrc/rows48.h: 16 users' P, S, Q rows; the
                                 // same nets as rowchain) for batches that do not fill the chip with 96-row work-groups: 0 never, 1 when
                                 // the batch's 16-user groups fill most of one round of the chip (see use_rows48), 2 whenever the net allows
  int rows48_share = 7;          // (bit 0: the forward, bit 1: the dgrad chain, bit 2: the forward's deferred-store sweep) SDRM_ROWS48_SHARE: the shared-tile form of the 48-row kernels (csrc/rows48.h: widths of 4 q + 2 column tiles - 352 is
                                 // 22 - no wave multiplies tiles beyond the layer; the waves of a pair split one tile's K-steps): 1 on, 0 the plain form
  int smp_persist = 1;           // SDRM_SAMPLE_PERSIST: reverse steps without kernel boundaries between the layers (csrc/sample_persist.h; full
                                 // resolution, PHILOX, L == W, one row chain): 0 never, 1 for at most SMP_PERSIST_MAX_ROWS (352) rows, 2 whenever it fits
  int split = 1;                 // SDRM_ROWS48_SPLIT: column-split row groups of that step (G work-groups of one XCD share a 48-row group and
                                 // exchange the activations through that XCD's L2, csrc/rows48.h) for batches of at most 2048 users:
                                 // 0 never, 1 by size (see rows48_parts), 2 / 4: that many work-groups per group whenever the grid fits the chip
  int rowchain = 1;              // SDRM_ROWCHAIN: row-owned train forward (csrc/rowchain.h) for nets with L == W, padded width 128..352:
                                 // 0 never, 1 when the batch fills whole rounds of one 96-row work-group per CU, 2 whenever the net allows
};

struct sdrm_engine {
  int L, W, T, H, max_rows, device;
  Tuning tune;
  int LP, WP, TP, K0, MPmax;
  int64_t P;
  int64_t off_we, off_be, off_w0, off_b0, off_a0, off_wh, off_bh, off_ah, off_wo, off_bo;
  // flat master copies
  float *p = nullptr, *m = nullptr, *v = nullptr, *g = nullptr;
  // padded compute copies
  float *W0c = nullptr, *b0c = nullptr, *Whc = nullptr, *bhc = nullptr, *Woc = nullptr, *boc = nullptr;
  float *WhcT = nullptr, *WocT = nullptr;   // transposed copies [in][out]: dgrad is then an NT GEMM like the forwards
  float *WhfT = nullptr, *WofT = nullptr;                 // the same of the transposes ([k = out][n = in]) for the row-owned dgrads
  float *W0f = nullptr, *Whf = nullptr, *Wof = nullptr;   // fragment-packed copies [WP/16][WP/16][64][4] for the row-owned forward
                                                          // (layer 0: the latent columns only); null when the net does not qualify
  bool cur_grouped = false;          // stacked row order of the last train_forward (elementwise.h: stacked_row)
  bool cur_sk = false;               // ... grouped by 16 users (the narrow nets' step, csrc/skinny_step.h)
  bool cur_g16 = false;              // ... grouped by 16 users by the 48-row row-owned forward (csrc/rows48.h)
  int cur_rows = 0;                  // stacked rows the last train_forward really wrote (whole groups; cur_MP rounds them up to the tile)
  bool cur_skip_pre = false;         // that forward was told not to store the pre-activations (the row-owned dgrads read the activations)
  int cur_parts = 1;                 // work-groups per row group of that forward (csrc/rows48.h: 1, or 2 / 4 column-split)
  // column-split row groups: hand-shake counters of the forward / of the dgrad chain [256 groups][32], never reset while the
  // launch geometry stays the same (xgeo); the abort word lives in host-visible memory (xabort_host / its device alias)
  unsigned *xcntF = nullptr, *xcntC = nullptr;
  unsigned *xabort_host = nullptr, *xabort_dev = nullptr;
  uint32_t xepochF = 0, xepochC = 0;
  int xgeoF = 0, xgeoC = 0;           // (parts << 16 | groups) of the launches the counters have counted
  bool xcd_ok = false;               // the probe launch of sdrm_create found work-group b on XCD b & 7
  unsigned* xcntS = nullptr;         // the persistent sampler's row-tile counters [256][32] and what they stand at (phases x column tiles)
  uint32_t xphaseS = 0;
  unsigned xskew = 0;                // test hook (sdrm_debug_split_skew): added once to the next split launch's counter base
  int cur_sk_np = 0;                 // ... and the loss partials its forward left (G, or 4 G: csrc/skinny_fwd4.h)
  bool tables_fresh = false;         // B0tab / the C0^T columns of W0c belong to the current parameters
  float* act = nullptr;              // activations prelu(pre[k]) [H+1][MPmax][WP], written by the row-owned forward beside pre[k]:
                                     // the weight gradients of that step read their operand without PReLU on load
  bool cur_act = false;              // the last train_forward stored them
  int ones_col = -1;                 // pad column round_up(W, 4) < WP of the layer inputs that carries 1.0 (the padded biases put it there,
                                     // the row-owned forward's staging into U): column ones_col of a weight-gradient slab is then the
                                     // bias gradient (csrc/wgrad2.h); -1: none
  bool bwd_strips = false;           // this backward's weight gradients: the strip-owned launch (bias gradients in slab column ones_col)
  float *temb = nullptr, *tembP = nullptr, *B0tab = nullptr;   // time-embedding table [T+1][T]; the same with rows padded to TP (the
                                                               // trailing columns of the train step's layer-0 operand); b0 + C0[t]
  float *sched = nullptr;  // [8][T+1]: beta alpha alphabar sqrt_ab one_minus_ab
  float *Us = nullptr;               // sampler's own layer-0 input [rows][LP] (survives train steps between sample_steps calls)
  float *smp_pre = nullptr, *smp_Y = nullptr;   // ... and its own layer buffers / eps-net output: a train step between two sampling steps
                                     // shares nothing with the call in progress (sdrm_train_forward: no join of the row chains)
  float *U = nullptr, *pre = nullptr, *Y = nullptr, *dY = nullptr, *dA = nullptr, *X = nullptr;
  float *slab0 = nullptr, *slabH = nullptr, *slabO = nullptr, *db0s = nullptr, *dbHs = nullptr, *dbOs = nullptr;
  float *alpha_part = nullptr;
  int alpha_part_stride = 0;
  double *loss_part = nullptr, *sums = nullptr;
  float *WeP = nullptr, *W0eP = nullptr;    // narrow nets with T <= 128: padded copies of emb_layer.weight [TPe][TPe] and of W0e [WP][TPe]
  int TPe = 0;                              // (TPe = T rounded up to 16): the forward makes its own rows of the time-embedding table
  float *Mred = nullptr, *snap = nullptr;   // the tail's hand-over to its second launch (csrc/tail.h): M [W][TP]; We | be | W0e before the update
  int *tdev = nullptr;
  int64_t *Tj_dev = nullptr;
  int *rowid_dev = nullptr;
  const float* grad_src = nullptr;   // where the last backward wrote the flat gradient (internal g or the caller's buffer)
  float *rev_dev = nullptr;          // [3][T+1] reverse-step coefficients c1, sqrt(alpha), sqrt(beta)
  SelectState* sel = nullptr;        // radix-select workspace of sdrm_equal_sparsity
  float* one_dev = nullptr;          // 1.0f (identity PReLU slope for layer 0 inside the batched weight-gradient launch)
  unsigned* feed_flag = nullptr;     // status word of the sparse batch feed (csrc/feed.h: out-of-range row ids / column indices)
  std::vector<int> smp_nact, smp_perm;
  std::vector<int64_t> smp_tj_sorted, smp_tj_orig;
  std::vector<float> h_beta, h_alpha, h_alphabar;
  int64_t adam_t = 0;
  // state of the last train_forward
  int cur_B = 0, cur_MP = 0;
  const float* cur_x0 = nullptr;
  bool fwd_done = false;
  int last_S = 1, last_dgrad_blocks = 0;
  bool bwd_begun = false;
  bool fold_sums = false;            // sdrm_train_step: the seed kernel folds the loss partials itself (no k_loss_sums launch)
  int bwd_S0 = 1, bwd_SH = 1, bwd_SO = 1, bwd_kc0 = 0, bwd_kcH = 0, bwd_kcO = 0, bwd_dgrad_blocks = 0;
  int bwd_hidden_apps = 0;           // slab sets of the shared hidden layer's weight gradient per K-slice: H (one per application), or 1 (already summed)
  int bwd_cfg_w = 0;                 // tile of the split-K launches, fixed by backward_chain for the whole backward
  // grow-only scratch of sdrm_vae_decode (padded latents / weights / hidden activations)
  float* dec_buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t dec_cap[6] = {0, 0, 0, 0, 0, 0};
  Exchange xch;                      // RCCL communicator of the user-sharded step (sdrm_comm_init_rank / sdrm_allreduce_init)
  mutable int64_t n_launches = 0;    // kernel launches issued through this handle since sdrm_create
  // The sampler's own copy of everything it reads of the net (padded weights, biases, the folded bias table b0 + C0[i], the two
  // PReLU slopes), taken by sdrm_sample_begin: a sampling call is a function of the parameters at its begin, whatever
  // sdrm_set_params or train steps do between its sdrm_sample_steps calls (the persistent narrow-net sampler, one launch
  // for the whole loop, is issued by sdrm_sample_begin itself and so reads the parameters of that moment).
  float* smp_w = nullptr;
  size_t smp_off[7] = {0, 0, 0, 0, 0, 0, 0};   // W0c, Whc, Woc, bhc, boc, B0tab, slopes
  struct SampleStateT {
    bool active; int n, MP, multires, mode, i_next; float nd; const float* z; const uint8_t* keep;
    uint64_t seed, call_id; int64_t row0;
    const float* xT; bool skinny; bool skinny_launched; int i_start;
  } smp = {false, 0, 0, 0, 0, 0, 1.f, nullptr, nullptr, 0, 0, 0, nullptr, false, false, 0};
  // sampler row chains: independent row ranges of one sampling call run on their own streams so that one
  // chain's launch gaps / prologues / tails are filled by another chain's kernels (chain 0 = caller's stream)
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_hold = nullptr;      // behind the weight gradients of a train step that runs between two sampling steps (hold_point)
  int n_chains = 1, chain_chunk = 0;
  int n_aux = 0;                     // chains on auxiliary streams: n_chains - 1 (the first chain on the caller's stream), or the one chain
                                     // of a small call, detached (chains_for)
  bool detach_armed = false;         // a train step was queued beside a small call on the caller's stream: ev_fork marks the point behind the
                                     // call's last step, the next sdrm_sample_steps moves the chain to aux[0] (sdrm_train_forward)
  bool hold_needed = false;          // the last train forward was a row-owned one (one work-group per CU): chains wait for such a step (hold_chains)
  bool chains_pending = false;
  bool hold_recorded = false;        // ... and its ev_hold was recorded behind that step's weight gradients (hold_point)
  bool train_since_sample = false;   // a train step was queued since the last sampling step: the chains' next launches wait for it (hold_chains)
  // event profiling (bench only)
  bool prof_on = false;
  int prof_cap = 0;
  int prof_only = -1;                   // >= 0: only launches of this class are bracketed (sdrm_profile_only)
  std::vector<hipEvent_t> prof_ev;      // 2 per recorded launch
  std::vector<int> prof_cls;
  std::vector<double> prof_flops;
  double prof_ms[16] = {0};
  double prof_fl[16] = {0};
  int64_t prof_n[16] = {0};
  std::string err;
};

typedef sdrm_engine::SampleStateT SampleState;

enum ProfClass { PC_FWD_L0 = 0, PC_FWD_HIDDEN, PC_FWD_OUT, PC_DGRAD, PC_WGRAD, PC_WGRAD_L0, PC_SMP_L0, PC_SMP_HIDDEN,
                 PC_SMP_OUT, PC_ROW_FWD, PC_WGRAD_STRIPS, PC_DGRAD_ROWS, PC_SMP_PERSIST, PC_COUNT };
// the template arguments are <LOADA,LOADB,XFA,XFB,EPI> of gemm_kernel (what rocprofv3 prints after the tile type)
static const char* kProfNames[PC_COUNT] = {
    "train: gemm_kernel<0,0,0,0,9> fwd layer0 (row-table bias)", "train: gemm_kernel<0,0,1,0,0> fwd hidden (prelu-in, bias)",
    "train: gemm_kernel<0,0,1,0,1> fwd out (prelu-in, tanh)",
    "train: gemm_kernel<0,0,0,0,3> dgrad (prelu' epilogue)",
    "train: gemm_batch_kernel<1,1,0,1,4> wgrad of all layers in one launch (prelu-in, split-K slabs)",
    "train: gemm_kernel<1,1,0,0,4> wgrad layer0 (split-K slabs)",
    "sample: gemm_kernel<0,0,0,0,10> fwd layer0 (bias table, prelu epilogue)", "sample: gemm_kernel<0,0,0,0,10> fwd hidden (bias, prelu epilogue)",
    "sample: gemm_kernel<0,0,0,0,1> fwd out (tanh)",
    "train: k_row_fwd row-owned forward (staging + all layers + loss partial sums, one work-group per CU)",
    "train: k_wgrad_strips weight gradients of all layers (strip-owned split-K, one work-group per CU)",
    "train: k_dgrad_chain / k_dgrad_rows input gradients (row-owned, prelu' epilogue, one work-group per CU; the chain: loss seeds + every layer in one launch)",
    "sample: k_sample_persist reverse steps without kernel boundaries (all layers + reverse update per step, row tiles synchronised through one XCD's L2)"};

namespace {

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _st = (call);                                                               \
    if (_st != hipSuccess) {                                                               \
      (e)->err = std::string(#call) + ": " + hipGetErrorString(_st);                       \
      return SDRM_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

// every kernel launch goes through here: the handle counts them (sdrm_launch_count: launches per step for bench.py)
#define SDRM_LAUNCH(eng, ...)                      \
  do {                                             \
    if (eng) ++(eng)->n_launches;                  \
    hipLaunchKernelGGL(__VA_ARGS__);               \
  } while (0)

int fail(sdrm_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

constexpr size_t SLACK = 4096;  // elements of zeroed tail on every buffer: unguarded tile loads may run past a matrix

// Dynamic LDS above 48 KB needs the kernel's limit raised.  Raised ONCE per kernel and process, to the whole 160 KB of a CU: a
// per-launch call is host time on every step of the narrow nets, and a per-engine value would let a second engine with a smaller
// image lower the limit under a first one with a larger image.
constexpr int LDS_MAX_BYTES = 160 * 1024;
hipError_t allow_full_lds(const void* kernel) {
  static std::mutex mu;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(kernel)) return hipSuccess;
  const hipError_t st = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES);
  if (st == hipSuccess) done.insert(kernel);
  return st;
}

template <typename Tp>
hipError_t dalloc(Tp** p, size_t n) {
  hipError_t st = hipMalloc((void**)p, (n + SLACK) * sizeof(Tp));
  if (st != hipSuccess) return st;
  return hipMemset(*p, 0, (n + SLACK) * sizeof(Tp));
}

float* pre_buf(sdrm_engine* e, int k) { return e->pre + (size_t)k * e->MPmax * e->WP; }
float* smp_buf(sdrm_engine* e, int k) { return e->smp_pre + (size_t)k * e->MPmax * e->WP; }   // the sampler's layer buffers
float* dpre_buf(sdrm_engine* e, int k) { return e->dA + (size_t)k * e->MPmax * e->WP; }   // d loss / d pre-activation k
const float* slope_ptr(sdrm_engine* e, int layer) { return e->p + (layer == 0 ? e->off_a0 : e->off_ah); }

// ---- GEMM launch helpers ----------------------------------------------------------------------
struct Prof { sdrm_engine* e; int cls; double flops; };

#ifdef SDRM_STAMPS
unsigned long long* g_wgrad_stamps = nullptr;   // diagnostic build: 8 stamp slots per work-group of the batched wgrad launch
int g_wgrad_stamps_cap = 0, g_wgrad_stamps_n = 0;
int g_stamp_class = -1;                         // -1: the batched wgrad launch; else the NT launches of this profile class
#endif


// Tile shapes (tools/gemm_tune.py, profiles/r01_gemm_tile_sweep_c.txt).  K is only ~350 deep, so a work-group is ~22
// K-steps long and what hides its prologue, epilogue and per-K-step barrier is many co-resident work-groups, not a
// big tile: 64x64x16 (20 KB of LDS, <= 80 VGPRs -> six per CU) for the large launches, 32x32x32 on the 16-wide MFMA
// for NT launches of up to 4096 rows (choose_cfg), the others are tuning alternates (SDRM_TILE).
typedef TileCfg<64, 64, 2, 2, 4, 16> Cfg0;     //  64x 64x16  (default)
typedef TileCfg<64, 64, 2, 2, 4, 32, 32, 1> Cfg1;  //  64x 64x32, one prefetch set
typedef TileCfg<64, 128, 2, 2, 4, 16> Cfg2;    //  64x128x16
typedef TileCfg<128, 128, 2, 2, 4, 16> Cfg3;   // 128x128x16
typedef TileCfg<32, 32, 2, 2, 4, 32, 16> Cfg4; //  32x 32x32 on v_mfma_f32_16x16x4_f32 (NT launches of <= 4096 rows)
constexpr int N_TILE_CFGS = 5;   // (a 192x64x16 tile - exactly one round of three fat work-groups per CU at 24576 x 352 - measured 3 % slower:
                                 //  profiles/r02_gemm_stagger_and_waveflip.txt)
const int kCfgBM[N_TILE_CFGS] = {64, 64, 64, 128, 32};
const int kCfgBN[N_TILE_CFGS] = {64, 64, 128, 128, 32};
constexpr int FUSE_REV_MAX_ROWS = 4096;   // = the launches that run on the 32x32 tile (4 accumulator rows per lane: two Philox calls);
                                          // measured (tools/shard_probe.py): 679 / 1358 / 2715 rows 21.1 / 24.9 / 36.6 -> 18.8 / 22.6 / 33.7 us per step;
                                          // on the 64x64 tile (16 rows per lane) 5429 rows 54.3 -> 56.9

// Tile of a split-K (weight-gradient) launch: the forced one, else the default.
int pick_cfg(const Tuning& t) { return (t.force_cfg >= 0 && t.force_cfg < N_TILE_CFGS) ? t.force_cfg : 0; }

// Tile for an unsplit (NT) launch of M rows; `nt32_rows` is the caller's threshold (Tuning::nt32_max_rows for sampling
// and plain forwards, ::nt32_max_rows_train for the train step).  With the k-minor LDS image and ds_read_b128
// fragments the 32x32x32 tile on the 16-wide MFMA has no ragged tile (352 = 11 x 32), four times the work-groups and a
// quarter of the dependent MFMA chain per wave.  Stand-alone (tools/gemm_tune.py, operands hot in L2) it matches or
// beats 64x64x16 at every size; inside the step, where every operand was just written by the previous launch, it wins
// up to a few thousand rows and loses above (tools/shard_probe.py, sample step at 679 / 1358 / 2715 / 5429 rows:
// 19.3 / 25.6 / 36.9 / 60.1 us against 23.5 / 31.6 / 49.9 / 54.2; train step at 3072 / 6144 / 12288 / 24576 stacked
// rows: 156 / 235 / 375 / 683 against 170 / 235 / 363 / 629) - the 64x64 tile moves half the operand bytes per flop.
// (64x32 and 32x64 tiles on the 16-wide MFMA were tried for the large launches: 644 / 661 us per train step against 618; round 5 tried
// 64x32x32, 32x64x32 and 64x64x32 on the 16-wide MFMA again where the 32x32 tile runs today - ML-100k's NT launches 350.5 / 345.6 / 378.5 us
// per train step against 348.0, the sampler's two chains at 5429 rows 46.3 / 45.7 / 58.9 us per step against 44.0, one chain at 679
// rows 22.1 / 21.0 / 32.7 against 16.7: profiles/r05_nt_tiles.txt.  The operand bytes per flop are not what bounds these launches.)
int choose_cfg(const Tuning& t, int M, int nt32_rows) {
  if (t.force_cfg >= 0 && t.force_cfg < N_TILE_CFGS) return t.force_cfg;
  return M <= nt32_rows ? 4 : 0;
}

int max_gemm_blocks(int M, int N) {
  int best = 0;
  for (int c = 0; c < N_TILE_CFGS; ++c) {
    const int b = ((M + kCfgBM[c] - 1) / kCfgBM[c]) * ((N + kCfgBN[c] - 1) / kCfgBN[c]);
    if (b > best) best = b;
  }
  return best;
}

template <class Cfg, int LA, int LB, int XA, int XB, int EPI>
hipError_t launch_gemm_cfg(GemmArgs& a, int M, int N, int splits, hipStream_t st, Prof pr) {
  const int tiles_m = (M + Cfg::BM - 1) / Cfg::BM, tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
  if (!gemm_set_grid(a, tiles_m, tiles_n, splits)) return hipErrorInvalidValue;   // beyond the exact range of the magic divisions
  // the tile loads address one tile x one K chunk with 32-bit offsets behind a 64-bit base (gemm.h: tile_resource)
  {
    const uint64_t kc = (uint64_t)std::max(a.kchunk, 1) + 64;
    const uint64_t spanA = LA == LD_KCONTIG ? ((uint64_t)Cfg::BM * a.lda + kc) : kc * a.lda;
    const uint64_t spanB = LB == LD_KCONTIG ? ((uint64_t)Cfg::BN * a.ldb + kc) : kc * a.ldb;
    if (spanA * 4 >= (1ull << 32) || spanB * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  }
  dim3 grid(EPI == EPI_SLAB ? (unsigned)(a.nblocks * splits) : (unsigned)a.nblocks, 1, 1);
  sdrm_engine* e = pr.e;
  const bool rec = e && e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == pr.cls);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(pr.cls);
    e->prof_flops.push_back(pr.flops);
    hipError_t st0 = hipEventRecord(e->prof_ev[2 * slot], st);
    if (st0 != hipSuccess) return st0;
  }
#ifdef SDRM_STAMPS
  if (g_wgrad_stamps && g_stamp_class >= 0 && pr.e && pr.cls == g_stamp_class && (int)grid.x <= g_wgrad_stamps_cap) {
    a.stamps = g_wgrad_stamps;
    g_wgrad_stamps_n = (int)grid.x;
  }
#endif
  SDRM_LAUNCH(pr.e, (gemm_kernel<Cfg, LA, LB, XA, XB, EPI>), grid, dim3(NTHREADS), 0, st, a);
  hipError_t rc = hipGetLastError();
  if (rec && rc == hipSuccess) rc = hipEventRecord(e->prof_ev[2 * slot + 1], st);
  return rc;
}

template <int LA, int LB, int XA, int XB, int EPI>
hipError_t launch_gemm(GemmArgs& a, int M, int N, int splits, hipStream_t st, Prof pr, int cfg) {
  switch (cfg) {
    case 1: return launch_gemm_cfg<Cfg1, LA, LB, XA, XB, EPI>(a, M, N, splits, st, pr);
    case 2: return launch_gemm_cfg<Cfg2, LA, LB, XA, XB, EPI>(a, M, N, splits, st, pr);
    case 3: return launch_gemm_cfg<Cfg3, LA, LB, XA, XB, EPI>(a, M, N, splits, st, pr);
    case 4: return launch_gemm_cfg<Cfg4, LA, LB, XA, XB, EPI>(a, M, N, splits, st, pr);
    default: return launch_gemm_cfg<Cfg0, LA, LB, XA, XB, EPI>(a, M, N, splits, st, pr);
  }
}

// forward Linear: C[M,N] = act(xf(A)[M,K] * Wc[N,K]^T + bias)
template <int XA, int EPI>
hipError_t gemm_forward(GemmArgs a, const float* A, int lda, const float* Wc, int ldw, int M, int N, int K,
                        hipStream_t st, Prof pr, int cfg) {
  a.A = A; a.lda = lda; a.limA = M;
  a.B = Wc; a.ldb = ldw; a.limB = N;
  a.K = K; a.kchunk = K;
  return launch_gemm<LD_KCONTIG, LD_KCONTIG, XA, XF_NONE, EPI>(a, M, N, 1, st, pr, cfg);
}

// dgrad: C[M,Kin] = (dC[M,Nout] * W[Nout,Kin]) * prelu'(aux), against the transposed copy WT[Kin][Nout] (NT form)
hipError_t gemm_dgrad(sdrm_engine* e, const float* dC, int lddc, const float* WT, int ldwt, int M, int Nout, int Kin,
                      float* out, const float* aux, const float* slopeE, float* partial, hipStream_t st, double flops,
                      int cfg) {
  GemmArgs a{};
  a.A = dC; a.lda = lddc; a.limA = M;
  a.B = WT; a.ldb = ldwt; a.limB = Kin;
  a.C = out; a.ldc = e->WP;
  a.K = Nout; a.kchunk = Nout;
  a.aux = aux; a.ldaux = e->WP; a.slopeE = slopeE; a.slope_partial = partial;
  return launch_gemm<LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_DPRELU>(a, M, Kin, 1, st, Prof{e, PC_DGRAD, flops}, cfg);
}

// wgrad: slab[s][Nout,Kin] = dC[rows s][.,Nout]^T * xf(Act)[rows s][., Kin] ; dbias[s][Nout] = column sums of dC
template <int XB>
hipError_t gemm_wgrad(const float* dC, int lddc, int Nout, const float* Act, int ldact, int Kin, const float* slopeB,
                      int Mrows, int S, int kchunk, float* slab, float* dbias, hipStream_t st, Prof pr, int cfg) {
  GemmArgs a{};
  a.A = dC; a.lda = lddc; a.limA = Nout;
  a.B = Act; a.ldb = ldact; a.limB = Kin;
  a.C = slab; a.ldc = Kin;
  a.K = Mrows; a.kchunk = kchunk;
  a.slopeB = slopeB;
  a.slab_stride = (size_t)Nout * Kin;
  a.dbias = dbias; a.dbias_stride = Nout;
  return launch_gemm<LD_MCONTIG, LD_MCONTIG, XF_NONE, XB, EPI_SLAB>(a, Nout, Kin, S, st, pr, cfg);
}

// One weight gradient of a batched launch (gemm_batch_kernel): the arguments gemm_wgrad would pass, with the grid
// bookkeeping launch_gemm_cfg does, for the default tile.  slopeB must be non-null: layer 0, whose operand is not a
// stored pre-activation, passes a slope of exactly 1 (v > 0 ? v : 1*v is v bit for bit).
struct WgradSpec {
  const float* dC; int lddc, Nout; const float* Act; int ldact, Kin; const float* slopeB; int S, kchunk; float* slab; float* dbias;
};

hipError_t launch_wgrad_batch(sdrm_engine* e, const WgradSpec* w, int n, int Mrows, hipStream_t st, Prof pr, bool plain_b = false) {
  GemmBatch b{};
  b.n = n;
  int grid = 0;
  for (int k = 0; k < n; ++k) {
    GemmArgs& a = b.p[k];
    a.A = w[k].dC; a.lda = w[k].lddc; a.limA = w[k].Nout;
    a.B = w[k].Act; a.ldb = w[k].ldact; a.limB = w[k].Kin;
    a.C = w[k].slab; a.ldc = w[k].Kin;
    a.K = Mrows; a.kchunk = w[k].kchunk;
    a.slopeB = w[k].slopeB;
    a.slab_stride = (size_t)w[k].Nout * w[k].Kin;
    a.dbias = w[k].dbias; a.dbias_stride = w[k].Nout;
    const int tiles_m = (w[k].Nout + Cfg0::BM - 1) / Cfg0::BM, tiles_n = (w[k].Kin + Cfg0::BN - 1) / Cfg0::BN;
    if (!gemm_set_grid(a, tiles_m, tiles_n, w[k].S)) return hipErrorInvalidValue;
    // 32-bit offsets inside one K chunk (gemm.h: tile_resource): both operands are M-contiguous here
    if (((uint64_t)std::max(a.kchunk, 1) + 64) * (uint64_t)std::max(a.lda, a.ldb) * 4 >= (1ull << 32)) return hipErrorInvalidValue;
    b.start[k] = grid;
#ifdef SDRM_STAMPS
    a.stamps = (g_wgrad_stamps && g_stamp_class < 0) ? g_wgrad_stamps + 8 * (size_t)grid : nullptr;
#endif
    grid += round_up(a.nblocks * w[k].S, 8);   // a multiple of 8: the XCD of a work-group is the same inside its problem
  }
  b.start[n] = grid;
#ifdef SDRM_STAMPS
  if (grid > g_wgrad_stamps_cap) for (int k = 0; k < n; ++k) b.p[k].stamps = nullptr;
  else if (g_stamp_class < 0) g_wgrad_stamps_n = grid;
#endif
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == pr.cls);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(pr.cls);
    e->prof_flops.push_back(pr.flops);
    hipError_t st0 = hipEventRecord(e->prof_ev[2 * slot], st);
    if (st0 != hipSuccess) return st0;
  }
  // plain_b: every B operand is stored as the kernel needs it (activations written by the row-owned forward): no PReLU on load
  if (plain_b)
    SDRM_LAUNCH(e, (gemm_batch_kernel<Cfg0, LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_SLAB>), dim3((unsigned)grid),
                       dim3(NTHREADS), 0, st, b);
  else
    SDRM_LAUNCH(e, (gemm_batch_kernel<Cfg0, LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_PRELU, EPI_SLAB>), dim3((unsigned)grid),
                       dim3(NTHREADS), 0, st, b);
  hipError_t rc = hipGetLastError();
  if (rec && rc == hipSuccess) rc = hipEventRecord(e->prof_ev[2 * slot + 1], st);
  return rc;
}

// Split-K plan of a backward's weight gradients: ONE slice count for all of them (so the one-call and the two-call backward
// add the same partial sums in the same order), chosen so that the batched launch is ONE round of what the device holds at
// once (5 work-groups per CU, 1280 on MI355X) when that cuts the reduction eight ways or more with slices of at most 1024
// rows, and 2.2 rounds (2816 work-groups) otherwise.  Measured on one box, train step in us
// (profiles/r02_wgrad_slices_any_count.txt): ML-1M, 114 tiles, 24576 rows: 11 slices (one round) 572, 22 / 24 slices 573 /
// 573, 10 / 12 / 13 slices (just under / over a round) 586-588 - equal times, but the 2240-row slices of the one-round
// plan do not fit an XCD's L2 and double the launch's HBM traffic, so this size takes 24; the same net at the row counts
// of a 4- / 8-GPU shard: 11 slices 204.0 / 139.7, 24 slices 207.8 / 144.3 (a slice of 128 rows is mostly prologue and
// epilogue); ML-100k, 702 tiles: 4 slices 356, 3: 359, 1-2: 361, 8 (round 1's multiple of 8): 369.
// The (slice, tile) units are dealt to the XCDs in contiguous runs (gemm.h), so any slice count fills the chip evenly.
void pick_splits(const Tuning& tn, int Mrows, int tiles_total, int& S, int& kchunk) {
  const int BK = 32;
  int max_by_rows = Mrows / (4 * BK);  // at least 4 K-steps of 32 rows per work-group
  if (max_by_rows < 1) max_by_rows = 1;
  const int tiles = tiles_total < 1 ? 1 : tiles_total;
  // one round when that cuts the reduction eight ways or more AND a slice stays within 1024 rows - the tiles of a slice share
  // its operand strips through one XCD's 4 MB L2: at 24576 rows 11 slices of 2240 rows take the same 204 us as 24 slices of
  // 1024 but move 653 MB instead of 334 MB (PMC) - else 2.2 rounds (floor: one block more is a round more)
  S = tn.wgrad_round / tiles;
  if (S < 8 || round_up((Mrows + S - 1) / S, BK) > 1024) S = tn.wgrad_blocks / tiles;
  if (S < 1) S = 1;
  if (tn.wgrad_slices > 0) S = tn.wgrad_slices;
  if (S > S_MAX) S = S_MAX;
  if (S > max_by_rows) S = max_by_rows;
  kchunk = round_up((Mrows + S - 1) / S, BK);
  S = (Mrows + kchunk - 1) / kchunk;
}

int wgrad_tiles(const Tuning& tn, int Nout, int Kin) {
  const int c = pick_cfg(tn);
  return ((Nout + kCfgBM[c] - 1) / kCfgBM[c]) * ((Kin + kCfgBN[c] - 1) / kCfgBN[c]);
}

// the parameter tensors and their compute copies, as k_adam (sdrm_adam_step, sdrm_set_params) walks them
void build_jobs(sdrm_engine* e, JobTable& tab) {
  const int L = e->L, W = e->W, T = e->T, H = e->H;
  int n = 0;
  auto add = [&](int64_t off, int rows, int cols, int flat_ld, int ncols, float* dst, int dst_ld, float* dstT = nullptr, int dstT_ld = 0) -> Job& {
    Job& j = tab.j[n++];
    j.flat_off = off; j.rows = rows; j.cols = cols; j.flat_ld = flat_ld; j.ncols = ncols;
    j.dst = dst; j.dst_ld = dst_ld; j.dstT = dstT; j.dstT_ld = dstT_ld; j.dst2 = nullptr; j.dst2_ld = 0;
    j.dstF = nullptr; j.dstFT = nullptr; j.fnct = e->WP / 16; j.fklast = -1; j.fklastT = -1;
    return j;
  };
  add(e->off_we, T, T, T, e->WeP ? T : 0, e->WeP, e->TPe);   // emb_layer.weight (a padded copy for the narrow nets' forward only)
  add(e->off_be, 1, T, T, 0, nullptr, 0);                    // emb_layer.bias
  {
    Job& j = add(e->off_w0, W, L + T, L + T, L, e->W0c, e->K0);
    j.dstF = e->W0f; j.fklast = rc_light_klast(L, e->LP);
    j.dst2 = e->W0eP; j.dst2_ld = e->TPe;
  }
  add(e->off_b0, 1, W, W, W, e->b0c, 0);
  add(e->off_a0, 1, 1, 1, 1, nullptr, 0);
  if (H >= 1) {
    Job& j = add(e->off_wh, W, W, W, W, e->Whc, e->WP, e->WhcT, e->WP);
    j.dstF = e->Whf; j.dstFT = e->WhfT; j.fklast = j.fklastT = rc_light_klast(W, e->WP);
    add(e->off_bh, 1, W, W, W, e->bhc, 0);
    add(e->off_ah, 1, 1, 1, 1, nullptr, 0);
  }
  {
    Job& j = add(e->off_wo, L, W, W, W, e->Woc, e->WP, e->WocT, e->LP);
    j.dstF = e->Wof; j.dstFT = e->WofT; j.fklast = rc_light_klast(W, e->WP); j.fklastT = rc_light_klast(L, e->LP);
  }
  add(e->off_bo, 1, L, L, L, e->boc, 0);
  tab.n = n;
}

// Adam's bias corrections of step e->adam_t (train_SDRM.py:337; torch.optim.Adam: step_size = lr / (1 - b1^k), denominators
// scaled by sqrt(1 - b2^k))
void adam_scalars(const sdrm_engine* e, float lr, float& step_size, float& bc2_sqrt) {
  const double k = (double)e->adam_t;
  const double bc1 = 1.0 - std::pow(0.9, k), bc2 = 1.0 - std::pow(0.999, k);
  step_size = (float)((double)lr / bc1);
  bc2_sqrt = (float)std::sqrt(bc2);
}

int launch_adam(sdrm_engine* e, const float* grad, float lr, int update, hipStream_t st) {
  JobTable tab;
  build_jobs(e, tab);
  AdamArgs a{};
  a.p = e->p; a.m = e->m; a.v = e->v; a.g = grad ? grad : e->g;
  a.b1 = 0.9f; a.b2 = 0.999f; a.eps = 1e-8f; a.wd = 1e-4f; a.update = update;
  if (update) adam_scalars(e, lr, a.step_size, a.bc2_sqrt);
  // enough work-groups that the largest tensor is one or two passes per thread (a pass is a chain of dependent loads)
  int64_t biggest = 0;
  for (int k = 0; k < tab.n; ++k) biggest = std::max<int64_t>(biggest, (int64_t)tab.j[k].rows * tab.j[k].cols);
  const int gx = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (biggest + 511) / 512));
  dim3 grid(gx, tab.n);
  SDRM_LAUNCH(e, k_adam, grid, dim3(256), 0, st, tab, a);
  HIP_TRY(e, hipGetLastError());
  e->tables_fresh = false;
  return SDRM_OK;
}

// Strip-owned weight gradients (csrc/wgrad2.h): every weight gradient of the step in one balanced round of one work-group per CU.
// Taken by the one-call backward when the forward was the row-owned one (the operands are stored as the kernel reads them:
// activations, a ones column for the bias gradients) and the slices stay within the slab count.
bool use_strips(const sdrm_engine* e) {
  return (e->cur_grouped || e->cur_g16) && e->cur_act && e->ones_col >= 0 && e->tune.strips > 0 && e->H + 2 <= WG2_MAX_PROBLEMS && e->WP >= 128 && e->WP <= 352;
}

template <int NT>
int launch_wgrad_strips_nt(sdrm_engine* e, const Wg2Args& a, double flops, hipStream_t st) {
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_WGRAD_STRIPS);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_WGRAD_STRIPS);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  SDRM_LAUNCH(e, (k_wgrad_strips<NT>), dim3((unsigned)(a.units * a.slices)), dim3(NTHREADS), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

int launch_wgrad_strips(sdrm_engine* e, int MP, double flops, hipStream_t st) {
  const int H = e->H;
  Wg2Args a{};
  int ktiles[WG2_MAX_PROBLEMS], n = 0;
  auto problem = [&](const float* A, int lda, const float* B, int ldb, float* slab, int kin, size_t slab_rows) {
    a.p[n].A = A; a.p[n].lda = lda; a.p[n].B = B; a.p[n].ldb = ldb; a.p[n].slab = slab; a.p[n].ldc = kin;
    a.p[n].slab_stride = slab_rows * (size_t)kin;
    ktiles[n++] = kin / 32;
  };
  auto actk = [&](int k) { return e->act + (size_t)k * e->MPmax * e->WP; };
  problem(dpre_buf(e, 0), e->WP, e->U, e->K0, e->slab0, e->K0, (size_t)e->WP);
  problem(e->dY, e->LP, actk(H), e->WP, e->slabO, e->WP, (size_t)e->LP);
  const int units = [&] { int t = e->K0 / 32 + (H + 1) * (e->WP / 32); return (t + 3) / 4; }();
  int S = 256 / units;
  if (S < 1) S = 1;
  if (S > S_MAX) S = S_MAX;
  MP = e->cur_rows;   // the rows the forward wrote (whole groups, a multiple of 16): the padding behind them is not summed over
  const int kchunk = round_up((MP + S - 1) / S, WG2_BK);
  const int slices = (MP + kchunk - 1) / kchunk;
  for (int k = H; k >= 1; --k)   // the shared hidden layer: one problem per application, slabs [application][slice]
    problem(dpre_buf(e, k), e->WP, actk(k - 1), e->WP, e->slabH + (size_t)(k - 1) * slices * e->WP * e->WP, e->WP, (size_t)e->WP);
  if (wg2_plan(ktiles, n, a) != units) return fail(e, SDRM_ERR_SHAPE, "strip-owned weight gradients: plan does not fit");
  a.slices = slices; a.rows = MP; a.kchunk = kchunk;
  e->bwd_S0 = e->bwd_SH = e->bwd_SO = slices;
  e->bwd_strips = true;
  switch (e->WP / 32) {
    case 4: return launch_wgrad_strips_nt<4>(e, a, flops, st);
    case 5: return launch_wgrad_strips_nt<5>(e, a, flops, st);
    case 6: return launch_wgrad_strips_nt<6>(e, a, flops, st);
    case 7: return launch_wgrad_strips_nt<7>(e, a, flops, st);
    case 8: return launch_wgrad_strips_nt<8>(e, a, flops, st);
    case 9: return launch_wgrad_strips_nt<9>(e, a, flops, st);
    case 10: return launch_wgrad_strips_nt<10>(e, a, flops, st);
    default: return launch_wgrad_strips_nt<11>(e, a, flops, st);
  }
}

EmbTabArgs emb_args(sdrm_engine* e) {
  EmbTabArgs a{};
  a.temb = e->temb; a.We = e->p + e->off_we; a.be = e->p + e->off_be; a.W0 = e->p + e->off_w0; a.b0 = e->p + e->off_b0;
  a.W0c = e->W0c; a.B0tab = e->B0tab;
  a.L = e->L; a.W = e->W; a.T = e->T; a.LP = e->LP; a.WP = e->WP; a.K0 = e->K0;
  a.ones_col = e->ones_col;
  return a;
}

// The time-embedding tables of the current parameters: B0tab[t] = b0 + C0[t] (what layer 0 of a train step or of a sampling step
// adds per row) and C0^T in the trailing columns of W0c (what the plain forward multiplies its one-hot(t) columns with).
int emb_tables(sdrm_engine* e, hipStream_t st, const float* warm = nullptr, size_t warm_floats = 0) {
  EmbTabArgs a = emb_args(e);
  a.warm = warm; a.warm_lines = warm ? (unsigned)std::min<size_t>(warm_floats / 32, 1u << 24) : 0u;
  SDRM_LAUNCH(e, k_emb_tables, dim3(e->T + 1), dim3(1024), 2 * e->T * sizeof(float), st, a);
  HIP_TRY(e, hipGetLastError());
  e->tables_fresh = true;
  return SDRM_OK;
}
// ... made only when the parameters changed since they were last made (sdrm_set_params, sdrm_adam_step, every train step: the tail
// updates the parameters and clears `tables_fresh`; the tables are then made by the next consumer - k_emb_tables in front of the
// row-owned forward, the leading blocks of k_prep_train, the narrow nets' forward itself when T <= 128, or this launch: narrow
// nets with T > 128 and the sampler pay it as a launch of its own)
// (Round 5 also made the next row-owned step's tables AHEAD - on aux[2], behind the tail of the last train step, beside the first sampling
// step that follows it in bench.py's walk: 8967 -> 8907 steps/s in the driver's form, 9069 -> 8994 with the default windows; the fork, the
// join and a forward that finds its batch cold cost more than the 10 us launch saved.  profiles/r05_walk_transitions.txt)
int ensure_tables(sdrm_engine* e, hipStream_t st) { return e->tables_fresh ? SDRM_OK : emb_tables(e, st); }

bool skinny_net(const sdrm_engine* e) { return e->tune.skinny && e->LP <= 64 && e->WP <= 64; }

// Row-owned train forward (rowchain.h): one 96-row work-group per CU.  It replaces staging + H + 2 GEMM launches + the loss
// partial sums when the batch fills whole rounds of the chip's 256 CUs (measured at B = 8192, L = 340: 169 us against
// 182 + 18.7 + 11.7 us); a last round that leaves more than a sixth of the CUs idle loses to the per-layer path.
// row-owned dgrads (dgrad_rows.h): the stacked rows are whole 96-row work-groups (the grouped order of the row-owned forward),
// reduction axis == output axis == the padded width (LP == WP: L == W)
// will the backward of a row-owned forward of `rows` stacked rows (padded to MP) be the row-owned dgrads (dr_layer: chain or one
// launch per layer)?  Asked by the forward too: they read activations, so it need not store pre-activations (rowchain.h: skip_pre)
bool row_dgrads_follow(const sdrm_engine* e, bool g16, int MP) {
  if (!e->WhfT || e->tune.dgrad_rows <= 0 || e->LP != e->WP || e->WP < 128 || e->WP > 352) return false;
  return g16 ? e->H + 1 <= DR_MAX_LAYERS : MP % RC_ROWS == 0;
}
bool use_dgrad_rows(const sdrm_engine* e, int MP) {
  return e->cur_grouped && e->WhfT && e->tune.dgrad_rows > 0 && e->LP == e->WP && MP % RC_ROWS == 0 && e->WP >= 128 && e->WP <= 352;
}

template <int CT>
int launch_dgrad_rows_ct(sdrm_engine* e, const DgradRowsArgs& a, int G, double flops, hipStream_t st) {
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_DGRAD_ROWS);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_DGRAD_ROWS);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  // (the weight copies' compact last K-step, elementwise.h: L == W on this path)
  if (rc_light_klast(e->W, e->WP) >= 0) SDRM_LAUNCH(e, (k_dgrad_rows<CT, true>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  else SDRM_LAUNCH(e, (k_dgrad_rows<CT, false>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

// out[MP][WP] = (G[MP][WP] * W) * prelu'(pre), W given as the fragment-packed [k = out][n = in] copy
DgradRowsArgs dgrad_rows_args(const sdrm_engine* e, const float* G, const float* WfT, const float* pre, const float* slope, float* out,
                              float* partial) {
  DgradRowsArgs a{};
  a.G = G; a.ldg = e->WP; a.WfT = WfT; a.pre = pre; a.ldp = e->WP; a.slope = slope; a.out = out; a.ldo = e->WP; a.slope_part = partial;
  // the layer's activations sit at the same place of the `act` buffer as its pre-activations in `pre`
  a.act = e->act ? e->act + (pre - e->pre) : nullptr; a.from_act = (e->cur_act && e->cur_skip_pre && a.act) ? 1 : 0;
  return a;
}

int launch_dgrad_rows(sdrm_engine* e, const float* G, const float* WfT, const float* pre, const float* slope, float* out, float* partial,
                      int MP, double flops, hipStream_t st) {
  const DgradRowsArgs a = dgrad_rows_args(e, G, WfT, pre, slope, out, partial);
  const int Gn = MP / RC_ROWS;
  switch (e->WP / 32) {
    case 4: return launch_dgrad_rows_ct<4>(e, a, Gn, flops, st);
    case 5: return launch_dgrad_rows_ct<5>(e, a, Gn, flops, st);
    case 6: return launch_dgrad_rows_ct<6>(e, a, Gn, flops, st);
    case 7: return launch_dgrad_rows_ct<7>(e, a, Gn, flops, st);
    case 8: return launch_dgrad_rows_ct<8>(e, a, Gn, flops, st);
    case 9: return launch_dgrad_rows_ct<9>(e, a, Gn, flops, st);
    case 10: return launch_dgrad_rows_ct<10>(e, a, Gn, flops, st);
    default: return launch_dgrad_rows_ct<11>(e, a, Gn, flops, st);
  }
}

// loss value + gradient seeds + every layer's dgrad in one launch (k_dgrad_chain)
template <int CT>
int launch_dgrad_chain_ct(sdrm_engine* e, const DgradChainArgs& a, int G, double flops, hipStream_t st) {
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_DGRAD_ROWS);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_DGRAD_ROWS);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  if (rc_light_klast(e->W, e->WP) >= 0) SDRM_LAUNCH(e, (k_dgrad_chain<CT, true>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  else SDRM_LAUNCH(e, (k_dgrad_chain<CT, false>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

int launch_dgrad_chain(sdrm_engine* e, const DgradChainArgs& a, int MP, double flops, hipStream_t st) {
  const int Gn = MP / RC_ROWS;
  switch (e->WP / 32) {
    case 4: return launch_dgrad_chain_ct<4>(e, a, Gn, flops, st);
    case 5: return launch_dgrad_chain_ct<5>(e, a, Gn, flops, st);
    case 6: return launch_dgrad_chain_ct<6>(e, a, Gn, flops, st);
    case 7: return launch_dgrad_chain_ct<7>(e, a, Gn, flops, st);
    case 8: return launch_dgrad_chain_ct<8>(e, a, Gn, flops, st);
    case 9: return launch_dgrad_chain_ct<9>(e, a, Gn, flops, st);
    case 10: return launch_dgrad_chain_ct<10>(e, a, Gn, flops, st);
    default: return launch_dgrad_chain_ct<11>(e, a, Gn, flops, st);
  }
}

bool use_rowchain(const sdrm_engine* e, int B) {
  if (!e->W0f || e->tune.rowchain <= 0 || e->tune.force_cfg >= 0) return false;   // a forced tile means: the per-layer kernels
  if (e->tune.rowchain >= 2) return true;
  const int G = (B + RC_USERS - 1) / RC_USERS;
  const int rounds = (G + 255) / 256;
  // one round: from 154 groups (B = 4897) on the three row-owned launches beat the per-layer path's eleven although two fifths of
  // the CUs idle (round 5, tools/rows48_probe.py: B = 5120 383 against 397 us, B = 6144 406 against 462); several rounds: the
  // last one at least five sixths full
  return rounds == 1 ? G >= 154 : G * 6 >= rounds * 256 * 5;
}

// The same step on 48-row work-groups (csrc/rows48.h), for batches the 96-row kernels would leave CUs idle with.  Returns the
// work-groups per 16-user row group, 0: not this path.
//   1: the groups of the batch fill most of ONE round of the chip (160..256 groups: 2545..4096 users; two work-groups fit a CU, but
//      a second round's worth then shares the matrix pipes: no faster than the per-layer path);
//   2 / 4 (column-split groups, exchanged through one XCD's L2): batches of at most 2048 / 1024 users, whose groups x parts fit the
//      chip's 256 CUs - every work-group of such a launch must be resident at once; by size only 2, for 1281 .. 2048 users.
// the abort word of the XCD-local launches lives in host-mapped memory the GPU writes: read it as volatile
inline unsigned xabort_read(const sdrm_engine* e) { return e->xabort_host ? *(volatile const unsigned*)e->xabort_host : 0u; }

int rows48_grid(int groups, int parts) { return parts == 1 ? groups : 8 * parts * ((groups + 7) / 8); }
int rows48_parts(const sdrm_engine* e, int B) {
  if (!e->W0f || e->tune.rows48 <= 0 || e->tune.force_cfg >= 0 || e->tune.rowchain >= 2) return 0;
  const int G = (B + R48_USERS - 1) / R48_USERS;
  const bool can_split = e->xcd_ok && e->tune.split > 0 && e->xabort_host && xabort_read(e) == 0u;
  if (can_split && e->tune.split >= 2) {
    const int parts = e->tune.split >= 4 ? 4 : 2;
    if (rows48_grid(G, parts) <= 256) return parts;
    if (parts == 4 && rows48_grid(G, 2) <= 256) return 2;
  }
  if (e->tune.rows48 >= 2) return 1;
  // (measured, tools/rows48_probe.py, profiles/r05_rows48_probe.txt: two work-groups per group beat the per-layer path from ~1300
  // users on - B = 2048: 174 against 201 us, B = 1536: 164 against 170 - four per group do not: B = 1024: 140 against 131, the
  // redundant staging and the two hand-shakes per kernel cost what the five launches saved)
  // ... unless a small sampling call is in progress on its detached chain (chains_for): beside it the eleven launches of the per-layer
  // path, which it may run along with, beat the column-split kernels, which hold it (tools/ab/walk_host.py, one rank of four - 2048 users,
  // 1358 sampled rows - in bench.py's walk: 25.2 k -> 27.1 k steps/s).  The two paths differ in the last bits of a step.
  const bool beside_sampler = e->smp.active && (e->detach_armed || (e->chains_pending && e->n_aux == e->n_chains));
  if (can_split && e->tune.split == 1 && G > 80 && rows48_grid(G, 2) <= 256) return beside_sampler ? 0 : 2;       // 1281 .. 2048 users
  return (G >= 160 && G <= 256) ? 1 : 0;   // (2545 .. 4096 users; measured: 2560 users 231 against 235 us, 2688 231 / 241, 2432 228 / 224)
}

// the hand-shake counters of a column-split launch: they count on from launch to launch while the geometry stays the same
int split_sync(sdrm_engine* e, bool chain, int groups, int parts, int phases, hipStream_t st, unsigned** cnt, unsigned* base) {
  unsigned*& c = chain ? e->xcntC : e->xcntF;
  uint32_t& epoch = chain ? e->xepochC : e->xepochF;
  int& geo = chain ? e->xgeoC : e->xgeoF;
  const int now = (parts << 16) | groups | (phases << 24);
  if (geo != now || epoch >= 0x7fffffffu / (uint32_t)(phases * parts + 1)) {
    HIP_TRY(e, hipMemsetAsync(c, 0, (size_t)256 * 32 * sizeof(unsigned), st));
    epoch = 0; geo = now;
  }
  *cnt = c;
  *base = epoch * (uint32_t)(phases * parts) + e->xskew;
  if (e->xskew) { e->xskew = 0; geo = 0; }   // (the counters of a skewed launch are worthless: start over with the next one)
  ++epoch;
  return SDRM_OK;
}

// a timed-out hand-shake of an earlier column-split launch (csrc/rows48.h): reported by the next call, the path switched off
int split_status(sdrm_engine* e) {
  if (xabort_read(e) == 0u) return SDRM_OK;
  *(volatile unsigned*)e->xabort_host = 0u;   // reported once; the counters start over should the path ever be switched on again
  e->xgeoF = e->xgeoC = 0;
  e->tune.split = 0;
  e->fwd_done = false;
  return fail(e, SDRM_ERR_HIP, "a column-split row-group launch (csrc/rows48.h) timed out waiting for a work-group of its group: the results of "
                               "that train step are invalid; the path is switched off for this handle");
}

template <int CT, int PARTS>
int launch_rows48_forward_ctp(sdrm_engine* e, const RowChainArgs& a, int grid, hipStream_t st) {
  if constexpr (PARTS == 1 && CT % 2 == 1) {   // 2 CT = 4 q + 2 column tiles: the shared-tile form
    if (e->tune.rows48_share & 1) {
      if (a.light) SDRM_LAUNCH(e, (k_rows48_fwd<CT, true, 1, true>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
      else SDRM_LAUNCH(e, (k_rows48_fwd<CT, false, 1, true>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
      return SDRM_OK;
    }
  }
  if (a.light) SDRM_LAUNCH(e, (k_rows48_fwd<CT, true, PARTS>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
  else SDRM_LAUNCH(e, (k_rows48_fwd<CT, false, PARTS>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
  return SDRM_OK;
}

template <int CT>
int launch_rows48_forward_ct(sdrm_engine* e, RowChainArgs& a, int G, int parts, hipStream_t st) {
  const double flops = 2.0 * 3 * a.B * ((double)e->W * e->L + (double)e->H * e->W * e->W + (double)e->L * e->W);
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_ROW_FWD);
  size_t slot = 0;
  if (parts > 1) {
    if (int rc = split_sync(e, false, G, parts, e->H + 1, st, &a.xcnt, &a.xbase)) return rc;
    a.xabort = e->xabort_dev; a.ngroups = G;
  }
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_ROW_FWD);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  const int grid = rows48_grid(G, parts);
  if (parts == 4) launch_rows48_forward_ctp<CT, 4>(e, a, grid, st);
  else if (parts == 2) launch_rows48_forward_ctp<CT, 2>(e, a, grid, st);
  else launch_rows48_forward_ctp<CT, 1>(e, a, grid, st);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

template <int CT, int PARTS>
int launch_rows48_chain_ctp(sdrm_engine* e, const DgradChain48Args& a, int grid, hipStream_t st) {
  if constexpr (PARTS == 1 && CT % 2 == 1) {   // 2 CT = 4 q + 2 column tiles: the shared-tile form
    if (e->tune.rows48_share & 2) {
      if (rc_light_klast(e->W, e->WP) >= 0) SDRM_LAUNCH(e, (k_rows48_dgrad_chain<CT, true, 1, true>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
      else SDRM_LAUNCH(e, (k_rows48_dgrad_chain<CT, false, 1, true>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
      return SDRM_OK;
    }
  }
  if (rc_light_klast(e->W, e->WP) >= 0) SDRM_LAUNCH(e, (k_rows48_dgrad_chain<CT, true, PARTS>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
  else SDRM_LAUNCH(e, (k_rows48_dgrad_chain<CT, false, PARTS>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, a);
  return SDRM_OK;
}

template <int CT>
int launch_rows48_chain_ct(sdrm_engine* e, DgradChain48Args& a, int G, int parts, double flops, hipStream_t st) {
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_DGRAD_ROWS);
  size_t slot = 0;
  if (parts > 1) {
    if (int rc = split_sync(e, true, G, parts, std::max(1, a.c.nlayers - 1), st, &a.xcnt, &a.xbase)) return rc;
    a.xabort = e->xabort_dev; a.ngroups = G;
  }
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_DGRAD_ROWS);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  const int grid = rows48_grid(G, parts);
  if (parts == 4) launch_rows48_chain_ctp<CT, 4>(e, a, grid, st);
  else if (parts == 2) launch_rows48_chain_ctp<CT, 2>(e, a, grid, st);
  else launch_rows48_chain_ctp<CT, 1>(e, a, grid, st);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

int launch_rows48_chain(sdrm_engine* e, DgradChain48Args& a, int G, int parts, double flops, hipStream_t st) {
  switch (e->WP / 32) {
    case 4: return launch_rows48_chain_ct<4>(e, a, G, parts, flops, st);
    case 5: return launch_rows48_chain_ct<5>(e, a, G, parts, flops, st);
    case 6: return launch_rows48_chain_ct<6>(e, a, G, parts, flops, st);
    case 7: return launch_rows48_chain_ct<7>(e, a, G, parts, flops, st);
    case 8: return launch_rows48_chain_ct<8>(e, a, G, parts, flops, st);
    case 9: return launch_rows48_chain_ct<9>(e, a, G, parts, flops, st);
    case 10: return launch_rows48_chain_ct<10>(e, a, G, parts, flops, st);
    default: return launch_rows48_chain_ct<11>(e, a, G, parts, flops, st);
  }
}

template <int CT>
int launch_row_forward_ct(sdrm_engine* e, const RowChainArgs& a, int G, hipStream_t st) {
  // profile class: the algorithmic flops of the H + 2 layers of the three passes (staging, loss sums and the streamed copies ride along)
  const double flops = 2.0 * 3 * a.B * ((double)e->W * e->L + (double)e->H * e->W * e->W + (double)e->L * e->W);
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_ROW_FWD);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_ROW_FWD);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  if (a.light) SDRM_LAUNCH(e, (k_row_fwd<CT, true>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  else SDRM_LAUNCH(e, (k_row_fwd<CT, false>), dim3((unsigned)G), dim3(NTHREADS), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  return SDRM_OK;
}

int launch_row_forward(sdrm_engine* e, const float* x0, int B, int64_t row0, int mode, const sdrm_train_randoms* rnd, uint64_t seed,
                       uint64_t step, float nd, int G, hipStream_t st, int rows48_parts_ = 0) {
  const int n = e->T + 1;
  RowChainArgs a{};
  a.x0 = x0;
  if (mode == SDRM_RNG_EXPLICIT) { a.noise = rnd->noise; a.t = rnd->t; a.keep = rnd->keep; }
  a.sqrt_ab = e->sched + 3 * n; a.one_minus_ab = e->sched + 4 * n; a.tembP = e->tembP;
  a.B = B; a.L = e->L; a.T = e->T; a.H = e->H;
  a.mode = mode; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.step = (uint32_t)step; a.row0 = row0; a.nd = nd;
  a.W0f = e->W0f; a.Whf = e->Whf; a.Wof = e->Wof; a.bh = e->bhc; a.bo = e->boc; a.B0tab = e->B0tab; a.ldtab = e->WP;
  a.slope0 = slope_ptr(e, 0); a.slopeh = e->H > 0 ? slope_ptr(e, 1) : slope_ptr(e, 0);
  a.U = e->U; a.K0 = e->K0; a.LPs = e->LP; a.tdev = e->tdev; a.ones_col = e->ones_col;
  a.light = rc_light_klast(e->W, e->WP) >= 0 ? 1 : 0;   // L == W on this path: one answer for every layer
  a.pre = e->pre; a.pre_stride = (size_t)e->MPmax * e->WP; a.ldp = e->WP; a.Y = e->Y; a.ldy = e->LP;
  a.act = e->act;
  a.loss_part = e->loss_part;
  a.sweep = (e->tune.rows48_share & 4) ? 1 : 0;
  {
    const int MPg = rows48_parts_ > 0 ? round_up(G * R48_ROWS, BM) : round_up(G * RC_ROWS, BM);
    a.skip_pre = row_dgrads_follow(e, rows48_parts_ > 0, MPg) ? 1 : 0;
    e->cur_skip_pre = a.skip_pre != 0;
  }
  if (rows48_parts_ > 0) {
    switch (e->WP / 32) {
      case 4: return launch_rows48_forward_ct<4>(e, a, G, rows48_parts_, st);
      case 5: return launch_rows48_forward_ct<5>(e, a, G, rows48_parts_, st);
      case 6: return launch_rows48_forward_ct<6>(e, a, G, rows48_parts_, st);
      case 7: return launch_rows48_forward_ct<7>(e, a, G, rows48_parts_, st);
      case 8: return launch_rows48_forward_ct<8>(e, a, G, rows48_parts_, st);
      case 9: return launch_rows48_forward_ct<9>(e, a, G, rows48_parts_, st);
      case 10: return launch_rows48_forward_ct<10>(e, a, G, rows48_parts_, st);
      case 11: return launch_rows48_forward_ct<11>(e, a, G, rows48_parts_, st);
      default: return fail(e, SDRM_ERR_SHAPE, "row-owned forward: padded width outside 128..352");
    }
  }
  switch (e->WP / 32) {
    case 4: return launch_row_forward_ct<4>(e, a, G, st);
    case 5: return launch_row_forward_ct<5>(e, a, G, st);
    case 6: return launch_row_forward_ct<6>(e, a, G, st);
    case 7: return launch_row_forward_ct<7>(e, a, G, st);
    case 8: return launch_row_forward_ct<8>(e, a, G, st);
    case 9: return launch_row_forward_ct<9>(e, a, G, st);
    case 10: return launch_row_forward_ct<10>(e, a, G, st);
    case 11: return launch_row_forward_ct<11>(e, a, G, st);
    default: return fail(e, SDRM_ERR_SHAPE, "row-owned forward: padded width outside 128..352");
  }
}

// the narrow nets' train step (csrc/skinny_step.h): forward (which = 0) / backward (which = 1) over G = ceil(B / 16) user groups
SkStepArgs sk_step_args(sdrm_engine* e, int B) {
  SkStepArgs a{};
  const int n = e->T + 1;
  a.W0c = e->W0c; a.K0 = e->K0; a.Whc = e->Whc; a.Woc = e->Woc; a.bh = e->bhc; a.bo = e->boc;
  a.WhcT = e->WhcT; a.WocT = e->WocT; a.B0tab = e->B0tab;
  a.WeP = e->WeP; a.W0eP = e->W0eP; a.be = e->p + e->off_be; a.b0 = e->b0c; a.TPe = e->TPe; a.intab = e->WeP ? 1 : 0;
  a.slope0 = slope_ptr(e, 0); a.slopeh = e->H > 0 ? slope_ptr(e, 1) : slope_ptr(e, 0);
  a.sqrt_ab = e->sched + 3 * n; a.one_minus_ab = e->sched + 4 * n; a.tembP = e->tembP;
  a.B = B; a.L = e->L; a.W = e->W; a.T = e->T; a.H = e->H; a.G = (B + SK_USERS - 1) / SK_USERS;
  a.LPs = e->LP; a.WPs = e->WP; a.TPs = e->TP;
  a.U = e->U; a.tdev = e->tdev; a.pre = e->pre; a.pre_stride = (size_t)e->MPmax * e->WP; a.Y = e->Y;
  a.loss_part = e->loss_part; a.NP = a.G;
  a.slab0 = e->slab0; a.slabH = e->slabH; a.slabO = e->slabO; a.db0s = e->db0s; a.dbHs = e->dbHs; a.dbOs = e->dbOs;
  a.alpha_part = e->alpha_part; a.alpha_part_stride = e->alpha_part_stride;
  return a;
}

template <int NL, int NW>
int launch_sk_step_nlnw(sdrm_engine* e, const SkStepArgs& ka, int which, int grid, hipStream_t st) {
  typedef SkCfg<NL, NW> C;
  const size_t lds = (which == 0 ? sk_fwd_lds_floats<NL, NW>(ka.intab ? ka.TPe : 0) : sk_bwd_lds_floats<NL, NW>(ka.TPs)) * sizeof(float);
  if (lds > 48 * 1024) HIP_TRY(e, allow_full_lds(which == 0 ? (const void*)k_skinny_fwd<NL, NW> : (const void*)k_skinny_bwd<NL, NW>));
  if (which == 0) SDRM_LAUNCH(e, (k_skinny_fwd<NL, NW>), dim3((unsigned)grid), dim3(C::NTHR), lds, st, ka);
  else SDRM_LAUNCH(e, (k_skinny_bwd<NL, NW>), dim3((unsigned)grid), dim3(C::NTHR), lds, st, ka);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

// the narrow nets' forward on 4-row MFMA units (csrc/skinny_fwd4.h): one work-group per 4 users; false: the shape's LDS image does
// not fit (the caller takes k_skinny_fwd)
template <int NL, int NW>
int launch_sk_fwd4_nlnw(sdrm_engine* e, const SkStepArgs& ka, hipStream_t st, bool* done) {
  const size_t lds = sk4_fwd_lds_floats<NL, NW>(ka.intab ? ka.TPe : 0) * sizeof(float);
  *done = false;
  if (lds > 160 * 1024) return SDRM_OK;
  if (lds > 48 * 1024) HIP_TRY(e, allow_full_lds((const void*)k_skinny_fwd4<NL, NW>));
  SDRM_LAUNCH(e, (k_skinny_fwd4<NL, NW>), dim3((unsigned)std::min(ka.NP, 8192)), dim3(SK4_THREADS), lds, st, ka);
  HIP_TRY(e, hipGetLastError());
  *done = true;
  return SDRM_OK;
}

int launch_sk_fwd4(sdrm_engine* e, const SkStepArgs& ka, hipStream_t st, bool* done) {
  const int NL = (e->L + 15) / 16, NW = (e->W + 15) / 16;
#define SK4_ROW(nl)                                                           \
  switch (NW) {                                                               \
    case 1: return launch_sk_fwd4_nlnw<nl, 1>(e, ka, st, done);               \
    case 2: return launch_sk_fwd4_nlnw<nl, 2>(e, ka, st, done);               \
    case 3: return launch_sk_fwd4_nlnw<nl, 3>(e, ka, st, done);               \
    default: return launch_sk_fwd4_nlnw<nl, 4>(e, ka, st, done);              \
  }
  switch (NL) {
    case 1: SK4_ROW(1)
    case 2: SK4_ROW(2)
    case 3: SK4_ROW(3)
    default: SK4_ROW(4)
  }
#undef SK4_ROW
}

int launch_sk_step(sdrm_engine* e, const SkStepArgs& ka, int which, int grid, hipStream_t st) {
  const int NL = (e->L + 15) / 16, NW = (e->W + 15) / 16;   // tiles with real columns (1..4 each)
#define SK_ROW(nl)                                                                  \
  switch (NW) {                                                                     \
    case 1: return launch_sk_step_nlnw<nl, 1>(e, ka, which, grid, st);              \
    case 2: return launch_sk_step_nlnw<nl, 2>(e, ka, which, grid, st);              \
    case 3: return launch_sk_step_nlnw<nl, 3>(e, ka, which, grid, st);              \
    default: return launch_sk_step_nlnw<nl, 4>(e, ka, which, grid, st);             \
  }
  switch (NL) {
    case 1: SK_ROW(1)
    case 2: SK_ROW(2)
    case 3: SK_ROW(3)
    default: SK_ROW(4)
  }
#undef SK_ROW
}

// eps-net layers 1..H and the output pre-activation inputs; layer 0 is launched by the caller
// (its bias / K differ between training and sampling).
// post_act (the sampler): every buffer holds the layer's ACTIVATION - the producer's epilogue applied PReLU, the operand is
// loaded as it is; otherwise (training, which needs the pre-activations for its backward) PReLU is applied on operand load.
// what a forward reads of the net: the live compute copies, or the sampler's snapshot of them
struct NetView { const float *W0c, *Whc, *Woc, *bhc, *boc, *B0tab, *slope0, *slopeh; };
NetView snapshot_view(const sdrm_engine* e) {
  const float* b = e->smp_w;
  return NetView{b + e->smp_off[0], b + e->smp_off[1], b + e->smp_off[2], b + e->smp_off[3], b + e->smp_off[4], b + e->smp_off[5],
                 b + e->smp_off[6], b + e->smp_off[6] + 1};
}

int hidden_forward(sdrm_engine* e, int MP, int rows, hipStream_t st, int cfg, int r0 = 0, int cls = PC_FWD_HIDDEN,
                   bool post_act = false, const NetView* nv = nullptr) {
  const double fl = 2.0 * rows * (double)e->W * e->W;
  const size_t ro = (size_t)r0 * e->WP;
  for (int k = 1; k <= e->H; ++k) {
    if (post_act) {
      GemmArgs a{};
      // (nv: a sampling step - the snapshot's weights, the sampler's own layer buffers)
      a.C = (nv ? smp_buf(e, k) : pre_buf(e, k)) + ro; a.ldc = e->WP; a.bias = nv ? nv->bhc : e->bhc; a.slopeE = nv ? nv->slopeh : slope_ptr(e, k);
      HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_PRELU>(a, (nv ? smp_buf(e, k - 1) : pre_buf(e, k - 1)) + ro, e->WP, nv ? nv->Whc : e->Whc, e->WP, MP, e->WP, e->WP,
                                                         st, Prof{e, cls, fl}, cfg)));
      continue;
    }
    GemmArgs a{};
    a.C = pre_buf(e, k) + ro; a.ldc = e->WP; a.bias = e->bhc; a.slopeA = slope_ptr(e, k - 1);
    HIP_TRY(e, (gemm_forward<XF_PRELU, EPI_BIAS>(a, pre_buf(e, k - 1) + ro, e->WP, e->Whc, e->WP, MP, e->WP, e->WP, st,
                                                 Prof{e, cls, fl}, cfg)));
  }
  return SDRM_OK;
}

// Chains still running on the auxiliary streams are folded back into `st` before anything else touches
// the engine's buffers or parameters.
int join_chains(sdrm_engine* e, hipStream_t st) {
  e->bwd_begun = false;   // a backward that was begun but never finished is abandoned
  if (!e->chains_pending) return SDRM_OK;
  for (int c = 0; c < e->n_aux; ++c) {
    HIP_TRY(e, hipEventRecord(e->ev_join[c], e->aux[c]));
    HIP_TRY(e, hipStreamWaitEvent(st, e->ev_join[c], 0));
  }
  e->chains_pending = false;
  return SDRM_OK;
}

// The row chains of a sampling call in progress wait for everything queued on `st` so far (the end of a train step that ran between
// two of its steps): their launches would otherwise share the chip with the step's one-work-group-per-CU kernels, which then end a
// whole work-group time later (measured: 8850 -> 8565 steps/s with no dependency at all).
int hold_chains(sdrm_engine* e, hipStream_t st) {
  if (!e->chains_pending) return SDRM_OK;
  if (e->hold_recorded) {   // the train step left its own mark: behind its weight gradients (hold_point)
    for (int c = 0; c < e->n_aux; ++c) HIP_TRY(e, hipStreamWaitEvent(e->aux[c], e->ev_hold, 0));
    e->hold_recorded = false;
    return SDRM_OK;
  }
  HIP_TRY(e, hipEventRecord(e->ev_fork, st));
  for (int c = 0; c < e->n_aux; ++c) HIP_TRY(e, hipStreamWaitEvent(e->aux[c], e->ev_fork, 0));
  return SDRM_OK;
}
// ... and the point they wait for: behind the step's last MFMA kernel, the weight gradients.  What follows - the tail's two launches -
// is latency- and HBM-bound on a fraction of the chip; the chains' next GEMMs run beside it (they read the call's snapshot of the net,
// the tail writes the live parameters).
int hold_point(sdrm_engine* e, hipStream_t st) {
  if (!e->chains_pending || !e->tune.hold_early || !e->hold_needed) return SDRM_OK;
  HIP_TRY(e, hipEventRecord(e->ev_hold, st));
  e->hold_recorded = true;
  return SDRM_OK;
}

// Row chains for a sampling call of n rows (rows are independent through the whole reverse loop, so a row range can run as a chain
// of launches on a stream of its own; the ramp and drain of one chain's launch are then filled by the other's).  Measured in round 5
// (tools/ab/chain_ab.py, profiles/r05_sampler_chains.txt; us per reverse step of the ML-1M net, whole calls): two chains win from
// ~2700 rows on - 2715 rows 30.4 -> 27.3, 4096 rows 40.3 -> 36.1, 5429 rows 48.8 -> 44.4, 8192 rows 67.3 -> 59.5 - and lose below
// (1358 rows 20.1 -> 22.9: a launch of half the rows no longer fills the chip).  Three chains: 42.6 at 5429 rows, worse elsewhere, and
// worse inside a job that trains between sampling steps; more than two are not what the host's launch rate bounds
// (tools/ab/chain_threads.py: one host thread per chain gives the same 43.4; a captured graph with forked streams replays at 65).
// The rule: two chains once the call has 2560 x 352 elements per layer, i.e. each chain's launch still has ~130 work-groups.
// A smaller call is one chain on the caller's stream - until a train step is queued beside it: from then on the chain runs on an
// auxiliary stream (n_aux = 1, "detached", sdrm_train_forward; the caller's stream then carries the train steps): a train step of the per-layer path - eleven dependent launches of 5-13 us on a mostly idle chip -
// and the four launches of a small sampling step are both latency chains, and two latency chains on two streams fill each other's
// gaps.  Such a train step does not hold the chains (hold_chains is for row-owned steps, whose one-work-group-per-CU kernels lose a
// whole work-group time to a busy CU).
// A train step between two sampling steps (bench.py's walk) does not join the chains: the sampler runs in layer buffers of its own, what
// the chains have queued may finish beside the start of the step, and their next launches wait for its end (hold_chains).  With no
// dependency at all - chains running on beside the whole train step - a launch of one work-group per CU finds some CUs busy with the
// other stream's work-groups and ends a whole work-group time later: 8850 -> 8565 steps/s.
// While an event profile is recorded (sdrm_profile_begin) the chains run one after the other on the caller's stream: intervals of
// launches that share the chip would overlap and say nothing about either kernel.
int chains_for(const Tuning& t, int n, int WP) {
  if (t.chains >= 1) return t.chains > 4 ? 4 : t.chains;
  return (size_t)n * (size_t)WP >= (size_t)2560 * 352 ? 2 : 1;
}

int upload_schedule(sdrm_engine* e, float beta1, float beta2) {
  const int T = e->T, n = T + 1;
  // fp32 arithmetic in the reference's op order (train_SDRM.py:300-303): linspace, affine, 1-b, log, cumsum, exp
  std::vector<float> b(n), al(n), ab(n), sq(n), om(n);
  const float step = 1.0f / (float)T;  // torch.linspace(0,1,T+1): start + i*step for the lower half, end - (n-1-i)*step above
  float run = 0.f;
  for (int i = 0; i < n; ++i) {
    const float lin = (i < n / 2) ? (0.f + step * (float)i) : (1.f - step * (float)(n - 1 - i));
    b[i] = (beta2 - beta1) * lin + beta1;
    al[i] = 1.f - b[i];
    run += std::log(al[i]);
    ab[i] = std::exp(run);
  }
  ab[0] = 1.f;
  for (int i = 0; i < n; ++i) { sq[i] = std::sqrt(ab[i]); om[i] = 1.f - ab[i]; }
  e->h_beta = b; e->h_alpha = al; e->h_alphabar = ab;
  HIP_TRY(e, hipMemcpy(e->sched + 0 * n, b.data(), n * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(e->sched + 1 * n, al.data(), n * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(e->sched + 2 * n, ab.data(), n * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(e->sched + 3 * n, sq.data(), n * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(e->sched + 4 * n, om.data(), n * 4, hipMemcpyHostToDevice));
  if (e->rev_dev) {   // (1-alpha)/sqrt(1-alphabar), sqrt(alpha), sqrt(beta) per step, fp32 like the reference (:23-24)
    std::vector<float> rev(3 * (size_t)n, 0.f);
    for (int i = 1; i < n; ++i) {
      rev[i] = (1.f - al[i]) / std::sqrt(1.f - ab[i]);
      rev[n + i] = std::sqrt(al[i]);
      rev[2 * n + i] = std::sqrt(b[i]);
    }
    HIP_TRY(e, hipMemcpy(e->rev_dev, rev.data(), rev.size() * 4, hipMemcpyHostToDevice));
  }
  return SDRM_OK;
}

int upload_temb(sdrm_engine* e) {
  const int T = e->T, half = T / 2;
  std::vector<float> tab((size_t)(T + 1) * T, 0.f), freqs(half);
  // train_SDRM.py:105-112 in fp32: exp(-ln(1e4) * k / half)
  const float nl = -(float)std::log(10000.0);
  for (int k = 0; k < half; ++k) freqs[k] = std::exp(nl * (float)k / (float)half);
  for (int t = 0; t <= T; ++t)
    for (int k = 0; k < half; ++k) {
      const float arg = (float)t * freqs[k];
      tab[(size_t)t * T + k] = std::cos(arg);
      tab[(size_t)t * T + half + k] = std::sin(arg);
    }
  HIP_TRY(e, hipMemcpy(e->temb, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  // rows padded to TP floats: what the train step's staging copies into the trailing columns of the layer-0 operand
  HIP_TRY(e, hipMemcpy2D(e->tembP, (size_t)e->TP * 4, tab.data(), (size_t)T * 4, (size_t)T * 4, (size_t)(T + 1), hipMemcpyHostToDevice));
  return SDRM_OK;
}

void reverse_coeffs(const sdrm_engine* e, int i, float& c1, float& sa, float& sb) {
  const float a = e->h_alpha[i], ab = e->h_alphabar[i], b = e->h_beta[i];
  c1 = (1.f - a) / std::sqrt(1.f - ab);
  sa = std::sqrt(a);
  sb = std::sqrt(b);
}

}  // namespace

// =================================================================================================
extern "C" {

int sdrm_debug_set_rowchain(sdrm_engine* e, int mode) {
  if (!e) return SDRM_ERR_ARG;
  if (e->bwd_begun) return fail(e, SDRM_ERR_STATE, "sdrm_debug_set_rowchain: a two-call backward is in progress");
  e->fwd_done = false;   // a pending forward's stacked row order belongs to the old setting
  e->tune.rowchain = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return SDRM_OK;
}

int sdrm_debug_set_rows48(sdrm_engine* e, int mode) {
  if (!e) return SDRM_ERR_ARG;
  if (e->bwd_begun) return fail(e, SDRM_ERR_STATE, "sdrm_debug_set_rows48: between sdrm_train_backward_begin and _finish");
  e->fwd_done = false;   // a pending forward's stacked row order belongs to the old setting
  e->tune.rows48 = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return SDRM_OK;
}

int sdrm_debug_set_rows48_split(sdrm_engine* e, int mode) {
  if (!e) return SDRM_ERR_ARG;
  if (e->bwd_begun) return fail(e, SDRM_ERR_STATE, "sdrm_debug_set_rows48_split: between sdrm_train_backward_begin and _finish");
  if (mode != 0 && mode != 1 && mode != 2 && mode != 4) return fail(e, SDRM_ERR_ARG, "sdrm_debug_set_rows48_split: 0, 1, 2 or 4");
  e->fwd_done = false;
  e->tune.split = mode;
  return SDRM_OK;
}

int sdrm_debug_rows48_split_available(const sdrm_engine* e) { return e && e->W0f && e->xcd_ok ? 1 : 0; }

int sdrm_debug_set_rows48_share(sdrm_engine* e, int on) {
  if (!e) return SDRM_ERR_ARG;
  e->tune.rows48_share = on == 1 ? 7 : (on == 2 ? 5 : (on == 3 ? 2 : 0));   // 1: forward (with its sweep) and chain, 2: the forward only, 3: the chain only
  return SDRM_OK;
}

int sdrm_debug_set_sample_persist(sdrm_engine* e, int mode) {
  if (!e) return SDRM_ERR_ARG;
  if (e->smp.active) return fail(e, SDRM_ERR_STATE, "sdrm_debug_set_sample_persist: inside a sampling call");
  e->tune.smp_persist = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return SDRM_OK;
}

int sdrm_debug_split_skew(sdrm_engine* e, unsigned skew) {
  if (!e) return SDRM_ERR_ARG;
  e->xskew = skew;
  return SDRM_OK;
}

int sdrm_debug_rowchain_available(const sdrm_engine* e) { return e && e->W0f ? 1 : 0; }

int sdrm_debug_set_wgrad_strips(sdrm_engine* e, int on) {
  if (!e) return SDRM_ERR_ARG;
  e->tune.strips = on ? 1 : 0;
  return SDRM_OK;
}

int sdrm_debug_set_dgrad_rows(sdrm_engine* e, int on) {
  if (!e) return SDRM_ERR_ARG;
  e->fwd_done = false;   // a pending row-owned forward chose what it stores (pre-activations or not) by the old setting
  e->tune.dgrad_rows = on < 0 ? 0 : (on > 2 ? 2 : on);
  return SDRM_OK;
}

int sdrm_debug_set_skinny(sdrm_engine* e, int on) {
  if (!e) return SDRM_ERR_ARG;
  e->tune.skinny = on < 0 ? 0 : (on > 2 ? 2 : on);
  return SDRM_OK;
}

int sdrm_debug_plan_wgrad(int rows, int n_out, int k_in, int* slices, int* rows_per_slice) {
  if (!slices || !rows_per_slice || rows < 1 || n_out < 1 || k_in < 1) return SDRM_ERR_ARG;
  pick_splits(Tuning{}, rows, wgrad_tiles(Tuning{}, n_out, k_in), *slices, *rows_per_slice);
  return SDRM_OK;
}

int sdrm_debug_philox_draws(sdrm_engine* e, uint64_t seed, uint32_t purpose, uint32_t step, int64_t row0, int rows, int quads,
                            float* normals, uint8_t* lowbits, void* stream) {
  if (!e || !normals || rows < 1 || quads < 1) return fail(e, SDRM_ERR_ARG, "sdrm_debug_philox_draws: bad argument");
  const size_t total = (size_t)rows * quads;
  SDRM_LAUNCH(e, k_philox_draws, dim3((unsigned)std::min<size_t>(8192, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
              (uint32_t)seed, (uint32_t)(seed >> 32), purpose, step, row0, rows, quads, normals, lowbits);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

int sdrm_debug_set_fused_reverse(sdrm_engine* e, int mode) {
  if (!e) return SDRM_ERR_ARG;
  e->tune.fuse_rev = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return SDRM_OK;
}

int sdrm_debug_set_gradient_buckets(sdrm_engine* e, int buckets) {
  if (!e || (buckets != 1 && buckets != 2)) return SDRM_ERR_ARG;
  e->tune.ar_buckets = buckets;
  return SDRM_OK;
}

int sdrm_debug_set_chains(sdrm_engine* e, int chains) {
  if (!e) return SDRM_ERR_ARG;
  e->tune.chains = chains;
  return SDRM_OK;
}

int sdrm_debug_chains(const sdrm_engine* e) { return e ? e->n_chains : 0; }

int sdrm_debug_set_tile(sdrm_engine* e, int cfg) {
  if (!e) return SDRM_ERR_ARG;
  if (e->bwd_begun || e->smp.active)
    return fail(e, SDRM_ERR_STATE, "sdrm_debug_set_tile: a two-call backward or a sampling call is in progress");
  e->fwd_done = false;   // the slope-partial layout of a backward is fixed by its forward's tile: a pending forward is dropped
  e->tune.force_cfg = cfg;
  return SDRM_OK;
}

int sdrm_debug_set_nt32_rows(sdrm_engine* e, int max_rows, int max_rows_train) {
  if (!e) return SDRM_ERR_ARG;
  if (max_rows >= 0) e->tune.nt32_max_rows = max_rows;
  if (max_rows_train >= 0) e->tune.nt32_max_rows_train = max_rows_train;
  return SDRM_OK;
}

#ifndef SDRM_SOURCE_HASH
#define SDRM_SOURCE_HASH "unhashed"
#endif
// "SDRM_SOURCE_HASH=<hex>" is also what sdrm_amd/_build.py looks for in the file to decide whether the binary belongs to
// the sources next to it.
static const char kSourceHashMarker[] = "SDRM_SOURCE_HASH=" SDRM_SOURCE_HASH;
const char* sdrm_source_hash(void) { return kSourceHashMarker + sizeof("SDRM_SOURCE_HASH=") - 1; }

const char* sdrm_build_info(void) {
  return "gfx950 fp32 v_mfma_f32_32x32x2_f32; block tile 64x64x16 (32x32x32 on v_mfma_f32_16x16x4_f32 for NT launches of at "
         "most 4096 rows, 8192 stacked rows in the train step), 4 waves, two LDS stages + register-double-buffered fragments, "
         "pipeline pieces in the MFMA shadows, k-minor LDS + ds_read_b128 for NT; split-K slabs for wgrad; persistent "
         "LDS-resident kernels for widths <= 64; one-round grids sized for 256 compute units in 8 XCDs (sdrm_create refuses any "
         "other device); sources " SDRM_SOURCE_HASH;
}

const char* sdrm_last_error(const sdrm_engine* e) { return e ? e->err.c_str() : "null engine"; }

// The chip this library is written for: gfx950 with all 256 compute units in one device (MI355X, SPX mode).  The one-round grids
// (k_row_fwd, k_dgrad_chain, k_wgrad_strips: one work-group per CU), the split-K plans (1280 resident work-groups) and the
// work-group -> XCD mapping (block b on XCD b & 7, xcd_remap) are sized for exactly that; on another CU count they would run,
// silently mis-balanced.  sdrm_create refuses instead (SDRM_ALLOW_ANY_DEVICE=1 lifts the CU-count check for experiments).
constexpr int SDRM_TARGET_CUS = 256;
int sdrm_debug_device_check(const char* gcn_arch, int compute_units) {
  if (!gcn_arch) return SDRM_ERR_ARG;
  if (std::strncmp(gcn_arch, "gfx950", 6) != 0 || (gcn_arch[6] != '\0' && gcn_arch[6] != ':')) return SDRM_ERR_DEVICE;
  return compute_units == SDRM_TARGET_CUS ? SDRM_OK : SDRM_ERR_DEVICE;
}
int64_t sdrm_param_count(const sdrm_engine* e) { return e ? e->P : -1; }

int sdrm_create(int L, int W, int T, int H, int max_rows, int device_id, sdrm_engine** out) {
  if (!out) return SDRM_ERR_ARG;
  *out = nullptr;
  if (L < 1 || L > 4096 || W < 1 || W > 4096 || T < 2 || T > 1024 || H < 0 || H > 16 || max_rows < 1 ||
      max_rows > (1 << 22))
    return SDRM_ERR_SHAPE;
  // k_tail_emb (csrc/tail.h) keeps TE_JT rows of emb_layer.weight beside a [TE_RB][TP] block of M in LDS: the image grows with T and
  // passes the CU's 160 KB at T = 1021 (the reference's search space ends at T = 198, hyperparameter_search.py:108)
  if (tail_emb_lds_floats(T, round_up(T + 1, 32)) * sizeof(float) > 160 * 1024) return SDRM_ERR_SHAPE;
  sdrm_engine* e = new sdrm_engine();
  // the only place the environment is read: the settings then belong to this handle
  if (const char* env = std::getenv("SDRM_TILE")) e->tune.force_cfg = std::atoi(env);
  if (const char* env = std::getenv("SDRM_CHAINS")) e->tune.chains = std::atoi(env);
  if (const char* env = std::getenv("SDRM_HOLD_EARLY")) e->tune.hold_early = std::atoi(env);
  if (const char* env = std::getenv("SDRM_DETACH")) e->tune.detach = std::atoi(env);
  if (const char* env = std::getenv("SDRM_FUSE_REV")) e->tune.fuse_rev = std::atoi(env);
  if (const char* env = std::getenv("SDRM_NT32_MAX_ROWS")) e->tune.nt32_max_rows = std::atoi(env);
  if (const char* env = std::getenv("SDRM_NT32_MAX_ROWS_TRAIN")) e->tune.nt32_max_rows_train = std::atoi(env);
  if (const char* env = std::getenv("SDRM_WGRAD_BLOCKS")) e->tune.wgrad_blocks = std::max(1, std::atoi(env));
  if (const char* env = std::getenv("SDRM_WGRAD_ROUND")) e->tune.wgrad_round = std::max(1, std::atoi(env));
  if (const char* env = std::getenv("SDRM_AR_BUCKETS")) e->tune.ar_buckets = std::atoi(env);
  if (const char* env = std::getenv("SDRM_WGRAD_SLICES")) e->tune.wgrad_slices = std::min(S_MAX, std::max(0, std::atoi(env)));
  if (const char* env = std::getenv("SDRM_ROWCHAIN")) e->tune.rowchain = std::atoi(env);
  if (const char* env = std::getenv("SDRM_ROWS48")) e->tune.rows48 = std::atoi(env);
  if (const char* env = std::getenv("SDRM_ROWS48_SPLIT")) e->tune.split = std::atoi(env);
  if (const char* env = std::getenv("SDRM_SAMPLE_PERSIST")) e->tune.smp_persist = std::atoi(env);
  if (const char* env = std::getenv("SDRM_ROWS48_SHARE")) e->tune.rows48_share = std::atoi(env);
  if (const char* env = std::getenv("SDRM_WGRAD_STRIPS")) e->tune.strips = std::atoi(env);
  if (const char* env = std::getenv("SDRM_DGRAD_ROWS")) e->tune.dgrad_rows = std::atoi(env);
  e->L = L; e->W = W; e->T = T; e->H = H; e->max_rows = max_rows; e->device = device_id;
  e->LP = round_up(L, 32); e->WP = round_up(W, 32); e->TP = round_up(T + 1, 32); e->K0 = e->LP + e->TP;
  e->MPmax = round_up(RC_ROWS * ((max_rows + RC_USERS - 1) / RC_USERS), 128);   // either stacked row order fits
  int64_t o = 0;
  e->off_we = o; o += (int64_t)T * T;
  e->off_be = o; o += T;
  e->off_w0 = o; o += (int64_t)W * (L + T);
  e->off_b0 = o; o += W;
  e->off_a0 = o; o += 1;
  if (H >= 1) {
    e->off_wh = o; o += (int64_t)W * W;
    e->off_bh = o; o += W;
    e->off_ah = o; o += 1;
  } else {
    e->off_wh = e->off_bh = e->off_ah = -1;
  }
  e->off_wo = o; o += (int64_t)L * W;
  e->off_bo = o; o += L;
  e->P = o;
  *out = e;  // handed out even on failure below so the caller can read the message and destroy
  HIP_TRY(e, hipSetDevice(device_id));
  {
    hipDeviceProp_t prop;
    HIP_TRY(e, hipGetDeviceProperties(&prop, device_id));
    const char* any = std::getenv("SDRM_ALLOW_ANY_DEVICE");
    const bool lifted = any && std::atoi(any) != 0;
    if (sdrm_debug_device_check(prop.gcnArchName, lifted ? SDRM_TARGET_CUS : prop.multiProcessorCount) != SDRM_OK)
      return fail(e, SDRM_ERR_DEVICE, std::string("sdrm_create: device ") + std::to_string(device_id) + " is " + prop.gcnArchName + " with " +
                                          std::to_string(prop.multiProcessorCount) + " compute units; this library is built for gfx950 with " +
                                          std::to_string(SDRM_TARGET_CUS) + " (MI355X, SPX mode): its one-round grids and XCD mapping assume them");
  }
  const size_t MP = e->MPmax;
  const int n = T + 1;
  HIP_TRY(e, dalloc(&e->p, e->P)); HIP_TRY(e, dalloc(&e->m, e->P));
  HIP_TRY(e, dalloc(&e->v, e->P)); HIP_TRY(e, dalloc(&e->g, e->P));
  HIP_TRY(e, dalloc(&e->W0c, (size_t)round_up(e->WP, 128) * e->K0)); HIP_TRY(e, dalloc(&e->b0c, e->WP));
  HIP_TRY(e, dalloc(&e->Whc, (size_t)round_up(e->WP, 128) * e->WP)); HIP_TRY(e, dalloc(&e->bhc, e->WP));
  HIP_TRY(e, dalloc(&e->Woc, (size_t)round_up(e->LP, 128) * e->WP)); HIP_TRY(e, dalloc(&e->boc, e->LP));
  HIP_TRY(e, dalloc(&e->WhcT, (size_t)round_up(e->WP, 128) * e->WP)); HIP_TRY(e, dalloc(&e->WocT, (size_t)round_up(e->WP, 128) * e->LP));
  if (L == W && e->WP >= 128 && e->WP <= 352) {   // the row-owned forward's shape envelope (rowchain.h); L == W (every call site of the
                                                  // reference): the ones column round_up(W, 4) must be a pad column of U as well
    HIP_TRY(e, dalloc(&e->W0f, (size_t)e->WP * e->WP)); HIP_TRY(e, dalloc(&e->Whf, (size_t)e->WP * e->WP));
    HIP_TRY(e, dalloc(&e->Wof, (size_t)e->WP * e->WP));
    HIP_TRY(e, dalloc(&e->WhfT, (size_t)e->WP * e->WP)); HIP_TRY(e, dalloc(&e->WofT, (size_t)e->WP * e->WP));
    HIP_TRY(e, dalloc(&e->act, (size_t)(H + 1) * e->MPmax * e->WP));
    if (round_up(W, 4) < e->WP) e->ones_col = round_up(W, 4);
  }
  HIP_TRY(e, dalloc(&e->temb, (size_t)n * T)); HIP_TRY(e, dalloc(&e->tembP, (size_t)n * e->TP));
  HIP_TRY(e, dalloc(&e->B0tab, (size_t)n * e->WP)); HIP_TRY(e, dalloc(&e->sched, (size_t)8 * n));
  HIP_TRY(e, dalloc(&e->rev_dev, (size_t)3 * n));
  {
    const size_t sz[7] = {(size_t)round_up(e->WP, 128) * e->K0, (size_t)round_up(e->WP, 128) * e->WP, (size_t)round_up(e->LP, 128) * e->WP,
                          (size_t)e->WP, (size_t)e->LP, (size_t)n * e->WP, 4};
    size_t tot = 0;
    for (int k = 0; k < 7; ++k) { e->smp_off[k] = tot; tot += (sz[k] + 63) / 64 * 64; }   // 256-byte aligned pieces
    HIP_TRY(e, dalloc(&e->smp_w, tot));
  }
  HIP_TRY(e, dalloc(&e->sel, 1));
  HIP_TRY(e, dalloc(&e->one_dev, 4));
  HIP_TRY(e, dalloc(&e->feed_flag, 4));
  if (e->W0f) {
    // column-split row groups (csrc/rows48.h): hand-shake counters, the host-visible abort word, and the check of the one property
    // of the chip the path rests on - work-group b of a launch runs on XCD b & 7 - with a probe launch
    HIP_TRY(e, dalloc(&e->xcntF, (size_t)256 * 32)); HIP_TRY(e, dalloc(&e->xcntC, (size_t)256 * 32));
    HIP_TRY(e, dalloc(&e->xcntS, (size_t)256 * 32));
    HIP_TRY(e, hipHostMalloc((void**)&e->xabort_host, sizeof(unsigned), hipHostMallocMapped));
    *e->xabort_host = 0u;
    HIP_TRY(e, hipHostGetDevicePointer((void**)&e->xabort_dev, e->xabort_host, 0));
    unsigned* probe = nullptr;
    constexpr int NPROBE = 512;
    HIP_TRY(e, dalloc(&probe, NPROBE));
    hipLaunchKernelGGL(k_xcc_probe, dim3(NPROBE), dim3(64), 0, 0, probe);
    std::vector<unsigned> xcc(NPROBE);
    HIP_TRY(e, hipMemcpy(xcc.data(), probe, NPROBE * sizeof(unsigned), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipFree(probe));
    bool ok = true;
    std::set<unsigned> ids;
    for (int b = 0; b < NPROBE; ++b) ok = ok && (xcc[b] & 0xfu) == (xcc[b & 7] & 0xfu);
    for (int x = 0; x < 8; ++x) ids.insert(xcc[x] & 0xfu);
    e->xcd_ok = ok && ids.size() == 8;
  }
  {
    const float one = 1.0f;
    HIP_TRY(e, hipMemcpy(e->one_dev, &one, 4, hipMemcpyHostToDevice));
    // the ones column: pad entry ones_col of the hidden layer's padded bias is 1 (its weight row there is zero, so the
    // pre-activation and the activation of that pad column are 1 in every row; the next layer's weight column there is zero, so
    // nothing else changes); layer 0's comes from B0tab (k_emb_tables), U's from the row-owned forward's staging
    if (e->ones_col >= 0) HIP_TRY(e, hipMemcpy(e->bhc + e->ones_col, &one, 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(e, dalloc(&e->U, MP * e->K0)); HIP_TRY(e, dalloc(&e->pre, (size_t)(H + 1) * MP * e->WP));
  HIP_TRY(e, dalloc(&e->Y, MP * e->LP)); HIP_TRY(e, dalloc(&e->dY, MP * e->LP));
  HIP_TRY(e, dalloc(&e->dA, (size_t)(H + 1) * MP * e->WP));   // dpre_buf(0..H)
  HIP_TRY(e, dalloc(&e->X, MP * e->LP)); HIP_TRY(e, dalloc(&e->Us, MP * e->LP));
  HIP_TRY(e, dalloc(&e->smp_pre, (size_t)(H + 1) * MP * e->WP)); HIP_TRY(e, dalloc(&e->smp_Y, MP * e->LP));
  HIP_TRY(e, dalloc(&e->slab0, (size_t)S_MAX * e->WP * e->K0)); HIP_TRY(e, dalloc(&e->db0s, (size_t)S_MAX * e->WP));
  HIP_TRY(e, dalloc(&e->slabO, (size_t)S_MAX * e->LP * e->WP)); HIP_TRY(e, dalloc(&e->dbOs, (size_t)S_MAX * e->LP));
  if (H >= 1) {
    HIP_TRY(e, dalloc(&e->slabH, (size_t)H * S_MAX * e->WP * e->WP));
    HIP_TRY(e, dalloc(&e->dbHs, (size_t)H * S_MAX * e->WP));
  }
  e->alpha_part_stride = std::max(max_gemm_blocks((int)MP, e->WP), (int)MP / 16);
  HIP_TRY(e, dalloc(&e->alpha_part, (size_t)(H + 1) * e->alpha_part_stride));
  HIP_TRY(e, dalloc(&e->loss_part, (size_t)4 * std::max<size_t>(LOSS_BLOCKS, 4 * (MP / SK_ROWS + 1)))); HIP_TRY(e, dalloc(&e->sums, 8));
  if (e->LP <= 64 && e->WP <= 64 && T <= 128) {
    e->TPe = round_up(T, 16);
    HIP_TRY(e, dalloc(&e->WeP, (size_t)e->TPe * e->TPe)); HIP_TRY(e, dalloc(&e->W0eP, (size_t)e->WP * e->TPe));
  }
  HIP_TRY(e, dalloc(&e->Mred, (size_t)W * e->TP)); HIP_TRY(e, dalloc(&e->snap, (size_t)T * T + T + (size_t)W * T));
  HIP_TRY(e, allow_full_lds((const void*)k_tail_emb));
  HIP_TRY(e, dalloc(&e->tdev, max_rows)); HIP_TRY(e, dalloc(&e->Tj_dev, max_rows)); HIP_TRY(e, dalloc(&e->rowid_dev, max_rows));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_hold, hipEventDisableTiming));
  for (int c = 0; c < 3; ++c) {
    HIP_TRY(e, hipStreamCreateWithFlags(&e->aux[c], hipStreamNonBlocking));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join[c], hipEventDisableTiming));
  }
  int rc = upload_schedule(e, 1e-4f, 0.02f);
  if (rc) return rc;
  rc = upload_temb(e);
  if (rc) return rc;
  HIP_TRY(e, hipDeviceSynchronize());
  return SDRM_OK;
}

int sdrm_destroy(sdrm_engine* e) {
  if (!e) return SDRM_ERR_ARG;
  (void)hipSetDevice(e->device);
  void* bufs[] = {e->p, e->m, e->v, e->g, e->W0c, e->b0c, e->Whc, e->bhc, e->Woc, e->boc, e->temb, e->tembP, e->B0tab,
                  e->sched, e->U, e->pre, e->Y, e->dY, e->dA, e->X, e->slab0, e->slabH, e->slabO, e->db0s,
                  e->dbHs, e->dbOs, e->alpha_part, e->loss_part, e->sums, e->Mred, e->snap, e->WeP, e->W0eP, e->tdev, e->Tj_dev, e->rowid_dev, e->rev_dev, e->Us, e->WhcT, e->WocT, e->sel, e->one_dev, e->feed_flag, e->smp_w, e->W0f, e->Whf, e->Wof, e->act, e->WhfT, e->WofT, e->smp_pre, e->smp_Y};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (e->xcntF) (void)hipFree(e->xcntF);
  if (e->xcntC) (void)hipFree(e->xcntC);
  if (e->xcntS) (void)hipFree(e->xcntS);
  for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
  (void)hipDeviceSynchronize();
  if (e->xabort_host) (void)hipHostFree(e->xabort_host);
  (void)sdrm_comm_destroy(e);
  for (float* b : e->dec_buf)
    if (b) (void)hipFree(b);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_hold) (void)hipEventDestroy(e->ev_hold);
  for (int c = 0; c < 3; ++c) {
    if (e->ev_join[c]) (void)hipEventDestroy(e->ev_join[c]);
    if (e->aux[c]) (void)hipStreamDestroy(e->aux[c]);
  }
  delete e;
  return SDRM_OK;
}

int sdrm_set_schedule(sdrm_engine* e, float beta1, float beta2) {
  if (!e) return SDRM_ERR_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipDeviceSynchronize());
  return upload_schedule(e, beta1, beta2);
}

int sdrm_get_schedule(const sdrm_engine* e, float* b, float* a, float* ab) {
  if (!e || !b || !a || !ab) return SDRM_ERR_ARG;
  const size_t n = (size_t)e->T + 1;
  std::memcpy(b, e->h_beta.data(), n * 4);
  std::memcpy(a, e->h_alpha.data(), n * 4);
  std::memcpy(ab, e->h_alphabar.data(), n * 4);
  return SDRM_OK;
}

int sdrm_set_params(sdrm_engine* e, const float* flat, void* stream) {
  if (!e || !flat) return fail(e, SDRM_ERR_ARG, "sdrm_set_params: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (int jr = join_chains(e, st)) return jr;
  HIP_TRY(e, hipMemcpyAsync(e->p, flat, e->P * 4, hipMemcpyDeviceToDevice, st));
  return launch_adam(e, nullptr, 0.f, 0, st);  // re-pack only
}

int sdrm_get_params(const sdrm_engine* e, float* flat, void* stream) {
  if (!e || !flat) return SDRM_ERR_ARG;
  sdrm_engine* me = const_cast<sdrm_engine*>(e);
  HIP_TRY(me, hipMemcpyAsync(flat, e->p, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SDRM_OK;
}

const float* sdrm_params_ptr(const sdrm_engine* e) { return e ? e->p : nullptr; }

int sdrm_get_grads(const sdrm_engine* e, float* flat, void* stream) {
  if (!e || !flat) return SDRM_ERR_ARG;
  sdrm_engine* me = const_cast<sdrm_engine*>(e);
  HIP_TRY(me, hipMemcpyAsync(flat, e->grad_src ? e->grad_src : e->g, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SDRM_OK;
}

int sdrm_get_adam_state(const sdrm_engine* e, float* m, float* v, int64_t* step_host, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  sdrm_engine* me = const_cast<sdrm_engine*>(e);
  if (m) HIP_TRY(me, hipMemcpyAsync(m, e->m, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (v) HIP_TRY(me, hipMemcpyAsync(v, e->v, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (step_host) *step_host = e->adam_t;
  return SDRM_OK;
}

int sdrm_set_adam_state(sdrm_engine* e, const float* m, const float* v, int64_t step, void* stream) {
  if (!e || step < 0) return fail(e, SDRM_ERR_ARG, "sdrm_set_adam_state: bad argument");
  if (m) HIP_TRY(e, hipMemcpyAsync(e->m, m, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (v) HIP_TRY(e, hipMemcpyAsync(e->v, v, e->P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  e->adam_t = step;
  return SDRM_OK;
}

int sdrm_adam_reset(sdrm_engine* e, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  HIP_TRY(e, hipMemsetAsync(e->m, 0, e->P * 4, (hipStream_t)stream));
  HIP_TRY(e, hipMemsetAsync(e->v, 0, e->P * 4, (hipStream_t)stream));
  e->adam_t = 0;
  return SDRM_OK;
}

// ---------------------------------------------------------------------------------------------
namespace {
bool sample_persist_fits(const sdrm_engine* e, const SampleState& s);   // (below, beside the sampler)
}
int sdrm_train_forward(sdrm_engine* e, const float* x0, int B, int64_t row0, int mode, const sdrm_train_randoms* rnd,
                       uint64_t seed, uint64_t step, float nd, double* sums, void* stream) {
  if (!e || !x0) return fail(e, SDRM_ERR_ARG, "sdrm_train_forward: null pointer");
  if (B < 1 || B > e->max_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_train_forward: B outside [1, max_rows]");
  if (mode == SDRM_RNG_EXPLICIT && (!rnd || !rnd->noise || !rnd->t || !rnd->keep))
    return fail(e, SDRM_ERR_ARG, "sdrm_train_forward: EXPLICIT mode needs noise, t and keep");
  if (mode != SDRM_RNG_EXPLICIT && mode != SDRM_RNG_PHILOX) return fail(e, SDRM_ERR_ARG, "bad rng mode");
  hipStream_t st = (hipStream_t)stream;
  // (No join_chains here: a sampling call in progress reads its own snapshot of the net and runs in its own buffers - Us, X, smp_pre,
  // smp_Y - so what its row chains have queued on the auxiliary streams may finish beside the start of this step; the chains' NEXT
  // launches wait for the step's end: hold_chains, from the next sdrm_sample_steps - one cross-stream dependency per train step instead of
  // a join and a fork.)
  e->bwd_begun = false;   // a backward that was begun but never finished is abandoned
  if (!e->train_since_sample) e->hold_needed = false;   // (set below by a row-owned forward; kept over several train steps in a row)
  e->train_since_sample = true; e->hold_recorded = false;
  // A small sampling call (one chain) runs on the caller's stream until a train step is queued beside it - on a stream of its own a
  // plain call is ~3 us per step SLOWER (tools/ab/detach_ab.py: 679 rows 17.0 -> 19.7, ML-100k 58.1 -> 62.4).  From here on its chain
  // is detached (chains_for): the mark behind its last step is taken now, in front of this step's launches.
  if (e->tune.detach > 0 && e->smp.active && !e->smp.skinny && e->smp.i_next >= 1 && e->n_chains == 1 && e->n_aux == 0 && !e->detach_armed &&
      !e->prof_on && !sample_persist_fits(e, e->smp)) {
    HIP_TRY(e, hipEventRecord(e->ev_fork, st));
    e->detach_armed = true;
  }
  const int MP = round_up(3 * B, BM), n = e->T + 1;
  const int cfg = choose_cfg(e->tune, MP, e->tune.nt32_max_rows_train);   // one tile for every NT launch of the step
  e->fwd_done = false;

  e->cur_grouped = false; e->cur_act = false; e->cur_sk = false; e->cur_g16 = false; e->cur_rows = MP; e->cur_parts = 1; e->cur_skip_pre = false;
  if (int xs = split_status(e)) return xs;
  if (skinny_net(e)) {
    // narrow net (csrc/skinny_step.h): staging, all layers and the loss partial sums of 16 users' P, S, Q rows per work-group in
    // ONE launch; the tables B0tab = b0 + C0[t] come from the last step's tail (or are made now, after a parameter upload)
    int rc = e->WeP ? SDRM_OK : ensure_tables(e, st);   // (T <= 128: the forward makes its users' rows of the table itself)
    if (rc) return rc;
    SkStepArgs ka = sk_step_args(e, B);
    ka.x0 = x0;
    if (mode == SDRM_RNG_EXPLICIT) { ka.noise = rnd->noise; ka.t = rnd->t; ka.keep = rnd->keep; }
    ka.mode = mode; ka.seed_lo = (uint32_t)seed; ka.seed_hi = (uint32_t)(seed >> 32); ka.step = (uint32_t)step;
    ka.row0 = row0; ka.nd = nd;
    bool done4 = false;
    if (e->tune.skinny == 1) {   // 4 users per work-group (csrc/skinny_fwd4.h); 2: the 16-user forward
      ka.NP = 4 * ka.G;
      rc = launch_sk_fwd4(e, ka, st, &done4);
      if (rc) return rc;
    }
    if (!done4) {
      ka.NP = ka.G;
      rc = launch_sk_step(e, ka, 0, ka.G, st);
      if (rc) return rc;
    }
    e->cur_sk_np = ka.NP;
    if (!e->fold_sums) {
      SDRM_LAUNCH(e, k_loss_sums, dim3(1), dim3(256), 0, st, (const double*)e->loss_part, ka.NP, (double)B * (double)e->L,
                         sums ? sums : e->sums);
      HIP_TRY(e, hipGetLastError());
    }
    e->cur_B = B; e->cur_MP = round_up(SK_ROWS * ka.G, BM); e->cur_rows = SK_ROWS * ka.G; e->cur_x0 = x0; e->cur_sk = true; e->fwd_done = true;
    return SDRM_OK;
  }

  if (use_rowchain(e, B)) {
    // row-owned forward (rowchain.h): the step's tables (the launch also pulls the batch into L2), then staging + every layer +
    // the loss partial sums in ONE launch
    int rc = emb_tables(e, st, x0, (size_t)B * e->L);
    if (rc) return rc;
    const int G = (B + RC_USERS - 1) / RC_USERS, MPg = round_up(G * RC_ROWS, BM);
    rc = launch_row_forward(e, x0, B, row0, mode, rnd, seed, step, nd, G, st);
    if (rc) return rc;
    if (!e->fold_sums) {
      SDRM_LAUNCH(e, k_loss_sums, dim3(1), dim3(256), 0, st, (const double*)e->loss_part, G, (double)B * (double)e->L,
                         sums ? sums : e->sums);
      HIP_TRY(e, hipGetLastError());
    }
    e->cur_B = B; e->cur_MP = MPg; e->cur_rows = G * RC_ROWS; e->cur_x0 = x0; e->cur_grouped = true; e->cur_act = true; e->fwd_done = true;
    e->hold_needed = true;
    return SDRM_OK;
  }

  if (const int parts = rows48_parts(e, B)) {
    // the same on 48-row work-groups (rows48.h; `parts` of them per row group): tables, then ONE launch
    int rc = emb_tables(e, st, x0, (size_t)B * e->L);
    if (rc) return rc;
    const int G = (B + R48_USERS - 1) / R48_USERS, MPg = round_up(G * R48_ROWS, BM);
    rc = launch_row_forward(e, x0, B, row0, mode, rnd, seed, step, nd, G, st, parts);
    if (rc) return rc;
    if (!e->fold_sums) {
      SDRM_LAUNCH(e, k_loss_sums, dim3(1), dim3(256), 0, st, (const double*)e->loss_part, G * parts, (double)B * (double)e->L,
                         sums ? sums : e->sums);
      HIP_TRY(e, hipGetLastError());
    }
    e->cur_B = B; e->cur_MP = MPg; e->cur_rows = G * R48_ROWS; e->cur_x0 = x0; e->cur_g16 = true; e->cur_parts = parts; e->cur_act = true;
    e->hold_needed = true;
    e->fwd_done = true;
    return SDRM_OK;
  }

  PrepTrainArgs pa{};
  pa.x0 = x0;
  if (mode == SDRM_RNG_EXPLICIT) { pa.noise = rnd->noise; pa.t = rnd->t; pa.keep = rnd->keep; }
  pa.sqrt_ab = e->sched + 3 * n; pa.one_minus_ab = e->sched + 4 * n; pa.tembP = e->tembP;
  pa.U = e->U; pa.tdev = e->tdev;
  pa.B = B; pa.L = e->L; pa.LP = e->LP; pa.K0 = e->K0; pa.T = e->T; pa.MP = MP;
  pa.mode = mode; pa.seed_lo = (uint32_t)seed; pa.seed_hi = (uint32_t)(seed >> 32); pa.step = (uint32_t)step;
  pa.row0 = row0; pa.nd = nd;
  {
    // the step's tables ride on the staging launch: C0^T in the trailing columns of W0c (what the plain sdrm_forward multiplies its
    // one-hot(t) columns with) and B0tab = b0 + C0[t], which layer 0 below adds per row
    pa.emb = emb_args(e);
    e->tables_fresh = true;
    pa.emb_row0 = B + (MP - 3 * B);
    pa.emb_chunks = (e->WP + 63) / 64;
    pa.emb_blocks = (e->T + 1) * pa.emb_chunks;
    const int main_blocks = (int)(((int64_t)pa.emb_row0 * (e->K0 / 4) + 255) / 256);
    SDRM_LAUNCH(e, k_prep_train, dim3(pa.emb_blocks + main_blocks), dim3(256), 2 * e->T * sizeof(float), st, pa);
    HIP_TRY(e, hipGetLastError());
  }
  {
    // Layer 0 contracts over the latent columns only (K = LP instead of LP + TP: a fifth less work at ML-1M): every row's
    // time-embedding term is a row of B0tab, added in the epilogue.  The trailing columns of U carry temb[t_row] (round 4; before: a
    // one-hot(t)): the layer-0 weight gradient multiplies them to deliver M = dpre0^T * temb (csrc/tail.h).
    GemmArgs a{};
    a.C = pre_buf(e, 0); a.ldc = e->WP; a.bias = e->B0tab; a.ldtab = e->WP; a.trow = e->tdev; a.trow_B = B;
    HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_ROWTAB>(a, e->U, e->K0, e->W0c, e->K0, MP, e->WP, e->LP, st,
                                                       Prof{e, PC_FWD_L0, 2.0 * 3 * B * (double)e->W * e->L}, cfg)));   // K = the latents: the time-embedding term is a table row
  }
  int rc = hidden_forward(e, MP, 3 * B, st, cfg);
  if (rc) return rc;
  {
    GemmArgs a{};
    a.C = e->Y; a.ldc = e->LP; a.bias = e->boc; a.slopeA = slope_ptr(e, e->H);
    a.rows_valid = MP; a.cols_valid = e->LP;
    HIP_TRY(e, (gemm_forward<XF_PRELU, EPI_BIAS_TANH>(a, pre_buf(e, e->H), e->WP, e->Woc, e->WP, MP, e->LP, e->WP, st,
                                                      Prof{e, PC_FWD_OUT, 2.0 * 3 * B * (double)e->L * e->W}, cfg)));
  }
  LossArgs la{};
  la.Y = e->Y; la.x0 = x0; la.B = B; la.L = e->L; la.LP = e->LP; la.part = e->loss_part;
  SDRM_LAUNCH(e, k_loss_partials, dim3(LOSS_BLOCKS), dim3(1024), 0, st, la);
  HIP_TRY(e, hipGetLastError());
  if (!e->fold_sums) {
    SDRM_LAUNCH(e, k_loss_sums, dim3(1), dim3(256), 0, st, (const double*)e->loss_part, LOSS_BLOCKS,
                       (double)B * (double)e->L, sums ? sums : e->sums);
    HIP_TRY(e, hipGetLastError());
  }
  e->cur_B = B; e->cur_MP = MP; e->cur_x0 = x0; e->fwd_done = true;
  return SDRM_OK;
}

// Backward in two calls, so that a data-parallel caller can all-reduce one gradient bucket while the other is
// still being computed.  `begin` runs the chain every other kernel waits for (loss seeds, the dgrads down to
// layer 0), then the layer-0 weight gradient, its slab reduction and the small, latency-bound embedding
// backward: when it returns, the FIRST bucket (flat[0, off_a0): emb_layer.*, dnn.0.weight/bias) is final in
// stream order.  `finish` runs the weight gradients of the upper layers - two thirds of the wgrad flops,
// nothing but Adam depends on them - and reduces the second bucket (slopes, hidden and output layer).
// (Forking the upper wgrads to a second stream so that they also overlap the embedding backward on ONE GPU was
// measured: the two cross-stream event waits cost more than the ~45 us they hide, 642 vs 627 us per step.)
namespace {

// loss seeds, the dgrad chain down to layer 0, and the layer-0 weight gradient (whose time-embedding columns deliver M, csrc/tail.h)
int backward_chain(sdrm_engine* e, const double* sums, float* loss, hipStream_t st, bool with_wgrad0) {
  const int B = e->cur_B, MP = e->cur_MP, H = e->H;
  e->bwd_strips = false;
  SeedArgs sa{};
  sa.sums = e->fold_sums ? nullptr : (sums ? sums : e->sums); sa.Y = e->Y; sa.x0 = e->cur_x0; sa.dY = e->dY; sa.loss = loss;
  sa.B = B; sa.L = e->L; sa.LP = e->LP; sa.MP = MP; sa.grouped = (e->cur_sk || e->cur_g16) ? 2 : (e->cur_grouped ? 1 : 0);
  sa.part = e->loss_part;
  sa.nblk = e->cur_grouped ? (B + RC_USERS - 1) / RC_USERS : (e->cur_g16 ? e->cur_parts * ((B + R48_USERS - 1) / R48_USERS) : LOSS_BLOCKS);
  sa.count = (double)B * (double)e->L;
  // the row-owned chain (dgrad_rows.h) computes the seeds itself; every other path launches k_loss_seed
  const bool chain48 = e->cur_g16 && e->WhfT && e->tune.dgrad_rows > 0 && e->LP == e->WP && H + 1 <= DR_MAX_LAYERS;
  const bool chain = chain48 || (use_dgrad_rows(e, MP) && e->tune.dgrad_rows == 1 && H + 1 <= DR_MAX_LAYERS);
  if (!chain && !e->cur_sk) {
    const int nslots = e->cur_grouped ? RC_USERS * ((B + RC_USERS - 1) / RC_USERS) : (e->cur_g16 ? R48_USERS * ((B + R48_USERS - 1) / R48_USERS) : B);
    const unsigned need = (unsigned)(((size_t)(nslots + (MP - 3 * nslots)) * (e->LP / 4) + 255) / 256);
    dim3 grid(std::min(need, 2048u));   // grid-stride beyond: see k_loss_seed
    SDRM_LAUNCH(e, k_loss_seed, grid, dim3(256), 0, st, sa);
    HIP_TRY(e, hipGetLastError());
  }
  int S0, SH, SO, kc0, kcH, kcO;
  pick_splits(e->tune, MP, wgrad_tiles(e->tune, e->WP, e->K0) + H * wgrad_tiles(e->tune, e->WP, e->WP) + wgrad_tiles(e->tune, e->LP, e->WP),
              S0, kc0);
  SH = SO = S0; kcH = kcO = kc0;
  // every dgrad writes [MP,WP]: one tile shape for all of them, so the slope partial counts agree
  const int cfg_d = choose_cfg(e->tune, MP, e->tune.nt32_max_rows_train);
  const int cfg_w = pick_cfg(e->tune);   // tile of every split-K launch of this backward (backward_wgrads reuses it)
  e->bwd_cfg_w = cfg_w;
  const int dgrad_blocks = ((MP + kCfgBM[cfg_d] - 1) / kCfgBM[cfg_d]) * ((e->WP + kCfgBN[cfg_d] - 1) / kCfgBN[cfg_d]);
  const double flO = 2.0 * 3 * B * (double)e->L * e->W, flH = 2.0 * 3 * B * (double)e->W * e->W;
  const double fl0 = 2.0 * 3 * B * (double)e->W * (e->L + e->T);
  e->bwd_hidden_apps = H;
  if (e->cur_sk) {
    // narrow net (csrc/skinny_step.h): the sums' fold, the loss value, the seeds, the whole dgrad chain AND every weight / bias /
    // slope gradient of a work-group's users in one launch; one slab set per work-group (the hidden layer's applications already
    // summed), at most S_MAX of them - a work-group then walks several groups of users
    SkStepArgs ka = sk_step_args(e, B);
    ka.NP = e->cur_sk_np;
    ka.x0 = e->cur_x0; ka.sums = sa.sums; ka.count = sa.count; ka.loss = loss;
    const int S = std::min(ka.G, S_MAX);
    int rc = launch_sk_step(e, ka, 1, S, st);
    if (rc) return rc;
    e->bwd_S0 = e->bwd_SH = e->bwd_SO = S; e->bwd_hidden_apps = 1; e->bwd_dgrad_blocks = S;
    e->bwd_kc0 = e->bwd_kcH = e->bwd_kcO = 0;
    return SDRM_OK;
  }
  if (chain48) {
    // one work-group per 48 stacked rows runs the whole chain (csrc/rows48.h): one slope partial per work-group and layer
    DgradChain48Args c8{};
    DgradChainArgs& ca = c8.c;
    ca.seed = sa; ca.nlayers = H + 1;
    ca.layer[0] = dgrad_rows_args(e, e->dY, e->WofT, pre_buf(e, H), slope_ptr(e, H), dpre_buf(e, H), e->alpha_part + (size_t)H * e->alpha_part_stride);
    for (int k = H; k >= 1; --k)
      ca.layer[H + 1 - k] = dgrad_rows_args(e, dpre_buf(e, k), e->WhfT, pre_buf(e, k - 1), slope_ptr(e, k - 1), dpre_buf(e, k - 1),
                                            e->alpha_part + (size_t)(k - 1) * e->alpha_part_stride);
    const int Gn = e->cur_rows / R48_ROWS;
    c8.pad_rows = MP - e->cur_rows;
    int rc = launch_rows48_chain(e, c8, Gn, e->cur_parts, flO + H * flH, st);
    if (rc) return rc;
    if (with_wgrad0)
      HIP_TRY(e, (gemm_wgrad<XF_NONE>(dpre_buf(e, 0), e->WP, e->WP, e->U, e->K0, e->K0, nullptr, MP, S0, kc0, e->slab0, e->db0s, st,
                                      Prof{e, PC_WGRAD_L0, fl0}, cfg_w)));
    e->bwd_kc0 = kc0;
    e->bwd_S0 = S0; e->bwd_SH = SH; e->bwd_SO = SO; e->bwd_dgrad_blocks = Gn * e->cur_parts;
    e->bwd_kcH = kcH; e->bwd_kcO = kcO;
    return SDRM_OK;
  }
  if (use_dgrad_rows(e, MP)) {
    // one work-group per 96 stacked rows (and layer): one slope partial per work-group
    int rc = SDRM_OK;
    if (chain) {
      DgradChainArgs ca{};
      ca.seed = sa; ca.nlayers = H + 1;
      ca.layer[0] = dgrad_rows_args(e, e->dY, e->WofT, pre_buf(e, H), slope_ptr(e, H), dpre_buf(e, H), e->alpha_part + (size_t)H * e->alpha_part_stride);
      for (int k = H; k >= 1; --k)
        ca.layer[H + 1 - k] = dgrad_rows_args(e, dpre_buf(e, k), e->WhfT, pre_buf(e, k - 1), slope_ptr(e, k - 1), dpre_buf(e, k - 1),
                                              e->alpha_part + (size_t)(k - 1) * e->alpha_part_stride);
      rc = launch_dgrad_chain(e, ca, MP, flO + H * flH, st);
    } else {
      rc = launch_dgrad_rows(e, e->dY, e->WofT, pre_buf(e, H), slope_ptr(e, H), dpre_buf(e, H),
                             e->alpha_part + (size_t)H * e->alpha_part_stride, MP, flO, st);
      for (int k = H; k >= 1 && !rc; --k)
        rc = launch_dgrad_rows(e, dpre_buf(e, k), e->WhfT, pre_buf(e, k - 1), slope_ptr(e, k - 1), dpre_buf(e, k - 1),
                               e->alpha_part + (size_t)(k - 1) * e->alpha_part_stride, MP, flH, st);
    }
    if (rc) return rc;
    if (with_wgrad0)
      HIP_TRY(e, (gemm_wgrad<XF_NONE>(dpre_buf(e, 0), e->WP, e->WP, e->U, e->K0, e->K0, nullptr, MP, S0, kc0, e->slab0, e->db0s, st,
                                      Prof{e, PC_WGRAD_L0, fl0}, cfg_w)));
    e->bwd_kc0 = kc0;
    e->bwd_S0 = S0; e->bwd_SH = SH; e->bwd_SO = SO; e->bwd_dgrad_blocks = MP / RC_ROWS;
    e->bwd_kcH = kcH; e->bwd_kcO = kcO;
    return SDRM_OK;
  }
  // dpre[k] = gradient w.r.t. pre-activation k (kept for the weight gradients that run later)
  HIP_TRY(e, gemm_dgrad(e, e->dY, e->LP, e->WocT, e->LP, MP, e->LP, e->WP, dpre_buf(e, H), pre_buf(e, H), slope_ptr(e, H),
                        e->alpha_part + (size_t)H * e->alpha_part_stride, st, flO, cfg_d));
  for (int k = H; k >= 1; --k)
    HIP_TRY(e, gemm_dgrad(e, dpre_buf(e, k), e->WP, e->WhcT, e->WP, MP, e->WP, e->WP, dpre_buf(e, k - 1), pre_buf(e, k - 1),
                          slope_ptr(e, k - 1), e->alpha_part + (size_t)(k - 1) * e->alpha_part_stride, st, flH, cfg_d));
  // layer 0 (no latent dgrad: XT.grad is never read, Q7)
  if (with_wgrad0)
    HIP_TRY(e, (gemm_wgrad<XF_NONE>(dpre_buf(e, 0), e->WP, e->WP, e->U, e->K0, e->K0, nullptr, MP, S0, kc0, e->slab0, e->db0s, st,
                                    Prof{e, PC_WGRAD_L0, fl0}, cfg_w)));
  e->bwd_kc0 = kc0;
  e->bwd_S0 = S0; e->bwd_SH = SH; e->bwd_SO = SO; e->bwd_dgrad_blocks = dgrad_blocks;
  e->bwd_kcH = kcH; e->bwd_kcO = kcO;
  return SDRM_OK;
}

// weight gradients of the output and hidden layers (two thirds of the wgrad flops, only Adam waits for them), and, for
// the one-call backward, of layer 0 as well: ONE batched launch (gemm_batch_kernel) - every input is ready once the
// dgrad chain is done.  More problems than a batch holds (H > 6) go in several batches; a forced tile shape
// (SDRM_TILE) falls back to one launch per layer.
int backward_wgrads(sdrm_engine* e, hipStream_t st, bool with_wgrad0) {
  if (e->cur_sk) return SDRM_OK;   // the narrow nets' backward launch has left every weight gradient in its slabs
  const int B = e->cur_B, MP = e->cur_MP, H = e->H, SH = e->bwd_SH, SO = e->bwd_SO;
  const double flO = 2.0 * 3 * B * (double)e->L * e->W, flH = 2.0 * 3 * B * (double)e->W * e->W;
  const double fl0 = 2.0 * 3 * B * (double)e->W * (e->L + e->T);
  if (e->bwd_cfg_w > 0) {   // a forced tile other than the default: the batched kernel is built for the default only
    const int cfg_w = e->bwd_cfg_w;
    if (with_wgrad0)
      HIP_TRY(e, (gemm_wgrad<XF_NONE>(dpre_buf(e, 0), e->WP, e->WP, e->U, e->K0, e->K0, nullptr, MP, e->bwd_S0, e->bwd_kc0, e->slab0,
                                      e->db0s, st, Prof{e, PC_WGRAD_L0, fl0}, cfg_w)));
    HIP_TRY(e, (gemm_wgrad<XF_PRELU>(e->dY, e->LP, e->LP, pre_buf(e, H), e->WP, e->WP, slope_ptr(e, H), MP, SO, e->bwd_kcO,
                                     e->slabO, e->dbOs, st, Prof{e, PC_WGRAD, flO}, cfg_w)));
    for (int k = H; k >= 1; --k)
      HIP_TRY(e, (gemm_wgrad<XF_PRELU>(dpre_buf(e, k), e->WP, e->WP, pre_buf(e, k - 1), e->WP, e->WP, slope_ptr(e, k - 1), MP, SH,
                                       e->bwd_kcH, e->slabH + (size_t)(k - 1) * SH * e->WP * e->WP,
                                       e->dbHs + (size_t)(k - 1) * SH * e->WP, st, Prof{e, PC_WGRAD, flH}, cfg_w)));
    return SDRM_OK;
  }
  if (with_wgrad0 && use_strips(e)) return launch_wgrad_strips(e, MP, fl0 + flO + H * flH, st);
  std::vector<WgradSpec> w;
  std::vector<double> fl;
  if (with_wgrad0) {
    w.push_back(WgradSpec{dpre_buf(e, 0), e->WP, e->WP, e->U, e->K0, e->K0, e->one_dev, e->bwd_S0, e->bwd_kc0, e->slab0, e->db0s});
    fl.push_back(fl0);
  }
  // operand of the upper layers' weight gradients: the stored pre-activation (PReLU on load), or the activation itself
  const bool plain = e->cur_act;
  auto opnd = [&](int k) { return plain ? e->act + (size_t)k * e->MPmax * e->WP : pre_buf(e, k); };
  w.push_back(WgradSpec{e->dY, e->LP, e->LP, opnd(H), e->WP, e->WP, slope_ptr(e, H), SO, e->bwd_kcO, e->slabO, e->dbOs});
  fl.push_back(flO);
  for (int k = H; k >= 1; --k) {
    w.push_back(WgradSpec{dpre_buf(e, k), e->WP, e->WP, opnd(k - 1), e->WP, e->WP, slope_ptr(e, k - 1), SH, e->bwd_kcH,
                          e->slabH + (size_t)(k - 1) * SH * e->WP * e->WP, e->dbHs + (size_t)(k - 1) * SH * e->WP});
    fl.push_back(flH);
  }
  for (size_t lo = 0; lo < w.size(); lo += GEMM_BATCH_MAX) {
    const int n = (int)std::min<size_t>(GEMM_BATCH_MAX, w.size() - lo);
    double f = 0.0;
    for (int k = 0; k < n; ++k) f += fl[lo + k];
    HIP_TRY(e, launch_wgrad_batch(e, w.data() + lo, n, MP, st, Prof{e, PC_WGRAD, f}, plain));
  }
  return SDRM_OK;
}

enum { BUCKET_FIRST = 1, BUCKET_SECOND = 2, BUCKET_BOTH = 3 };

// The tail of a backward (csrc/tail.h): slabs -> flat gradient (written where the caller wants it, e.g. a DDP bucket - no copy
// afterwards), the embedding path's gradients from the time-embedding columns of the layer-0 slabs, and - `update` (the
// single-GPU step) - Adam straight from the sums with the re-pack of the compute copies: ONE launch for everything that needs
// only the slabs, a second for the embedding path (FIRST bucket: W0e and emb_layer.*, out of what the first launch left).
int backward_tail(sdrm_engine* e, float* gout, int which, bool update, float lr, hipStream_t st) {
  const int L = e->L, W = e->W, T = e->T, H = e->H;
  const int S0 = e->bwd_S0, SH = e->bwd_SH, SO = e->bwd_SO;
  const int bias_col = e->bwd_strips ? e->ones_col : -1;   // strip-owned weight gradients: the bias gradient is slab column ones_col
  TailArgs a{};
  int n = 0, blocks = 0;
  auto add = [&](int kind, int64_t off, int rows, int cols, int flat_ld, const float* src, int src_ld, size_t slab_stride, int nslabs,
                 int nblocks) -> TailJob& {
    TailJob& j = a.j[n];
    j.kind = kind; j.flat_off = off; j.rows = rows; j.cols = cols; j.flat_ld = flat_ld;
    j.src = src; j.src_ld = src_ld; j.slab_stride = slab_stride; j.nslabs = nslabs; j.inner = 1; j.nblocks = nblocks; j.lanes = 4;
    j.red = nullptr; j.red_ld = 0;
    j.dst = j.dstT = j.dstF = j.dstFT = nullptr; j.dst_ld = j.dstT_ld = 0; j.fnct = e->WP / 16; j.fklast = j.fklastT = -1;
    a.start[n++] = blocks;
    blocks += nblocks;
    return j;
  };
  // a weight's sub-tiles (csrc/tail.h): 16 x 16 with four lanes per column quad (each a quarter of the slabs) when there are more
  // than 32 slabs, 16 x 32 with two up to 32, 32 x 32 with one up to eight
  auto mat = [&](int64_t off, int rows, int cols, int flat_ld, const float* src, int src_ld, size_t slab_stride, int nslabs) -> TailJob& {
    const int lanes = nslabs > 32 ? 4 : (nslabs > 8 ? 2 : 1), tsr = lanes == 1 ? 32 : 16, tsc = lanes == 4 ? 16 : 32;
    TailJob& j = add(TJ_MAT, off, rows, cols, flat_ld, src, src_ld, slab_stride, nslabs, ((rows + tsr - 1) / tsr) * ((cols + tsc - 1) / tsc));
    j.lanes = lanes;
    return j;
  };
  auto vec = [&](int64_t off, int len, const float* col_src, int col_ld, size_t col_stride, const float* sum_src, size_t sum_stride, int nslabs,
                 float* dst) -> TailJob& {
    TailJob& j = bias_col >= 0 ? add(TJ_VEC, off, len, 1, 1, col_src + bias_col, col_ld, col_stride, nslabs, (len + 63) / 64)
                               : add(TJ_VEC, off, len, 1, 1, sum_src, 1, sum_stride, nslabs, (len + 63) / 64);   // 64 entries per work-group
    j.dst = dst;
    return j;
  };
  auto snap = [&](int64_t off, int rows, int cols, int flat_ld, float* dst, int dst_ld) {
    TailJob& j = add(TJ_SNAP, off, rows, cols, flat_ld, nullptr, 0, 0, 0, rows * ((cols + TAIL_THREADS * 8 - 1) / (TAIL_THREADS * 8)));   // pieces of 2048 elements of a row
    j.red = dst; j.red_ld = dst_ld;
  };
  if (which & BUCKET_FIRST) {
    {
      TailJob& j = mat(e->off_w0, W, L, L + T, e->slab0, e->K0, (size_t)e->WP * e->K0, S0);
      j.dst = e->W0c; j.dst_ld = e->K0; j.dstF = e->W0f; j.fklast = rc_light_klast(L, e->LP);
    }
    vec(e->off_b0, W, e->slab0, e->K0, (size_t)e->WP * e->K0, e->db0s, (size_t)e->WP, S0, e->b0c);
    // for the second launch: M = the time-embedding columns of the layer-0 slabs, summed; emb_layer.* and W0e as they are now
    {
      TailJob& j = mat(0, W, T, 0, e->slab0 + e->LP, e->K0, (size_t)e->WP * e->K0, S0);
      j.red = e->Mred; j.red_ld = e->TP;
    }
    snap(e->off_we, 1, T * T + T, 0, e->snap, 0);
    snap(e->off_w0 + L, W, T, L + T, e->snap + (size_t)T * T + T, T);
  }
  if (which & BUCKET_SECOND) {
    // PReLU slopes: per-work-group partials of the dgrad epilogues, [application][alpha_part_stride]
    {
      TailJob& j = add(TJ_SCALAR, e->off_a0, 1, 1, 1, e->alpha_part, 0, (size_t)e->alpha_part_stride, 1, 1);
      j.inner = e->bwd_dgrad_blocks;
    }
    if (H >= 1) {
      const int HS = e->bwd_hidden_apps * SH;   // slab sets of the shared hidden layer: one per application and K-slice, or per K-slice
      TailJob& j = mat(e->off_wh, W, W, W, e->slabH, e->WP, (size_t)e->WP * e->WP, HS);
      j.dst = e->Whc; j.dst_ld = e->WP; j.dstT = e->WhcT; j.dstT_ld = e->WP; j.dstF = e->Whf; j.dstFT = e->WhfT;
      j.fklast = j.fklastT = rc_light_klast(W, e->WP);
      vec(e->off_bh, W, e->slabH, e->WP, (size_t)e->WP * e->WP, e->dbHs, (size_t)e->WP, HS, e->bhc);
      TailJob& s = add(TJ_SCALAR, e->off_ah, 1, 1, 1, e->alpha_part + e->alpha_part_stride, 0, (size_t)e->alpha_part_stride, H, 1);
      s.inner = e->bwd_dgrad_blocks;
    }
    {
      TailJob& j = mat(e->off_wo, L, W, W, e->slabO, e->WP, (size_t)e->LP * e->WP, SO);
      j.dst = e->Woc; j.dst_ld = e->WP; j.dstT = e->WocT; j.dstT_ld = e->LP; j.dstF = e->Wof; j.dstFT = e->WofT;
      j.fklast = rc_light_klast(W, e->WP); j.fklastT = rc_light_klast(L, e->LP);
    }
    vec(e->off_bo, L, e->slabO, e->WP, (size_t)e->LP * e->WP, e->dbOs, (size_t)e->LP, SO, e->boc);
  }
  a.start[n] = blocks;
  a.n = n;
  a.p = e->p; a.m = e->m; a.v = e->v; a.g = gout;
  a.L = L; a.W = W; a.T = T; a.LP = e->LP; a.TP = e->TP;
  a.off_we = e->off_we; a.off_be = e->off_be; a.off_w0 = e->off_w0; a.off_b0 = e->off_b0;
  a.Mred = e->Mred; a.snap = e->snap; a.WeP = e->WeP; a.W0eP = e->W0eP; a.TPe = e->TPe;
  a.b1 = 0.9f; a.b2 = 0.999f; a.eps = 1e-8f; a.wd = 1e-4f; a.update = update ? 1 : 0;
  if (update) adam_scalars(e, lr, a.step_size, a.bc2_sqrt);
  int per_lane = 0;   // most slabs any lane of a weight job sums
  for (int q = 0; q < n; ++q)
    if (a.j[q].kind == TJ_MAT) per_lane = std::max(per_lane, (a.j[q].nslabs + a.j[q].lanes - 1) / a.j[q].lanes);
  if (per_lane <= 8) SDRM_LAUNCH(e, k_tail<8>, dim3((unsigned)blocks), dim3(TAIL_THREADS), 0, st, a);
  else SDRM_LAUNCH(e, k_tail<16>, dim3((unsigned)blocks), dim3(TAIL_THREADS), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  if (update) e->tables_fresh = false;
  if (which & BUCKET_FIRST) {
    const int eb = tail_emb_blocks_a(W, T) + tail_emb_blocks_b(T) + tail_emb_blocks_c(T);
    SDRM_LAUNCH(e, k_tail_emb, dim3((unsigned)eb), dim3(256), tail_emb_lds_floats(T, e->TP) * sizeof(float), st, a);
    HIP_TRY(e, hipGetLastError());
  }
  return SDRM_OK;
}

}  // namespace

int sdrm_train_backward_begin(sdrm_engine* e, const double* sums, float* grad, float* loss, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  if (!e->fwd_done) return fail(e, SDRM_ERR_STATE, "sdrm_train_backward: no forward to back-propagate");
  hipStream_t st = (hipStream_t)stream;
  e->bwd_begun = false;
  float* gout = grad ? grad : e->g;
  e->grad_src = gout;
  int rc = backward_chain(e, sums, loss, st, true);
  if (!rc) rc = backward_tail(e, gout, BUCKET_FIRST, false, 0.f, st);
  if (rc) return rc;
  e->bwd_begun = true;
  return SDRM_OK;
}

int sdrm_train_backward_finish(sdrm_engine* e, float* grad, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  if (!e->fwd_done || !e->bwd_begun) return fail(e, SDRM_ERR_STATE, "sdrm_train_backward_finish: sdrm_train_backward_begin not run");
  hipStream_t st = (hipStream_t)stream;
  float* gout = grad ? grad : e->g;
  if (gout != e->grad_src) return fail(e, SDRM_ERR_ARG, "sdrm_train_backward_finish: different gradient buffer than begin");
  int rc = backward_wgrads(e, st, false);
  if (!rc) rc = hold_point(e, st);
  if (!rc) rc = backward_tail(e, gout, BUCKET_SECOND, false, 0.f, st);
  e->bwd_begun = false;
  return rc;
}

// the whole backward in one call: same kernels, one tail for both buckets; `fused_lr` (the single-GPU step): Adam applied by the
// tail itself, straight from the slab sums (the flat gradient is still written: sdrm_get_grads)
// (Round 5 measured the single-GPU tail as two concurrent halves - the hidden / output layers' jobs on an auxiliary stream beside layer 0's
// jobs + k_tail_emb, optionally + the next step's k_emb_tables: the fork and the join cost more than the overlap gains on this runtime,
// train step 450.7 -> 467.3 / 457.4 us at B = 8192, 72.8 -> 98.9 at B = 160, ADM 47.2 -> 66.4: profiles/r05_tail_split.txt.)
static int train_backward_impl(sdrm_engine* e, const double* sums, float* grad, float* loss, hipStream_t st, const float* fused_lr) {
  e->bwd_begun = false;
  float* gout = grad ? grad : e->g;
  e->grad_src = gout;
  int rc = backward_chain(e, sums, loss, st, false);
  if (!rc) rc = backward_wgrads(e, st, true);
  if (!rc) rc = hold_point(e, st);
  if (!rc) rc = backward_tail(e, gout, BUCKET_BOTH, fused_lr != nullptr, fused_lr ? *fused_lr : 0.f, st);
  return rc;
}

int sdrm_train_backward(sdrm_engine* e, const double* sums, float* grad, float* loss, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  if (!e->fwd_done) return fail(e, SDRM_ERR_STATE, "sdrm_train_backward: no forward to back-propagate");
  return train_backward_impl(e, sums, grad, loss, (hipStream_t)stream, nullptr);
}

int sdrm_grad_buckets(const sdrm_engine* e, int64_t* first_off, int64_t* first_len, int64_t* second_off, int64_t* second_len) {
  if (!e || !first_off || !first_len || !second_off || !second_len) return SDRM_ERR_ARG;
  *first_off = 0; *first_len = e->off_a0;
  *second_off = e->off_a0; *second_len = e->P - e->off_a0;
  return SDRM_OK;
}

int sdrm_adam_step(sdrm_engine* e, const float* grad, float lr, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  if (int jr = join_chains(e, (hipStream_t)stream)) return jr;
  e->adam_t += 1;
  return launch_adam(e, grad, lr, 1, (hipStream_t)stream);
}

int sdrm_train_step(sdrm_engine* e, const float* x0, int B, float lr, int mode, const sdrm_train_randoms* rnd,
                    uint64_t seed, uint64_t step, float nd, float* loss, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  e->fold_sums = true;    // one process, one GPU: nobody needs the five sums between the forward and the backward ...
  int rc = sdrm_train_forward(e, x0, B, 0, mode, rnd, seed, step, nd, nullptr, stream);
  e->fold_sums = false;
  if (rc) return rc;
  e->fold_sums = true;
  e->adam_t += 1;         // ... nor the flat gradient between the backward and Adam: the tail applies it (csrc/tail.h)
  rc = train_backward_impl(e, nullptr, nullptr, loss, (hipStream_t)stream, &lr);
  e->fold_sums = false;
  if (rc) e->adam_t -= 1;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// The exchange step of the user-sharded train step, RCCL called from inside the library.
#define NCCL_TRY(e, api, call)                                                                  \
  do {                                                                                          \
    ncclResult_t _r = (call);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      (e)->err = std::string(#call) + ": " + (api)->GetErrorString(_r);                         \
      return SDRM_ERR_RCCL;                                                                     \
    }                                                                                           \
  } while (0)

int sdrm_comm_available(void) { return rccl_api(nullptr) != nullptr ? 1 : 0; }

int sdrm_comm_unique_id(void* id_host) {
  if (!id_host) return SDRM_ERR_ARG;
  const RcclApi* api = rccl_api(nullptr);
  if (!api) return SDRM_ERR_RCCL;
  static_assert(sizeof(ncclUniqueId) == SDRM_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return SDRM_ERR_RCCL;
  std::memcpy(id_host, &id, sizeof(id));
  return SDRM_OK;
}

namespace {
int exchange_streams(sdrm_engine* e, void* aux_stream) {
  Exchange& x = e->xch;
  if (aux_stream) { x.aux = (hipStream_t)aux_stream; x.aux_owned = false; }
  else { HIP_TRY(e, hipStreamCreateWithFlags(&x.aux, hipStreamNonBlocking)); x.aux_owned = true; }
  HIP_TRY(e, hipEventCreateWithFlags(&x.ev_bucket, hipEventDisableTiming));
  HIP_TRY(e, hipEventCreateWithFlags(&x.ev_done, hipEventDisableTiming));
  return SDRM_OK;
}
}  // namespace

int sdrm_comm_init_rank(sdrm_engine* e, int nranks, int rank, const void* id_host) {
  if (!e || !id_host || nranks < 1 || rank < 0 || rank >= nranks) return fail(e, SDRM_ERR_ARG, "sdrm_comm_init_rank: bad argument");
  if (e->xch.comm) return fail(e, SDRM_ERR_STATE, "sdrm_comm_init_rank: this engine already has a communicator");
  std::string why;
  const RcclApi* api = rccl_api(&why);
  if (!api) return fail(e, SDRM_ERR_RCCL, why);
  HIP_TRY(e, hipSetDevice(e->device));
  ncclUniqueId id;
  std::memcpy(&id, id_host, sizeof(id));
  ncclComm_t comm = nullptr;
  NCCL_TRY(e, api, api->CommInitRank(&comm, nranks, id, rank));
  e->xch.comm = comm; e->xch.comm_owned = true; e->xch.nranks = nranks; e->xch.rank = rank;
  return exchange_streams(e, nullptr);
}

int sdrm_allreduce_init(sdrm_engine* e, void* rccl_comm, void* aux_stream) {
  if (!e || !rccl_comm) return fail(e, SDRM_ERR_ARG, "sdrm_allreduce_init: null pointer");
  if (e->xch.comm) return fail(e, SDRM_ERR_STATE, "sdrm_allreduce_init: this engine already has a communicator");
  std::string why;
  const RcclApi* api = rccl_api(&why);
  if (!api) return fail(e, SDRM_ERR_RCCL, why);
  ncclComm_t comm = (ncclComm_t)rccl_comm;
  int n = 0, r = -1;
  NCCL_TRY(e, api, api->CommCount(comm, &n));
  NCCL_TRY(e, api, api->CommUserRank(comm, &r));
  e->xch.comm = comm; e->xch.comm_owned = false; e->xch.nranks = n; e->xch.rank = r;
  return exchange_streams(e, aux_stream);
}

int sdrm_comm_info(const sdrm_engine* e, int* nranks, int* rank) {
  if (!e) return SDRM_ERR_ARG;
  if (nranks) *nranks = e->xch.comm ? e->xch.nranks : 0;
  if (rank) *rank = e->xch.comm ? e->xch.rank : -1;
  return SDRM_OK;
}

void* sdrm_debug_comm_handle(const sdrm_engine* e) { return e ? (void*)e->xch.comm : nullptr; }

int sdrm_comm_destroy(sdrm_engine* e) {
  if (!e) return SDRM_ERR_ARG;
  Exchange& x = e->xch;
  if (!x.comm) return SDRM_OK;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  if (x.comm_owned)
    if (const RcclApi* api = rccl_api(nullptr)) (void)api->CommDestroy(x.comm);
  if (x.ev_bucket) (void)hipEventDestroy(x.ev_bucket);
  if (x.ev_done) (void)hipEventDestroy(x.ev_done);
  if (x.aux && x.aux_owned) (void)hipStreamDestroy(x.aux);
  x = Exchange{};
  return SDRM_OK;
}

// One user-sharded train step, exchange included (what a non-Python caller needs: sdrm_amd/parallel.py does the same
// with torch.distributed between the three phases).  Order on `stream` unless noted:
//   forwards + local loss sums -> all-reduce(5 doubles) -> loss seeds, dgrad chain, layer-0 weight gradient, first
//   bucket final -> [aux stream: all-reduce(first bucket)] beside the upper layers' weight gradients -> second bucket
//   final -> [aux: all-reduce(second bucket)] -> Adam.
// Every collective of the communicator is ordered against the next by stream dependencies, on every rank alike.
int sdrm_train_step_sharded(sdrm_engine* e, const float* x0, int B, int64_t row0, float lr, int mode,
                            const sdrm_train_randoms* rnd, uint64_t seed, uint64_t step, float nd, float* loss, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  Exchange& x = e->xch;
  if (!x.comm) return fail(e, SDRM_ERR_STATE, "sdrm_train_step_sharded: no communicator (sdrm_comm_init_rank / sdrm_allreduce_init)");
  const RcclApi* api = rccl_api(nullptr);
  if (!api) return fail(e, SDRM_ERR_RCCL, "librccl is gone");
  hipStream_t st = (hipStream_t)stream;
  int rc = sdrm_train_forward(e, x0, B, row0, mode, rnd, seed, step, nd, e->sums, stream);
  if (rc) return rc;
  NCCL_TRY(e, api, api->AllReduce(e->sums, e->sums, 5, ncclDouble, ncclSum, x.comm, st));
  // Default: ONE all-reduce of the whole gradient after the one-call backward (all weight gradients in one batched launch).
  // The two-bucket form below hides the first bucket's all-reduce behind the upper layers' weight gradients, but its event
  // hand-offs and split launches cost 36-45 us per step by themselves (one-rank communicator, where the collectives are
  // free: tools/exchange_probe.py, profiles/r02_exchange_probe.txt) - more than a 0.6 MB all-reduce between GPUs of one
  // node takes.  sdrm_debug_set_gradient_buckets(e, 2) / SDRM_AR_BUCKETS=2 selects it for nets whose gradient is large
  // enough for the overlap to pay.
  const int buckets = e->tune.ar_buckets == 2 ? 2 : 1;
  if (buckets == 1) {
    rc = sdrm_train_backward(e, e->sums, nullptr, loss, stream);
    if (rc) return rc;
    NCCL_TRY(e, api, api->AllReduce(e->g, e->g, (size_t)e->P, ncclFloat, ncclSum, x.comm, st));
    return sdrm_adam_step(e, nullptr, lr, stream);
  }
  rc = sdrm_train_backward_begin(e, e->sums, nullptr, loss, stream);
  if (rc) return rc;
  const int64_t n0 = e->off_a0, n1 = e->P - e->off_a0;
  HIP_TRY(e, hipEventRecord(x.ev_bucket, st));
  HIP_TRY(e, hipStreamWaitEvent(x.aux, x.ev_bucket, 0));
  NCCL_TRY(e, api, api->AllReduce(e->g, e->g, (size_t)n0, ncclFloat, ncclSum, x.comm, x.aux));
  rc = sdrm_train_backward_finish(e, nullptr, stream);
  if (rc) return rc;
  HIP_TRY(e, hipEventRecord(x.ev_bucket, st));
  HIP_TRY(e, hipStreamWaitEvent(x.aux, x.ev_bucket, 0));
  NCCL_TRY(e, api, api->AllReduce(e->g + n0, e->g + n0, (size_t)n1, ncclFloat, ncclSum, x.comm, x.aux));
  HIP_TRY(e, hipEventRecord(x.ev_done, x.aux));
  HIP_TRY(e, hipStreamWaitEvent(st, x.ev_done, 0));
  return sdrm_adam_step(e, nullptr, lr, stream);
}

int sdrm_get_train_outputs(const sdrm_engine* e, float* psq, void* stream) {
  if (!e || !psq) return SDRM_ERR_ARG;
  sdrm_engine* me = const_cast<sdrm_engine*>(e);
  if (!e->fwd_done) return fail(me, SDRM_ERR_STATE, "sdrm_get_train_outputs: no forward yet");
  SDRM_LAUNCH(e, k_unpad_psq, dim3(256), dim3(256), 0, (hipStream_t)stream, (const float*)e->Y, e->cur_B, e->L,
                     e->LP, (e->cur_sk || e->cur_g16) ? 2 : (e->cur_grouped ? 1 : 0), psq);
  HIP_TRY(me, hipGetLastError());
  return SDRM_OK;
}

// ---------------------------------------------------------------------------------------------
static int forward_rows(sdrm_engine* e, const float* x, const int64_t* t, int t_uniform, int n, int mode,
                        const uint8_t* keep, uint64_t seed, uint64_t step, int64_t row0, float* out, int ldout,
                        int cols_valid, hipStream_t st) {
  if (int jr = join_chains(e, st)) return jr;
  const int MP = round_up(n, BM);
  const int cfg = choose_cfg(e->tune, MP, e->tune.nt32_max_rows);
  PrepFwdArgs pa{};
  pa.x = x; pa.t = t; pa.t_uniform = t_uniform; pa.keep = keep; pa.U = e->U;
  pa.n = n; pa.L = e->L; pa.LP = e->LP; pa.K0 = e->K0; pa.T = e->T; pa.MP = MP;
  pa.mode = mode; pa.seed_lo = (uint32_t)seed; pa.seed_hi = (uint32_t)(seed >> 32); pa.step = (uint32_t)step;
  pa.row0 = row0;
  pa.bpr = (e->K0 / 2 + 255) / 256;
  dim3 grid((unsigned)((size_t)pa.bpr * MP));
  SDRM_LAUNCH(e, k_prep_forward, grid, dim3(256), 0, st, pa);
  HIP_TRY(e, hipGetLastError());
  int rc = ensure_tables(e, st);
  if (rc) return rc;
  {
    GemmArgs a{};
    a.C = pre_buf(e, 0); a.ldc = e->WP; a.bias = e->b0c;
    HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS>(a, e->U, e->K0, e->W0c, e->K0, MP, e->WP, e->K0, st,
                                                Prof{e, PC_FWD_L0, 2.0 * n * (double)e->W * (e->L + e->T)}, cfg)));
  }
  rc = hidden_forward(e, MP, n, st, cfg);
  if (rc) return rc;
  GemmArgs a{};
  a.C = out; a.ldc = ldout; a.bias = e->boc; a.slopeA = slope_ptr(e, e->H);
  a.rows_valid = n; a.cols_valid = cols_valid;
  HIP_TRY(e, (gemm_forward<XF_PRELU, EPI_BIAS_TANH_G>(a, pre_buf(e, e->H), e->WP, e->Woc, e->WP, MP, e->LP, e->WP, st,
                                                    Prof{e, PC_FWD_OUT, 2.0 * n * (double)e->L * e->W}, cfg)));
  return SDRM_OK;
}

int sdrm_forward(sdrm_engine* e, const float* x, const int64_t* t, int n, int mode, const uint8_t* keep, uint64_t seed,
                 uint64_t step, int64_t row0, float* out, void* stream) {
  if (!e || !x || !t || !out) return fail(e, SDRM_ERR_ARG, "sdrm_forward: null pointer");
  if (n < 1 || n > 3 * e->max_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_forward: n outside [1, 3*max_rows]");
  if (mode == SDRM_RNG_EXPLICIT && !keep) return fail(e, SDRM_ERR_ARG, "sdrm_forward: EXPLICIT mode needs keep");
  e->fwd_done = false;
  return forward_rows(e, x, t, 0, n, mode, keep, seed, step, row0, out, e->L, e->L, (hipStream_t)stream);
}

int sdrm_reverse_step(sdrm_engine* e, float* x, int n, int i, const float* z, const uint8_t* keep, void* stream) {
  if (!e || !x || !keep) return fail(e, SDRM_ERR_ARG, "sdrm_reverse_step: null pointer");
  if (n < 1 || n > 3 * e->max_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_reverse_step: n outside [1, 3*max_rows]");
  if (i < 1 || i > e->T) return fail(e, SDRM_ERR_ARG, "sdrm_reverse_step: step outside [1, T]");
  hipStream_t st = (hipStream_t)stream;
  e->fwd_done = false;
  int rc = forward_rows(e, x, nullptr, i, n, SDRM_RNG_EXPLICIT, keep, 0, 0, 0, e->Y, e->LP, e->LP, st);
  if (rc) return rc;
  float c1, sa, sb;
  reverse_coeffs(e, i, c1, sa, sb);
  SDRM_LAUNCH(e, k_reverse_apply, dim3(256), dim3(256), 0, st, x, (const float*)e->Y, e->LP, z, n, e->L, c1, sa, sb);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

int sdrm_perturb_input(sdrm_engine* e, const float* x, const int64_t* t, const float* noise, int n, float* out,
                       void* stream) {
  if (!e || !x || !t || !noise || !out || n < 1) return fail(e, SDRM_ERR_ARG, "sdrm_perturb_input: bad argument");
  const int nn = e->T + 1;
  SDRM_LAUNCH(e, k_perturb, dim3(256), dim3(256), 0, (hipStream_t)stream, x, t, noise,
                     (const float*)(e->sched + 3 * nn), (const float*)(e->sched + 4 * nn), n, e->L, e->T, out);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

// The narrow-net sampler: ONE persistent launch runs the whole reverse loop (rows are independent across all timesteps).  It is
// issued by sdrm_sample_begin, on that call's stream, so the call reads the parameters as they are at its begin - the same
// contract as the snapshot of the per-layer path; sdrm_sample_steps is then only book-keeping for the resumable API.
static int launch_skinny_sampler(sdrm_engine* e, hipStream_t st) {
  SampleState& s = e->smp;
  const int n = s.n, L = e->L;
  SkinnyArgs ka{};
  ka.W0c = e->W0c; ka.K0 = e->K0; ka.Whc = e->Whc; ka.Woc = e->Woc; ka.bh = e->bhc; ka.bo = e->boc;
  ka.B0tab = e->B0tab; ka.slope0 = slope_ptr(e, 0); ka.slopeh = e->H > 0 ? slope_ptr(e, 1) : slope_ptr(e, 0);
  ka.rev = e->rev_dev; ka.xT = s.xT; ka.Z = s.z; ka.keep = s.keep;
  ka.Tj = s.multires ? e->Tj_dev : nullptr; ka.rowid = s.multires ? e->rowid_dev : nullptr;
  ka.out = e->X; ka.n = n; ka.L = L; ka.W = e->W; ka.T = e->T; ka.H = e->H;
  ka.mode = s.mode; ka.seed_lo = (uint32_t)s.seed; ka.seed_hi = (uint32_t)(s.seed >> 32);
  ka.call_id = (uint32_t)s.call_id; ka.row0 = s.row0; ka.nd = s.nd;
  ka.LPs = e->LP; ka.WPs = e->WP;
  const int NL = (L + 15) / 16, NW = (e->W + 15) / 16;           // tiles with real columns (1..4 each)
  dim3 grid((n + 15) / 16), block(64 * (NL > NW ? NL : NW));   // 16 rows per work-group, one wave per column tile
#define SKINNY_LAUNCH(nl, nw) SDRM_LAUNCH(e, (k_skinny_sample<nl, nw>), grid, block, 0, st, ka)
#define SKINNY_ROW(nl)                                 \
switch (NW) {                                        \
  case 1: SKINNY_LAUNCH(nl, 1); break;               \
  case 2: SKINNY_LAUNCH(nl, 2); break;               \
  case 3: SKINNY_LAUNCH(nl, 3); break;               \
  default: SKINNY_LAUNCH(nl, 4); break;              \
}
  switch (NL) {
    case 1: SKINNY_ROW(1); break;
    case 2: SKINNY_ROW(2); break;
    case 3: SKINNY_ROW(3); break;
    default: SKINNY_ROW(4); break;
  }
#undef SKINNY_ROW
#undef SKINNY_LAUNCH
  HIP_TRY(e, hipGetLastError());
  s.skinny_launched = true;
  return SDRM_OK;
}

int sdrm_sample_begin(sdrm_engine* e, int n, float nd, int multires, int mode, const float* xT, const float* z,
                      const uint8_t* keep, const int64_t* Tj, uint64_t seed, uint64_t call_id, int64_t row0,
                      int64_t* Tj_out, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  e->smp.active = false;
  if (n < 1 || n > 3 * e->max_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_sample: n outside [1, 3*max_rows]");
  if (mode == SDRM_RNG_EXPLICIT && (!xT || !z || !keep || (multires && !Tj)))
    return fail(e, SDRM_ERR_ARG, "sdrm_sample: EXPLICIT mode needs xT, z, keep (and Tj for multi-resolution)");
  if (mode != SDRM_RNG_EXPLICIT && mode != SDRM_RNG_PHILOX) return fail(e, SDRM_ERR_ARG, "bad rng mode");
  if (multires && n > e->max_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_sample: multi-resolution n > max_rows");
  hipStream_t st = (hipStream_t)stream;
  if (int jr = join_chains(e, st)) return jr;
  const int T = e->T, L = e->L, MP = round_up(n, BM);
  e->fwd_done = false;
  e->n_chains = chains_for(e->tune, n, e->WP);
  e->chain_chunk = round_up((n + e->n_chains - 1) / e->n_chains, BM);
  e->n_chains = (n + e->chain_chunk - 1) / e->chain_chunk;
  e->n_aux = e->n_chains > 1 ? e->n_chains - 1 : 0;   // (the one chain of a small call: detached by the first train step queued beside it)
  e->detach_armed = false;
  int rc = ensure_tables(e, st);
  if (rc) return rc;
  int i_start = T;
  e->smp_nact.assign(T + 2, n);                       // n_act[i] = rows with Tj >= i (all rows when full resolution)
  if (multires) {
    // Start steps on the host: drawn with the same Philox call the device would make (PHILOX), or copied
    // back (EXPLICIT; one sync per sampling call).  Slots are then ordered by descending Tj.
    std::vector<int64_t> tj(n);
    if (mode == SDRM_RNG_PHILOX) {
      for (int r = 0; r < n; ++r) {
        const U4 w = philox4x32_10((uint32_t)(row0 + r), 0u, PURPOSE_SAMPLE_TJ, (uint32_t)call_id, (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
        tj[r] = 1 + (int64_t)bounded(w.x, (uint32_t)(T - 1 > 1 ? T - 1 : 1));   // np.random.randint(1, T), :42
      }
    } else {
      HIP_TRY(e, hipMemcpyAsync(tj.data(), Tj, (size_t)n * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(e, hipStreamSynchronize(st));
      for (int r = 0; r < n; ++r) tj[r] = tj[r] < 0 ? 0 : (tj[r] > T ? T : tj[r]);
    }
    std::vector<int> perm(n);
    for (int r = 0; r < n; ++r) perm[r] = r;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return tj[a] > tj[b]; });
    e->smp_tj_sorted.resize(n);
    for (int s = 0; s < n; ++s) e->smp_tj_sorted[s] = tj[perm[s]];
    e->smp_perm = perm;
    std::vector<int> hist(T + 2, 0);
    for (int r = 0; r < n; ++r) hist[(int)tj[r]]++;
    int acc = 0;
    for (int i = T; i >= 0; --i) { acc += hist[i]; e->smp_nact[i] = acc; }
    i_start = (int)e->smp_tj_sorted[0];
    HIP_TRY(e, hipMemcpyAsync(e->rowid_dev, e->smp_perm.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(e, hipMemcpyAsync(e->Tj_dev, e->smp_tj_sorted.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (Tj_out) {
      e->smp_tj_orig = tj;
      HIP_TRY(e, hipMemcpyAsync(Tj_out, e->smp_tj_orig.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
  }
  const bool skinny = skinny_net(e);
  if (skinny) {
    e->smp = SampleState{true, n, MP, multires, mode, i_start, nd, z, keep, seed, call_id, row0, xT, true, false, i_start};
    const int rc = launch_skinny_sampler(e, st);
    if (rc != SDRM_OK) e->smp.active = false;
    return rc;
  }
  {
    // the call's snapshot of the net (see sdrm_engine::smp_w): one launch for all of its pieces
    float* b = e->smp_w;
    CopySegs cs{};
    auto seg = [&](int i, int k, const float* src, size_t n_, size_t extra = 0) {
      cs.src[i] = src; cs.dst[i] = b + e->smp_off[k] + extra; cs.n[i] = src ? (unsigned)n_ : 0u;
    };
    seg(0, 0, e->W0c, (size_t)e->WP * e->K0);
    seg(1, 1, e->H >= 1 ? e->Whc : nullptr, (size_t)e->WP * e->WP);
    seg(2, 2, e->Woc, (size_t)e->LP * e->WP);
    seg(3, 3, e->H >= 1 ? e->bhc : nullptr, (size_t)e->WP);
    seg(4, 4, e->boc, (size_t)e->LP);
    seg(5, 5, e->B0tab, (size_t)(T + 1) * e->WP);
    seg(6, 6, slope_ptr(e, 0), 1);
    seg(7, 6, e->H >= 1 ? slope_ptr(e, 1) : nullptr, 1, 1);
    SDRM_LAUNCH(e, k_copy_segments, dim3(64, 8), dim3(256), 0, st, cs);
    HIP_TRY(e, hipGetLastError());
  }
  SampleInitArgs ia{};
  ia.xT = xT; ia.keep = keep; ia.Tj = multires ? e->Tj_dev : nullptr; ia.rowid = multires ? e->rowid_dev : nullptr;
  ia.X = e->X; ia.U = e->Us; ia.n = n; ia.L = L; ia.LP = e->LP; ia.K0 = e->LP; ia.MP = MP; ia.T = T;
  ia.mode = mode; ia.seed_lo = (uint32_t)seed; ia.seed_hi = (uint32_t)(seed >> 32);
  ia.call_id = (uint32_t)call_id; ia.row0 = row0;
  ia.bpr = (e->LP / 4 + 255) / 256;
  dim3 grid((unsigned)((size_t)ia.bpr * MP));
  SDRM_LAUNCH(e, k_sample_init, grid, dim3(256), 0, st, ia);
  HIP_TRY(e, hipGetLastError());
  e->smp = SampleState{true, n, MP, multires, mode, i_start, nd, z, keep, seed, call_id, row0, xT, false, false, i_start};
  if (e->xcntS) {   // the persistent sampler's counters start every call at zero (row tiles differ from call to call)
    HIP_TRY(e, hipMemsetAsync(e->xcntS, 0, (size_t)256 * 32 * sizeof(unsigned), st));
    e->xphaseS = 0;
  }
  return SDRM_OK;
}

namespace {
// Reverse steps in one launch (csrc/sample_persist.h): full resolution, PHILOX (the reverse update rides in the out layer's
// epilogue), a net whose layers share one tiling (L == W), one row chain, a forced tile only if it is the 32x32 one, the chip's
// block -> XCD mapping, and every work-group resident at once: 32x32 tiles, at most two per CU on the fullest XCD.
// by size: up to 11 row tiles of 32 (at most two row tiles = 22 work-groups per XCD: one per CU).  Measured (tools/sample_persist_probe.py,
// profiles/r05_sample_persist_probe.txt): n = 339: 12.1 us per step for a whole call in one launch against 16.5 for the three
// launches per step (15.7 driven one step per call); n = 679 - the 8-GPU shard: 22 row tiles put 33 work-groups on six XCDs' 32
// CUs, the doubled CU sets every phase - 16.0 against 16.4 (19.9 one step per call), n = 1024: 18.8 against 17.7: not taken there
constexpr int SMP_PERSIST_MAX_ROWS = 352;
bool sample_persist_fits(const sdrm_engine* e, const SampleState& s) {
  if (e->tune.smp_persist <= 0 || !e->xcd_ok || !e->xcntS || !e->xabort_host || xabort_read(e) != 0u) return false;
  if (s.multires || s.mode != SDRM_RNG_PHILOX || e->LP != e->WP || e->n_chains != 1) return false;
  if (e->tune.force_cfg >= 0 && e->tune.force_cfg != 4) return false;
  if (e->tune.fuse_rev == 0) return false;   // (a caller who asked for the stand-alone reverse update gets the per-layer path)
  const int tiles_m = s.MP / 32, tiles_n = e->WP / 32;
  const int per_xcd = ((tiles_m + 7) / 8) * tiles_n;
  if (per_xcd > 64 || tiles_m > 256) return false;
  return e->tune.smp_persist >= 2 || s.n <= SMP_PERSIST_MAX_ROWS;
}

int launch_sample_persist(sdrm_engine* e, SampleState& s, int count, hipStream_t st) {
  const NetView nv = snapshot_view(e);
  const int MP = s.MP, tiles_m = MP / 32, tiles_n = e->WP / 32;
  SamplePersistArgs P{};
  auto layer = [&](GemmArgs& a, const float* A, int lda, const float* Wc, int ldw, float* C, int ldc, int K) {
    a.A = A; a.lda = lda; a.limA = MP; a.B = Wc; a.ldb = ldw; a.limB = e->WP; a.K = K; a.kchunk = K; a.C = C; a.ldc = ldc;
    return gemm_set_grid(a, tiles_m, tiles_n, 1);
  };
  bool ok = layer(P.l0, e->Us, e->LP, nv.W0c, e->K0, smp_buf(e, 0), e->WP, e->LP);
  P.l0.slopeE = nv.slope0;
  ok = ok && layer(P.lh, smp_buf(e, 0), e->WP, nv.Whc, e->WP, smp_buf(e, 1), e->WP, e->WP);
  P.lh.bias = nv.bhc; P.lh.slopeE = nv.slopeh;
  ok = ok && layer(P.lo, smp_buf(e, e->H), e->WP, nv.Woc, e->WP, e->smp_Y, e->LP, e->WP);
  if (!ok) return fail(e, SDRM_ERR_SHAPE, "persistent sampler: grid beyond the tile arithmetic");
  GemmArgs& a = P.lo;
  a.bias = nv.boc; a.rows_valid = MP; a.cols_valid = e->LP;
  a.revX = e->X; a.revU = e->Us; a.rev_ldx = e->LP; a.rev_s0 = 0; a.rev_n = s.n; a.rev_L = e->L; a.rev_nd = s.nd;
  a.rev_seed_lo = (uint32_t)s.seed; a.rev_seed_hi = (uint32_t)(s.seed >> 32); a.rev_call_id = (uint32_t)s.call_id; a.rev_row0 = s.row0;
  P.pre_stride = (size_t)e->MPmax * e->WP;
  P.B0tab = nv.B0tab; P.ldtab = e->WP; P.rev = e->rev_dev;
  P.T = e->T; P.H = e->H; P.i_first = s.i_next; P.count = count;
  P.row_tiles = tiles_m; P.tiles_n = tiles_n;
  P.cnt = e->xcntS; P.base = e->xphaseS * (uint32_t)tiles_n; P.abort_ = e->xabort_dev;
  const int steps = std::min(count, s.i_next);
  e->xphaseS += (uint32_t)(steps * (e->H + 2));
  const int grid = 8 * tiles_n * ((tiles_m + 7) / 8);
  const double flops = 2.0 * s.n * ((double)e->W * e->L + (double)e->H * e->W * e->W + (double)e->L * e->W) * steps;
  const bool rec = e->prof_on && (int)e->prof_cls.size() < e->prof_cap && (e->prof_only < 0 || e->prof_only == PC_SMP_PERSIST);
  size_t slot = 0;
  if (rec) {
    slot = e->prof_cls.size();
    e->prof_cls.push_back(PC_SMP_PERSIST);
    e->prof_flops.push_back(flops);
    HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot], st));
  }
  SDRM_LAUNCH(e, (k_sample_persist<Cfg4>), dim3((unsigned)grid), dim3(NTHREADS), 0, st, P);
  HIP_TRY(e, hipGetLastError());
  if (rec) HIP_TRY(e, hipEventRecord(e->prof_ev[2 * slot + 1], st));
  s.i_next -= steps;
  return SDRM_OK;
}
}  // namespace

int sdrm_sample_steps(sdrm_engine* e, int count, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  if (!e->smp.active) return fail(e, SDRM_ERR_STATE, "sdrm_sample_steps: no sampling call in progress");
  hipStream_t st = (hipStream_t)stream;
  SampleState& s = e->smp;
  const int n = s.n, L = e->L;
  const size_t nL = (size_t)n * L;
  if (s.skinny) {
    // One persistent launch runs the whole reverse loop (rows are independent across all timesteps); the
    // step counter is then only book-keeping for the resumable API.
    if (!s.skinny_launched) return fail(e, SDRM_ERR_STATE, "sdrm_sample_steps: the sampler was not launched");
    s.i_next = s.i_next > count ? s.i_next - count : 0;
    return SDRM_OK;
  }
  // A train forward that was waiting for its backward is dropped (the documented rule of the two-call train API; the sampler has had
  // layer buffers of its own since round 5 - smp_pre, smp_Y - so whole train steps between sampling steps share nothing with the call).
  e->fwd_done = false;
  e->bwd_begun = false;
  // Train steps may run between sdrm_sample_steps calls (bench.py interleaves them): the sampler reads its own snapshot of
  // the net (sdrm_sample_begin), so they change nothing of this call.
  if (count > 0 && s.i_next >= 1 && sample_persist_fits(e, s)) return launch_sample_persist(e, s, count, st);
  const NetView nv = snapshot_view(e);
  const bool serial = e->prof_on;                    // (chains_for: an event profile wants launches that do not share the chip)
  if (serial) {
    e->detach_armed = false;
    if (int jr = join_chains(e, st)) return jr;
  } else if (e->detach_armed) {                         // the first steps behind a train step that was queued beside this small call
    e->n_aux = 1;
    HIP_TRY(e, hipStreamWaitEvent(e->aux[0], e->ev_fork, 0));
    e->chains_pending = true;
    e->detach_armed = false;
    if (e->hold_needed)                                 // (a row-owned step: not beside its MFMA kernels)
      if (int hr = hold_chains(e, st)) return hr;
  } else if (e->n_aux > 0 && e->chains_pending && e->train_since_sample) {
    if (e->hold_needed)
      if (int hr = hold_chains(e, st)) return hr;
  } else if (e->n_aux > 0 && !e->chains_pending) {   // fork: the chains on auxiliary streams start after everything queued on st so far
    HIP_TRY(e, hipEventRecord(e->ev_fork, st));
    for (int c = 0; c < e->n_aux; ++c) HIP_TRY(e, hipStreamWaitEvent(e->aux[c], e->ev_fork, 0));
    e->chains_pending = true;                        // joined lazily by the next entry point that needs the result
  }
  e->train_since_sample = false;
  for (int done = 0; done < count && s.i_next >= 1; ++done, --s.i_next) {
    const int i = s.i_next;
    const int na = e->smp_nact[i];                 // active prefix at this step
    for (int c = 0; c < e->n_chains; ++c) {
      const int s0 = c * e->chain_chunk, s1 = std::min(na, s0 + e->chain_chunk);
      if (s1 <= s0) break;
      hipStream_t sc = serial ? st : (e->n_aux == e->n_chains ? e->aux[c] : (c == 0 ? st : e->aux[c - 1]));
      const int rows = s1 - s0, MP = round_up(rows, BM);
      const int cfg = choose_cfg(e->tune, MP, e->tune.nt32_max_rows);
      {
        // the sampler keeps ACTIVATIONS in the layer buffers (EPI_BIAS_PRELU): no backward will ask for the pre-activations
        GemmArgs a{};
        a.C = smp_buf(e, 0) + (size_t)s0 * e->WP; a.ldc = e->WP; a.bias = nv.B0tab + (size_t)i * e->WP; a.slopeE = nv.slope0;
        HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_PRELU>(a, e->Us + (size_t)s0 * e->LP, e->LP, nv.W0c, e->K0, MP, e->WP, e->LP, sc,
                                                          Prof{e, PC_SMP_L0, 2.0 * rows * (double)e->W * e->L}, cfg)));
      }
      int rc = hidden_forward(e, MP, rows, sc, cfg, s0, PC_SMP_HIDDEN, true, &nv);
      if (rc) return rc;
      float c1, sqrt_alpha, sqrt_beta;
      reverse_coeffs(e, i, c1, sqrt_alpha, sqrt_beta);
      // (round 5: multi-resolution steps too - their active prefix is a launch like any other, the epilogue keys Philox by the slot's
      // original row; EXPLICIT randoms keep the stand-alone kernel)
      const bool fused = s.mode == SDRM_RNG_PHILOX && (e->tune.fuse_rev == 2 || (e->tune.fuse_rev == 1 && rows <= FUSE_REV_MAX_ROWS));
      {
        GemmArgs a{};
        a.C = e->smp_Y + (size_t)s0 * e->LP; a.ldc = e->LP; a.bias = nv.boc;
        a.rows_valid = MP; a.cols_valid = e->LP;
        const Prof pr{e, PC_SMP_OUT, 2.0 * rows * (double)e->L * e->W};
        if (fused) {
          // eps_hat never reaches memory: the epilogue applies denoise_add_noise to the sampler state and writes the
          // next step's dropped-out input (one launch less per reverse step)
          a.revX = e->X; a.revU = e->Us; a.rev_ldx = e->LP; a.rev_s0 = s0; a.rev_n = s1; a.rev_L = L; a.rev_step = i;
          a.rev_c1 = c1; a.rev_sqrt_alpha = sqrt_alpha; a.rev_sqrt_beta = sqrt_beta; a.rev_nd = s.nd;
          a.rev_seed_lo = (uint32_t)s.seed; a.rev_seed_hi = (uint32_t)(s.seed >> 32); a.rev_call_id = (uint32_t)s.call_id;
          a.rev_row0 = s.row0; a.rev_rowid = s.multires ? e->rowid_dev : nullptr;
          HIP_TRY(e, (gemm_forward<XF_NONE, EPI_TANH_REV>(a, smp_buf(e, e->H) + (size_t)s0 * e->WP, e->WP, nv.Woc, e->WP, MP,
                                                          e->LP, e->WP, sc, pr, cfg)));
          continue;
        }
        HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_TANH>(a, smp_buf(e, e->H) + (size_t)s0 * e->WP, e->WP, nv.Woc, e->WP, MP,
                                                         e->LP, e->WP, sc, pr, cfg)));
      }
      ReverseArgs ra{};
      ra.X = e->X; ra.Y = e->smp_Y; ra.U = e->Us;
      ra.Z = (s.mode == SDRM_RNG_EXPLICIT && i > 1) ? s.z + (size_t)i * nL : nullptr;
      ra.keep_next = (s.mode == SDRM_RNG_EXPLICIT && i > 1) ? s.keep + (size_t)(i - 1) * nL : nullptr;
      ra.Tj = s.multires ? e->Tj_dev : nullptr;
      ra.rowid = s.multires ? e->rowid_dev : nullptr;
      ra.s0 = s0; ra.n = s1; ra.L = L; ra.LP = e->LP; ra.K0 = e->LP; ra.step_i = i; ra.nd = s.nd;
      ra.c1 = c1; ra.sqrt_alpha = sqrt_alpha; ra.sqrt_beta = sqrt_beta;
      ra.mode = s.mode; ra.seed_lo = (uint32_t)s.seed; ra.seed_hi = (uint32_t)(s.seed >> 32);
      ra.call_id = (uint32_t)s.call_id; ra.row0 = s.row0;
      ra.bpr = ((L + 3) / 4 + 255) / 256;
      ra.rpb = std::max(1, 256 / ((L + 3) / 4));
      const unsigned rev_grid = ra.rpb > 1 ? (unsigned)((rows + ra.rpb - 1) / ra.rpb) : (unsigned)((size_t)ra.bpr * rows);
      SDRM_LAUNCH(e, k_reverse_update, dim3(rev_grid), dim3(256), 0, sc, ra);
      HIP_TRY(e, hipGetLastError());
    }
  }
  return SDRM_OK;
}

int sdrm_sample_remaining(const sdrm_engine* e) { return (e && e->smp.active) ? e->smp.i_next : 0; }

int sdrm_sample_end(sdrm_engine* e, float* out, void* stream) {
  if (!e || !out) return SDRM_ERR_ARG;
  if (!e->smp.active) return fail(e, SDRM_ERR_STATE, "sdrm_sample_end: no sampling call in progress");
  if (e->smp.i_next >= 1) return fail(e, SDRM_ERR_STATE, "sdrm_sample_end: reverse steps still pending");
  if (int jr = join_chains(e, (hipStream_t)stream)) return jr;
  if (xabort_read(e) != 0u) {   // a hand-shake of the persistent sampler (or of a split train step) timed out
    *(volatile unsigned*)e->xabort_host = 0u;
    e->tune.smp_persist = 0; e->tune.split = 0; e->xgeoF = e->xgeoC = 0;
    e->smp.active = false;
    return fail(e, SDRM_ERR_HIP, "a launch that synchronises work-groups through an XCD's L2 (csrc/sample_persist.h, csrc/rows48.h) timed out: "
                                 "the sampling call's result is invalid; those paths are switched off for this handle");
  }
  if (e->smp.skinny)   // the persistent kernel wrote dense [n,L] rows in original order
    HIP_TRY(e, hipMemcpyAsync(out, e->X, (size_t)e->smp.n * e->L * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  else
    SDRM_LAUNCH(e, k_unpad_rows, dim3((unsigned)std::min<size_t>(4096, ((size_t)e->smp.n * e->L + 255) / 256)), dim3(256), 0,
                (hipStream_t)stream, (const float*)e->X, e->LP, out, e->smp.n, e->L,
                (const int*)(e->smp.multires ? e->rowid_dev : nullptr));
  HIP_TRY(e, hipGetLastError());
  e->smp.active = false;
  return SDRM_OK;
}

int sdrm_sample(sdrm_engine* e, int n, float nd, int multires, int mode, const float* xT, const float* z,
                const uint8_t* keep, const int64_t* Tj, uint64_t seed, uint64_t call_id, int64_t row0, float* out,
                int64_t* Tj_out, void* stream) {
  if (!e || !out) return fail(e, SDRM_ERR_ARG, "sdrm_sample: null pointer");
  int rc = sdrm_sample_begin(e, n, nd, multires, mode, xT, z, keep, Tj, seed, call_id, row0, Tj_out, stream);
  if (rc) return rc;
  rc = sdrm_sample_steps(e, e->T + 1, stream);
  if (rc) return rc;
  return sdrm_sample_end(e, out, stream);
}

int sdrm_get_preacts(const sdrm_engine* e, int layer, float* out, void* stream) {
  if (!e || !out) return SDRM_ERR_ARG;
  sdrm_engine* me = const_cast<sdrm_engine*>(e);
  if (!e->fwd_done) return fail(me, SDRM_ERR_STATE, "sdrm_get_preacts: no train forward yet");
  if (layer < 0 || layer > e->H) return fail(me, SDRM_ERR_ARG, "sdrm_get_preacts: layer outside [0,H]");
  const float* actl = (e->cur_act && e->cur_skip_pre && e->act) ? e->act + (size_t)layer * e->MPmax * e->WP : nullptr;
  SDRM_LAUNCH(e, k_unpad_pre, dim3(256), dim3(256), 0, (hipStream_t)stream, (const float*)pre_buf(me, layer), actl,
                     (const float*)slope_ptr(me, layer), e->cur_B, e->W, e->WP, (e->cur_sk || e->cur_g16) ? 2 : (e->cur_grouped ? 1 : 0), out);
  HIP_TRY(me, hipGetLastError());
  return SDRM_OK;
}

// ---------------------------------------------------------------------------------------------
// Sparse batch feed (dataloaders.py:46-79, train_SDRM.py:323), csrc/feed.h.
int sdrm_csr_rows_to_dense(sdrm_engine* e, const int64_t* indptr, const int32_t* indices, const float* data, int64_t n_rows,
                           const int64_t* rows, int64_t row0, int b, int n_items, float* out, void* stream) {
  if (!e || !indptr || !indices || !out) return fail(e, SDRM_ERR_ARG, "sdrm_csr_rows_to_dense: null pointer");
  if (b < 1 || n_items < 1 || row0 < 0 || n_rows < 1)
    return fail(e, SDRM_ERR_SHAPE, "sdrm_csr_rows_to_dense: b < 1, n_items < 1, n_rows < 1 or row0 < 0");
  if (!rows && row0 + b > n_rows) return fail(e, SDRM_ERR_SHAPE, "sdrm_csr_rows_to_dense: rows row0 .. row0 + b - 1 end behind the matrix");
  FeedArgs a{};
  a.indptr = indptr; a.indices = indices; a.data = data; a.rows = rows; a.row0 = row0; a.n_rows = n_rows; a.b = b; a.n_items = n_items;
  a.out = out; a.flag = e->feed_flag;
  SDRM_LAUNCH(e, k_csr_rows_to_dense, dim3(b), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

// what the feed launches since the last call found (synchronises `stream`), and clears it
int sdrm_feed_status(sdrm_engine* e, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  unsigned flag = 0;
  HIP_TRY(e, hipMemcpyAsync(&flag, e->feed_flag, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
  if (!flag) return SDRM_OK;
  HIP_TRY(e, hipMemsetAsync(e->feed_flag, 0, sizeof(flag), (hipStream_t)stream));
  std::string msg = "sdrm_csr_rows_to_dense:";
  if (flag & FEED_BAD_ROW) msg += " a row id outside [0, n_rows) (its output row is zero);";
  if (flag & FEED_BAD_PTR) msg += " an indptr pair that is negative or not ordered (its output row is zero);";
  if (flag & FEED_BAD_COL) msg += " a column index outside [0, n_items) (that entry was skipped);";
  return fail(e, SDRM_ERR_ARG, msg);
}

// ---------------------------------------------------------------------------------------------
// Equal-sparsity binarisation of sampled data (main.py:177-180), csrc/select.h.
namespace {

// np.quantile(a, q) for float32 `a` and a Python-float q (numpy 2.x: q and the virtual index take a's dtype):
//   virtual = float32(n-1) * float32(q); previous = floor(virtual); next = previous + 1; gamma = virtual - previous
// (volatile: every operation rounds to float32, no contraction)
void quantile_ranks(int64_t n, double q, int64_t& r0, int64_t& r1, float& gamma) {
  volatile float q32 = (float)q;
  volatile float nm1 = (float)(n - 1);
  volatile float virt = nm1 * q32;
  if (virt >= nm1) { r0 = r1 = n - 1; gamma = 0.f; }
  else if (virt < 0.f) { r0 = r1 = 0; gamma = 0.f; }
  else {
    const float prev = std::floor(virt);
    volatile float g = virt - prev;
    gamma = g;
    r0 = (int64_t)prev; r1 = r0 + 1;
    if (r0 > n - 1) r0 = n - 1;
    if (r1 > n - 1) r1 = n - 1;
  }
}

// The three select sweeps, then the threshold and the binarise sweep.
int select_and_binarize(sdrm_engine* e, const float* x, int64_t n, float gamma, uint8_t* out, float* threshold, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256 + 1);
  for (int pass = 0; pass < SEL_PASSES; ++pass) {
    if (pass == 0) SDRM_LAUNCH(e, (k_select_hist<4>), dim3(blocks), dim3(256), 0, st, x, n, e->sel, pass);
    else SDRM_LAUNCH(e, (k_select_hist<1>), dim3(blocks), dim3(256), 0, st, x, n, e->sel, pass);
    HIP_TRY(e, hipGetLastError());
    SDRM_LAUNCH(e, k_select_pick, dim3(1), dim3(256), 0, st, e->sel, pass, gamma);
    HIP_TRY(e, hipGetLastError());
  }
  if (threshold) HIP_TRY(e, hipMemcpyAsync(threshold, &e->sel->threshold, 4, hipMemcpyDeviceToDevice, st));
  if (out) {
    SDRM_LAUNCH(e, k_binarize_ge, dim3(blocks), dim3(256), 0, st, x, n, (const float*)&e->sel->threshold, out);
    HIP_TRY(e, hipGetLastError());
  }
  return SDRM_OK;
}

int dec_grow(sdrm_engine* e, int slot, size_t n) {
  if (n <= e->dec_cap[slot]) return SDRM_OK;
  if (e->dec_buf[slot]) { HIP_TRY(e, hipDeviceSynchronize()); HIP_TRY(e, hipFree(e->dec_buf[slot])); e->dec_buf[slot] = nullptr; e->dec_cap[slot] = 0; }
  HIP_TRY(e, dalloc(&e->dec_buf[slot], n));
  e->dec_cap[slot] = n;
  return SDRM_OK;
}

// decoder(z) = Linear(hidden, items)(tanh(Linear(latent, hidden)(z)))   (train_SDRM.py:212-214, :252-254): one staging launch, two GEMMs
int decode_launches(sdrm_engine* e, const sdrm_vae_decoder* d, const float* z, int n, float* out, hipStream_t st) {
  const int Lp = round_up(d->latent, 32), Hp = round_up(d->hidden, 32), Ip = round_up(d->n_items, 32);
  const int MP = round_up(n, 128), Hr = round_up(Hp, 128), Ir = round_up(Ip, 128);
  enum { Z = 0, W1, B1, HID, W2, B2 };
  int rc;
  if ((rc = dec_grow(e, Z, (size_t)MP * Lp)) || (rc = dec_grow(e, W1, (size_t)Hr * Lp)) || (rc = dec_grow(e, B1, Hr)) ||
      (rc = dec_grow(e, HID, (size_t)MP * Hp)) || (rc = dec_grow(e, W2, (size_t)Ir * Hp)) || (rc = dec_grow(e, B2, Ir)))
    return rc;
  {
    PadSegs sg{};
    int64_t most = 0;
    int k = 0;
    auto pad = [&](const float* src, int rows, int cols, float* dst, int rowsP, int colsP) {
      sg.src[k] = src; sg.dst[k] = dst; sg.rows[k] = rows; sg.cols[k] = cols; sg.rowsP[k] = rowsP; sg.colsP[k] = colsP;
      most = std::max<int64_t>(most, (int64_t)rowsP * (colsP / 4));
      ++k;
    };
    pad(z, n, d->latent, e->dec_buf[Z], MP, Lp);
    pad(d->w1, d->hidden, d->latent, e->dec_buf[W1], Hr, Lp);
    pad(d->b1, 1, d->hidden, e->dec_buf[B1], 1, Hr);
    pad(d->w2, d->n_items, d->hidden, e->dec_buf[W2], Ir, Hp);
    pad(d->b2, 1, d->n_items, e->dec_buf[B2], 1, Ir);
    SDRM_LAUNCH(e, k_pad2d, dim3((unsigned)std::min<int64_t>(2048, (most + 255) / 256), 5), dim3(256), 0, st, sg);
    HIP_TRY(e, hipGetLastError());
  }
  const int rows64 = round_up(n, BM);
  const int cfg = choose_cfg(e->tune, rows64, e->tune.nt32_max_rows);
  {
    GemmArgs a{};
    a.C = e->dec_buf[HID]; a.ldc = Hp; a.bias = e->dec_buf[B1];
    HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_TANH>(a, e->dec_buf[Z], Lp, e->dec_buf[W1], Lp, rows64, Hp, Lp, st,
                                                     Prof{nullptr, 0, 0.0}, cfg)));
  }
  GemmArgs a{};
  a.C = out; a.ldc = d->n_items; a.bias = e->dec_buf[B2];
  a.rows_valid = n; a.cols_valid = d->n_items;
  HIP_TRY(e, (gemm_forward<XF_NONE, EPI_BIAS_G>(a, e->dec_buf[HID], Hp, e->dec_buf[W2], Hp, rows64, Ip, Hp, st, Prof{nullptr, 0, 0.0}, cfg)));
  return SDRM_OK;
}

int check_decoder(sdrm_engine* e, const sdrm_vae_decoder* d, const float* z, int n, const char* who) {
  if (!e || !d || !z || !d->w1 || !d->b1 || !d->w2 || !d->b2) return fail(e, SDRM_ERR_ARG, std::string(who) + ": null pointer");
  if (n < 1 || n > (1 << 22) || d->latent < 1 || d->latent > 4096 || d->hidden < 1 || d->hidden > 16384 || d->n_items < 1 ||
      d->n_items > (1 << 20))
    return fail(e, SDRM_ERR_SHAPE, std::string(who) + ": n, latent, hidden or n_items outside the supported envelope");
  return SDRM_OK;
}

}  // namespace

int sdrm_vae_decode(sdrm_engine* e, const sdrm_vae_decoder* dec, const float* z, int n, float* out, void* stream) {
  if (int rc = check_decoder(e, dec, z, n, "sdrm_vae_decode")) return rc;
  if (!out) return fail(e, SDRM_ERR_ARG, "sdrm_vae_decode: null output");
  if (int jr = join_chains(e, (hipStream_t)stream)) return jr;
  return decode_launches(e, dec, z, n, out, (hipStream_t)stream);
}

int sdrm_equal_sparsity(sdrm_engine* e, const float* x, int64_t n, double q, uint8_t* out, float* threshold, void* stream) {
  if (!e || !x) return fail(e, SDRM_ERR_ARG, "sdrm_equal_sparsity: null pointer");
  if (n < 1) return fail(e, SDRM_ERR_SHAPE, "sdrm_equal_sparsity: n < 1");
  if (!(q >= 0.0 && q <= 1.0)) return fail(e, SDRM_ERR_ARG, "sdrm_equal_sparsity: q outside [0,1]");
  if (((uintptr_t)x & 15u) || (out && ((uintptr_t)out & 3u)))
    return fail(e, SDRM_ERR_ARG, "sdrm_equal_sparsity: x must be 16-byte and out 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int64_t r0, r1;
  float gamma;
  quantile_ranks(n, q, r0, r1, gamma);
  SDRM_LAUNCH(e, k_select_init, dim3(8), dim3(256), 0, st, e->sel, r0, r1);
  HIP_TRY(e, hipGetLastError());
  return select_and_binarize(e, x, n, gamma, out, threshold, st);
}

// ---------------------------------------------------------------------------------------------
// Recall@k / NDCG@k against held-out interactions (utilities.py:116-171), csrc/rank.h.
int sdrm_rank_metrics(sdrm_engine* e, const float* scores, int U, int I, const int64_t* held_indptr,
                      const int32_t* held_indices, const int64_t* train_indptr, const int32_t* train_indices,
                      const int32_t* ks_host, int nk, const double* tp, const double* idcg, double* recall, double* ndcg,
                      void* stream) {
  if (!e || !scores || !held_indptr || !held_indices || !ks_host || !tp || !idcg || !recall || !ndcg)
    return fail(e, SDRM_ERR_ARG, "sdrm_rank_metrics: null pointer");
  if ((train_indptr == nullptr) != (train_indices == nullptr))
    return fail(e, SDRM_ERR_ARG, "sdrm_rank_metrics: train_indptr and train_indices go together");
  if (U < 1 || I < 1 || I > 36864) return fail(e, SDRM_ERR_SHAPE, "sdrm_rank_metrics: U < 1 or I outside [1, 36864]");
  if (nk < 1 || nk > RANK_MAX_NK) return fail(e, SDRM_ERR_ARG, "sdrm_rank_metrics: nk outside [1, 8]");
  RankArgs a{};
  a.scores = scores; a.U = U; a.I = I;
  a.held_indptr = held_indptr; a.held_indices = held_indices;
  a.train_indptr = train_indptr; a.train_indices = train_indices;
  a.nk = nk; a.kmax = 0;
  for (int q = 0; q < nk; ++q) {
    if (ks_host[q] < 1 || ks_host[q] > RANK_MAX_K || ks_host[q] > I)
      return fail(e, SDRM_ERR_ARG, "sdrm_rank_metrics: k outside [1, min(128, I)]");
    a.ks[q] = ks_host[q];
    if (ks_host[q] > a.kmax) a.kmax = ks_host[q];
  }
  a.tp = tp; a.idcg = idcg; a.recall = recall; a.ndcg = ndcg;
  const size_t lds = (size_t)I * sizeof(float);
  if (lds > 48 * 1024)
    HIP_TRY(e, hipFuncSetAttribute((const void*)k_rank_metrics, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SDRM_LAUNCH(e, k_rank_metrics, dim3(U), dim3(256), lds, (hipStream_t)stream, a);
  HIP_TRY(e, hipGetLastError());
  return SDRM_OK;
}

// ---------------------------------------------------------------------------------------------
int64_t sdrm_launch_count(const sdrm_engine* e) { return e ? e->n_launches : -1; }
int sdrm_profile_classes(void) { return PC_COUNT; }
const char* sdrm_profile_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }

int sdrm_profile_begin(sdrm_engine* e, int capacity) {
  if (!e || capacity < 1) return SDRM_ERR_ARG;
  while ((int)e->prof_ev.size() < 2 * capacity) {
    hipEvent_t ev;
    HIP_TRY(e, hipEventCreate(&ev));
    e->prof_ev.push_back(ev);
  }
  e->prof_cap = capacity;
  e->prof_cls.clear();
  e->prof_flops.clear();
  for (int i = 0; i < 16; ++i) { e->prof_ms[i] = 0; e->prof_fl[i] = 0; e->prof_n[i] = 0; }
  e->prof_on = true;
  return SDRM_OK;
}

int sdrm_profile_only(sdrm_engine* e, int cls) {
  if (!e || cls >= PC_COUNT) return SDRM_ERR_ARG;
  e->prof_only = cls < 0 ? -1 : cls;
  return SDRM_OK;
}

int sdrm_profile_end(sdrm_engine* e, void* stream) {
  if (!e) return SDRM_ERR_ARG;
  e->prof_on = false;
  if (int jr = join_chains(e, (hipStream_t)stream)) return jr;
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
  for (size_t i = 0; i < e->prof_cls.size(); ++i) {
    float ms = 0.f;
    HIP_TRY(e, hipEventElapsedTime(&ms, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]));
    const int c = e->prof_cls[i];
    e->prof_ms[c] += ms; e->prof_fl[c] += e->prof_flops[i]; e->prof_n[c] += 1;
  }
  return SDRM_OK;
}

int sdrm_profile_get(const sdrm_engine* e, int cls, double* total_ms, int64_t* launches, double* flops) {
  if (!e || cls < 0 || cls >= PC_COUNT) return SDRM_ERR_ARG;
  if (total_ms) *total_ms = e->prof_ms[cls];
  if (launches) *launches = e->prof_n[cls];
  if (flops) *flops = e->prof_fl[cls];
  return SDRM_OK;
}

// ---------------------------------------------------------------------------------------------
int sdrm_debug_gemm(int variant, int cfg, const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
  if (!A || !B || !C || cfg < 0 || cfg >= N_TILE_CFGS) return SDRM_ERR_ARG;
  if (M % 32 || N % 32 || K % 32 || variant < 0 || variant > 2) return SDRM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  // The kernel reads whole tiles without bounds checks, so stage the caller's matrices in zero-padded
  // scratch (test hook only: allocates and synchronises).
  const int Mp = round_up(M, 128), Np = round_up(N, 128), Kp = round_up(K, 128);
  const int ar = variant == 2 ? Kp : Mp, ac = variant == 2 ? Mp : K;     // A as stored: [M,K] or [K,M]
  const int br = variant == 0 ? Np : Kp, bc = variant == 0 ? K : Np;     // B as stored: [N,K] or [K,N]
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  if (dalloc(&dA, (size_t)ar * ac) != hipSuccess || dalloc(&dB, (size_t)br * bc) != hipSuccess ||
      dalloc(&dC, (size_t)Mp * Np) != hipSuccess)
    return SDRM_ERR_NOMEM;
  const int a_rows = variant == 2 ? K : M, a_cols = variant == 2 ? M : K;
  const int b_rows = variant == 0 ? N : K, b_cols = variant == 0 ? K : N;
  hipError_t rc = hipMemcpy2DAsync(dA, (size_t)ac * 4, A, (size_t)a_cols * 4, (size_t)a_cols * 4, a_rows, hipMemcpyDeviceToDevice, st);
  if (rc == hipSuccess)
    rc = hipMemcpy2DAsync(dB, (size_t)bc * 4, B, (size_t)b_cols * 4, (size_t)b_cols * 4, b_rows, hipMemcpyDeviceToDevice, st);
  GemmArgs a{};
  a.C = dC; a.ldc = Np; a.K = K; a.kchunk = K;
  a.A = dA; a.lda = ac; a.limA = M; a.B = dB; a.ldb = bc; a.limB = N;
  if (rc == hipSuccess) {
    const Prof np{nullptr, 0, 0.0};
    if (variant == 0) rc = launch_gemm<LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
    else if (variant == 1) rc = launch_gemm<LD_KCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
    else rc = launch_gemm<LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
  }
  if (rc == hipSuccess)
    rc = hipMemcpy2DAsync(C, (size_t)N * 4, dC, (size_t)Np * 4, (size_t)N * 4, M, hipMemcpyDeviceToDevice, st);
  if (rc == hipSuccess) rc = hipStreamSynchronize(st);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return rc == hipSuccess ? SDRM_OK : SDRM_ERR_HIP;
}

/* Timing variant of the hook for tools/gemm_tune.py: same kernel, `reps` launches on pre-padded scratch,
 * returns the mean microseconds per launch measured with HIP events on `stream`. */
#ifdef SDRM_STAMPS
// diagnostic build: stamps of the engine's own batched weight-gradient launch (the last one run)
extern "C" int sdrm_debug_stamp_class(int cls) { g_stamp_class = cls; return SDRM_OK; }
extern "C" int sdrm_debug_wgrad_stamps_begin(int max_blocks) {
  if (g_wgrad_stamps) (void)hipFree(g_wgrad_stamps);
  g_wgrad_stamps = nullptr;
  if (dalloc(&g_wgrad_stamps, (size_t)8 * max_blocks) != hipSuccess) return SDRM_ERR_NOMEM;
  g_wgrad_stamps_cap = max_blocks; g_wgrad_stamps_n = 0;
  return SDRM_OK;
}
extern "C" int sdrm_debug_wgrad_stamps_read(unsigned long long* host_out, int max_blocks) {
  if (!g_wgrad_stamps || hipDeviceSynchronize() != hipSuccess) return SDRM_ERR_STATE;
  const int nb = g_wgrad_stamps_n < max_blocks ? g_wgrad_stamps_n : max_blocks;
  if (hipMemcpy(host_out, g_wgrad_stamps, (size_t)nb * 64, hipMemcpyDeviceToHost) != hipSuccess) return SDRM_ERR_HIP;
  return nb;
}
// host_out: 8 slots per block (gemm.h); `warm` launches first, back to back, so that the clock the stamps see is the one
// the chip holds under this load (MI355X_MICROARCH.md, DVFS give-back item 6: >= 2 s)
extern "C" int sdrm_debug_gemm_stamps(int variant, int cfg, int M, int N, int K, unsigned long long* host_out, int max_blocks,
                                      int warm) {
  if (cfg < 0 || cfg >= N_TILE_CFGS) return SDRM_ERR_ARG;
  const int Mp = round_up(M, 128), Np = round_up(N, 128), Kp = round_up(K, 128);
  const int ar = variant == 2 ? Kp : Mp, ac = variant == 2 ? Mp : K;
  const int br = variant == 0 ? Np : Kp, bc = variant == 0 ? K : Np;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  unsigned long long* dS = nullptr;
  if (dalloc(&dA, (size_t)ar * ac) != hipSuccess || dalloc(&dB, (size_t)br * bc) != hipSuccess ||
      dalloc(&dC, (size_t)Mp * Np) != hipSuccess || dalloc(&dS, (size_t)8 * max_blocks) != hipSuccess)
    return SDRM_ERR_NOMEM;
  GemmArgs a{};
  a.C = dC; a.ldc = Np; a.K = K; a.kchunk = K; a.stamps = dS;
  a.A = dA; a.lda = ac; a.limA = M; a.B = dB; a.ldb = bc; a.limB = N;
  hipError_t rc = hipSuccess;
  for (int i = 0; i < 3 + (warm > 0 ? warm : 0) && rc == hipSuccess; ++i) {
    const Prof np{nullptr, 0, 0.0};
    if (variant == 0) rc = launch_gemm<LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, nullptr, np, cfg);
    else if (variant == 1) rc = launch_gemm<LD_KCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, nullptr, np, cfg);
    else rc = launch_gemm<LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, nullptr, np, cfg);
  }
  if (rc == hipSuccess) rc = hipDeviceSynchronize();
  const int nb = a.nblocks < max_blocks ? a.nblocks : max_blocks;
  if (rc == hipSuccess) rc = hipMemcpy(host_out, dS, (size_t)nb * 64, hipMemcpyDeviceToHost);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dS);
  return rc == hipSuccess ? nb : SDRM_ERR_HIP;
}
#endif

int sdrm_debug_gemm_time(int variant, int cfg, int M, int N, int K, int reps, float* us_out, void* stream) {
  if (!us_out || reps < 1 || M % 32 || N % 32 || K % 32 || variant < 0 || variant > 2 || cfg < 0 || cfg >= N_TILE_CFGS) return SDRM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int Mp = round_up(M, 128), Np = round_up(N, 128), Kp = round_up(K, 128);
  const int ar = variant == 2 ? Kp : Mp, ac = variant == 2 ? Mp : K;
  const int br = variant == 0 ? Np : Kp, bc = variant == 0 ? K : Np;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  if (dalloc(&dA, (size_t)ar * ac) != hipSuccess || dalloc(&dB, (size_t)br * bc) != hipSuccess ||
      dalloc(&dC, (size_t)Mp * Np) != hipSuccess)
    return SDRM_ERR_NOMEM;
  GemmArgs a{};
  a.C = dC; a.ldc = Np; a.K = K; a.kchunk = K;
  a.A = dA; a.lda = ac; a.limA = M; a.B = dB; a.ldb = bc; a.limB = N;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipError_t rc = hipSuccess;
  for (int i = 0; i < reps + 3 && rc == hipSuccess; ++i) {
    if (i == 3) (void)hipEventRecord(e0, st);
    const Prof np{nullptr, 0, 0.0};
    if (variant == 0) rc = launch_gemm<LD_KCONTIG, LD_KCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
    else if (variant == 1) rc = launch_gemm<LD_KCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
    else rc = launch_gemm<LD_MCONTIG, LD_MCONTIG, XF_NONE, XF_NONE, EPI_PLAIN>(a, M, N, 1, st, np, cfg);
  }
  (void)hipEventRecord(e1, st);
  if (rc == hipSuccess) rc = hipStreamSynchronize(st);
  float ms = 0.f;
  if (rc == hipSuccess) rc = hipEventElapsedTime(&ms, e0, e1);
  *us_out = ms * 1e3f / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return rc == hipSuccess ? SDRM_OK : SDRM_ERR_HIP;
}

}  // extern "C"
