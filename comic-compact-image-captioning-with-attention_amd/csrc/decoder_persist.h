// Persistent time loops of the training decoder: host-side interface of decoder_persist.hip (forward) and
// decoder_persist_bwd.hip (backward), used by comic_decoder_train_step (decoder_exec.hip).
#pragma once
#include "common.h"

struct ComicPersistFwdArgs {
  // resident operands
  const float* K_panel;   // forward panel of the LSTM kernel (comic_pack_lstm_panels, mode 0)
  const float* bias;      // [4D]
  const float* W_q;       // [D][D]
  const float* keys;      // [B][M][D]
  const float* values;    // [B][M][D] (== keys when tied)
  const float *ln_g, *ln_b, *v, *tau;
  const int32_t* lens;    // [B]
  // dropout masks (null = off)
  const float* mask_in;   // [T][B][E+A]
  const float* mask_out;  // [T][B][D]
  const float* mask_alpha;// [T][B][H][M]
  float keep_in, keep_out, keep_alpha;
  // per-step buffers, time-major
  float* xh_all;          // [Tp][B][Wd]   operand rows [x ; att ; h]; x parts and row 0 filled by the caller
  float* gates_all;       // [Tp][B][4D]
  float* cnew_all;        // [Tp][B][D]
  float* y_all;           // [Tp][B][D]
  float* q_all;           // [Tp][B][D]
  float* cs;              // [Tp+1][B][D]  row 0 = initial state
  float* hs;              // [Tp+1][B][D]
  float* att_all;         // [Tp+1][B][D]  row 0 = zeros
  float* alpha_all;       // [Tp][B][H][M]
  float* attn_hist;       // [Tp][B][H][M]
  float* ctx_all;         // [Tp][B][D]
  float* statp;           // M > 64: [Tp][B][4][M/2][4] partial LayerNorm sums of the channel quarters; sentinel-filled
  unsigned long long* stamps;   // diagnostic phase clock of workgroup 0 (null = off)
  unsigned* sync;         // kPersistSyncWords words: the error word (comic_persist_prepare clears it)
  int B, D, E, Wd, M, H, Tp;
  int method, prob, tied;
  int grp0, n_groups;     // this launch serves the 16-row groups [grp0, grp0 + n_groups) of the batch (n_groups <= 4)
  // greedy decode (GREEDY instantiation): no teacher forcing / masks / lens / saved activations
  int greedy;
  const float* emb;       // [V][E]
  const float* W_o;       // [D][ld_wo] output projection (columns < V)
  const float* b_o;       // [V]
  int V, ld_wo, start_id, end_id;
  float* argp;            // [Tp][B][132] per row 64 partial (value, column) maxima of the logits + the group's stop word; sentinel-filled
  int32_t* ids_tb;        // [Tp][B] token ids (out)
  int32_t* first_eos;     // [B] step of the first EOS (out; pre-filled with Tp)
  float* logits_tb;       // [Tp][B][V] or null
};

constexpr int kPersistSyncWords = 64;       // the error word, and from word 32 one "all my rows emitted EOS" word per group (greedy)
// "not written yet" pattern of the handed-off buffers (xh_all, y_all, q_all): comic_persist_prepare fills them
#define COMIC_PERSIST_SENTINEL 0xFFFFDEADu

bool comic_persist_fwd_supported(int B, int D, int E, int A, int M, int H, int Cv, int method, int context_layer,
                                 int tied);
bool comic_persist_fwd_bigm(int M, int tied);   // the forward loop runs in its channel-quarter form: it needs ComicPersistFwdArgs::statp
// fills the hand-off buffers (up to kPersistRanges ranges of floats, sizes multiples of 4) with the sentinel and clears the
// sync words; call BEFORE the kernels that write the x parts and the step-0 row
constexpr int kPersistRanges = 10;
struct ComicPersistRanges {
  float* p[kPersistRanges];
  long n[kPersistRanges];
};
// Riders of the prepare launch (the training step's other start-of-step chores, one launch instead of three): the forward
// MFMA panel of the LSTM kernel (comic_pack_lstm_panels, mode 0) and W_o with rows padded to Vp floats.  Null = none.
struct ComicPrologueExtra {
  const float* K;       // [Wd][4D]
  float* panel;         // forward panel, n_pack floats (null: skip)
  int D, Wd;
  long n_pack;
  const float* W_o;     // [D][V]
  float* wo_pad;        // [D][Vp] (null: skip)
  int V, Vp;
  long n_pad;
};
// n_zero: words of `sync` to clear (>= kPersistSyncWords; the training executor keeps the grouped GEMM's tickets behind them)
int comic_persist_prepare(const ComicPersistRanges& r, unsigned* sync, int n_zero, hipStream_t st,
                          const ComicPrologueExtra* extra = nullptr);
bool comic_persist_greedy_supported(int B, int D, int E, int A, int M, int H, int Cv, int V, int method,
                                    int context_layer, int tied);
int comic_persist_check_greedy(const unsigned* sync, int32_t* first_eos, hipStream_t st);
bool comic_persist_fits_device(int B);   // CUs of the current device >= workgroups of the launch
int comic_persist_fwd_launch(const ComicPersistFwdArgs& a, hipStream_t st);
// End-of-step gate: when a bounded spin of one of the step's persistent launches expired (the error word of `sync`
// is set: the step's activations and gradients are garbage), loss_rows[0] and map_loss[0] become NaN -- so both the
// sequence loss the host reduces from loss_rows and the map loss read NaN -- and every range of `r` (the gradient
// views, dfm, dim_embed) is zero-filled, so an optimiser step that follows without a host check applies no gradient.
// A healthy step costs one launch whose workgroups read the error word and return.
struct ComicGateRanges {
  float* p[16];
  long n[16];
};
// step_flag (may be null): 1 / 0 = this step was voided / is healthy; sticky (may be null): += 1 per voided step.
int comic_persist_gate(const unsigned* sync, float* loss_rows, float* map_loss, const ComicGateRanges& r, float* step_flag,
                       float* sticky, hipStream_t st);

// ---- backward loop (decoder_persist_bwd.hip) ---------------------------------------------------------------------------
struct ComicPersistBwdArgs {
  const float* K;         // [Wd][4D] LSTM kernel, row-major (rows E.. are read in place: a row is one operand feature)
  const float* W_q;       // [D][D]
  const float* keys;      // [B][M][D] (tied: also the values)
  const float *ln_g, *ln_b, *v, *tau;
  const int32_t* lens;
  const float* mask_in;   // [T][B][E+A] or null
  const float* mask_out;  // [T][B][D] or null
  const float* mask_alpha;// [T][B][H][M] or null
  float keep_in, keep_out, keep_alpha;
  // saved by the forward loop
  const float* q_all;     // [Tp][B][D]
  const float* alpha_all; // [Tp][B][H][M]
  const float* gates_all; // [Tp][B][4D]
  const float* cs;        // [Tp+1][B][D]
  const float* cnew_all;  // [Tp][B][D]
  const float* dy_all;    // [Tp][B][D]  d cell output from the logits path
  const float* dmap;      // [Tp][B][M] map-loss term of d alpha_d, or null
  // hand-off buffers, sentinel-filled by the caller
  float* dq_part;         // [Tp][B][4][D]   the four partials of d q_t
  float* dq_sum;          // [Tp][groups][32][16][16]      blocked (see decoder_persist_bwd.hip), groups = ceil(B / 16)
  float* dg_blk;          // [Tp][groups][128][16][16]     blocked
  float* dstate;          // [Tp][B][2D]   d att | d h of the step's operand row
  float* dotp;            // M > 28: [Tp][B][4][16] per-head partial sums of alpha * d alpha of the row quarters; sentinel-filled
  // outputs
  float* dq_all;          // [Tp][B][D]
  float* dg_all;          // [Tp][B][4D]   row-major (operand of the d K / d b / d emb reductions after the loop)
  float* dc;              // [B][D] gradient of the initial cell state
  float* dh;              // [B][D]
  float* dkeys;           // [B][M][D]
  float* pgrad;           // [4B][3D+1] rows of [d v | d ln_g | d ln_b | d tau]
  unsigned* sync;         // the error word (cleared by the forward launch of the same step)
  unsigned long long* stamps;   // diagnostic phase clock of workgroup 0 (null = off)
  int B, E, M, H, Tp;
  int method;
  int grp0, n_groups;     // as in ComicPersistFwdArgs
  int own_rows;           // 1: the own-rows form of the large memories (template MODE 2) whatever M is (dkeys zero-filled by the caller)
};

bool comic_persist_bwd_supported(int B, int D, int E, int A, int M, int H, int Cv, int method, int prob,
                                 int context_layer, int tied);
unsigned long long* comic_persist_stamps(int which, int Tp, hipStream_t st);
void comic_persist_set_stamps(bool on);   // diagnostic phase clocks for this thread's launches (COMIC_DEC_STAMPS)
int comic_persist_bwd_launch(const ComicPersistBwdArgs& a, hipStream_t st);
int comic_dropout_rows(float* x, const float* mask, float keep, long rows, int cols, int ld, hipStream_t st);
