// Native executors for the attention-LSTM decoder: one teacher-forced forward+backward
// (XE or SCST-weighted), greedy decode and beam search.  They issue the step kernels of
// decoder.hip / gemm.hip / decode.hip on one HIP stream with no host synchronisation and no
// allocation (caller-provided workspace), so a whole step can be captured in a hipGraph.
//
// Mirrors (not translates) the reference graph builders:
//   ModelBase._decoder_rnn / _decoder_rnn_scst ...... src/model_base.py:109-269
//   rnn_decoder_training / _search / _beam_search ... common/ops_rnn.py:49-243
//   MultiHeadAttentionWrapperV3.call ................ common/ops_rnn.py:660-755
//   _train_caption_model (losses) ................... src/model_base.py:325-405
// Differences by design: the x-independent pieces are hoisted out of the time loop
// (embedding lookup for all steps, output projection + cross-entropy as one batched GEMM,
// all weight-gradient GEMMs batched over time), states are kept per step for the backward
// pass instead of TF's TensorArray stack, and dropout masks are explicit inputs.
#include <atomic>
#include <stdlib.h>

#include <algorithm>
#include <string.h>

#include "common.h"
#include "decoder_math.h"
#include "decoder_persist.h"
#include "gemm_group.h"
#include "lstm_prep.h"
#include "lstm_stream_dev.h"

// internal cross-file entries (gemm.hip, decoder.hip)
int comic_gemm_f32_ws(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                      int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta, void* ws, int64_t ws_bytes,
                      hipStream_t st);
int comic_xent_ex(float* logits, const int32_t* targets_bt, const float* coef_bt, const float* wmask_bt,
                  const int32_t* lens, float* loss_rows, float* dlogits, int32_t* ids_tb, int t_rows, int t_stride,
                  int B, int V, hipStream_t st);

int comic_attn_fwd_ex(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                      const float* ln_g, const float* ln_b, const float* v, const float* tau, const float* mask_alpha,
                      float keep_alpha, float* alpha, float* alpha_d, float* ctx, const int32_t* lens, int t,
                      const float* att_prev, float* att_next, float* xh_next, int xh_ld, const float* mask_next,
                      int mask_ld, float keep_in, int q_parts, float* q_out, float* scores_ws, hipStream_t st,
                      int mem_div = 1);
int comic_attn_splits(int B, int M);
long comic_attn_bwd_scratch(int B, int H, int M, int D);
int comic_colsum_ws(const float* in, float* out, int rows, int cols, float beta, float* ws, hipStream_t st);
int comic_gemm_f32_partial(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int trans_b,
                           void* ws, int64_t ws_bytes, int* S_out, hipStream_t st);
int comic_lstm_gates_bwd_ex(const float* gates_act, const float* c_prev, const float* c_new, const float* dy,
                            const float* dy_part, int S, const float* mask_out, float keep_out, const int32_t* lens,
                            int t, float* dc_state, float* dh_state, float* dg, int B, int D, hipStream_t st);
int comic_attn_bwd_ex(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                      const float* ln_g, const float* ln_b, const float* v, const float* tau, const float* alpha,
                      const float* mask_alpha, float keep_alpha, const float* dctx, const float* dmap, float* dq,
                      float* dkeys, float* dvalues, float* pgrad, const int32_t* lens, int t, hipStream_t st,
                      int pgrad_overwrite, float* ws_s = nullptr, float* ws_d = nullptr);
int comic_lstm_gates_fwd_ex(const float* g, const float* c_prev, const float* h_prev, float* gates_act, float* c_new,
                            float* y, const float* mask_out, float keep_out, const int32_t* lens, int t,
                            float* c_state, float* h_state, int B, int D, float* xh_next, int xh_ld, int S,
                            const float* bias, hipStream_t st);

int comic_xent_maploss(float* logits, const int32_t* targets_bt, const float* coef_bt, const float* wmask_bt,
                       const int32_t* lens, float* loss_rows, float* dlogits, int ld_dl, int32_t* ids_tb, int t_rows,
                       int t_stride, int B, int V, const float* hist, float* dmap, float* partial, float* map_loss,
                       unsigned* ticket, int H, int M, float scale, hipStream_t st);
int comic_embed_bwd_set(const int32_t* ids, const float* dout, float* dtable, int rows, int E, int V, hipStream_t st);
// gemm.hip
int comic_gemm_bf16x3_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta, void* ws,
                           int64_t ws_bytes, hipStream_t st);
// decode.hip
bool comic_beam_logits_supported(int D, int V, int R, int W);
int64_t comic_beam_logits_pack_bytes(int D, int V);
int64_t comic_beam_logits_partial_floats(int D, int V, int R, int W, int max_steps);
int comic_beam_logits_begin(float* partials, int B, int W, int V, int max_steps, hipStream_t st);
int comic_beam_pack_wo(const float* W_o, const float* b_o, int ld, void* wo_frag, int D, int V, hipStream_t st);
int comic_beam_logits_step(const float* y, const void* y_frag_in, const void* wo_frag, float* partials, float* log_probs,
                           int32_t* finished, int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores,
                           int32_t* steps_executed, int t, int max_steps, int B, int W, int D, int V, int end_id,
                           const LstmPrepArgs* prep, hipStream_t st);
bool comic_lstm_stream_supported(int D, int E, int A, int R);
int64_t comic_lstm_stream_kfrag_floats(int D, int Wd);
int64_t comic_lstm_stream_xfrag_floats(int R, int Wd);
int64_t comic_lstm_stream_part_bytes(int D, int Wd, int R);
int comic_lstm_stream_pack(const float* K, void* k_frag, int D, int Wd, hipStream_t st);
int comic_lstm_stream_step(const float* table, const int32_t* ids, const int32_t* parent, int W, const float* att_src,
                           const float* h_src, const float* c_src, const void* k_frag, const float* bias, void* x_frag,
                           float* c_in, float* part, int64_t part_bytes, float* c_state, float* h_state, float* y,
                           void* y_frag, int R, int E, int A, int D, int V, int skip_prep, hipStream_t st);
bool comic_stream_gemm_supported(int Kin, int N, int R);
int64_t comic_stream_gemm_wfrag_floats(int Kin, int N);
int64_t comic_stream_gemm_part_bytes(int Kin, int N, int R);
int comic_stream_gemm_pack(const float* Wm, void* w_frag, int Kin, int N, hipStream_t st);
int comic_stream_gemm(const void* x_frag, const void* w_frag, float* part, int64_t part_bytes, int R, int Kin, int N, int* S,
                      hipStream_t st);
LstmStreamArgs comic_stream_gemm_args(const void* x_frag, const void* w_frag, float* part, int R, int Kin, int N, int* S,
                                      int* n_wg, int* lds_bytes, int max_wg);
int comic_beam_logits_chunks(int V);
int comic_beam_logits_launch(const float* y, const void* y_frag_in, const void* wo_frag, float* partials, int max_steps, int B,
                             int W, int D, int V, const LstmStreamArgs* q, int n_q, int q_lds, hipStream_t st);
int comic_beam_merge_launch(float* partials, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                            int32_t* parent_ids, float* scores, int32_t* steps_executed, int t, int max_steps, int B, int W,
                            int V, int end_id, const LstmPrepArgs* prep, hipStream_t st);
int comic_stream_gemm2(const void* x_frag, const void* w_a, float* part_a, int N_a, int* S_a, const void* w_b, float* part_b,
                       int N_b, int* S_b, int64_t part_bytes_each, int R, int Kin, hipStream_t st);
bool comic_beam_step_small_supported(int V, int W);
int comic_beam_counters_zero(void* cnt, int n, hipStream_t st);
int comic_beam_step_small(const float* logits, const float* bias, int S, int ld, long slice_stride, float* log_probs,
                          int32_t* finished, int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores, int B,
                          int W, int V, int end_id, void* cnt, int32_t* steps_executed, int t, int max_steps,
                          const LstmPrepArgs* prep, hipStream_t st);
int comic_beam_step_lp(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                       int32_t* parent_ids, float* scores, int B, int W, int V, int end_id, float lpw, hipStream_t st);
int comic_beam_step_ws(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                       int32_t* parent_ids, float* scores, int B, int W, int V, int end_id, void* ws, int64_t ws_bytes,
                       hipStream_t st);
// decoder_fused.hip
int comic_fused_step_supported(int D, int Wd);
long comic_lstm_panel_floats(int D, int Wd, int mode);
int comic_pack_lstm_panels(const float* K, float* fwd_panel, float* bwd_panel, int D, int Wd, hipStream_t st);
int comic_lstm_step_fused(const float* xh, int ld_xh, const float* K, const float* bias, const float* c_prev,
                          const float* h_prev, float* gates_act, float* c_new, float* y, const float* mask_out,
                          float keep_out, const int32_t* lens, int t, float* c_state, float* h_state, float* xh_next,
                          int xh_ld, int B, int D, int Wd, hipStream_t st);
int comic_pack_wq_panel(const float* Wq, float* panel, int D, hipStream_t st);
int comic_lstm_grad_fused(const float* dq, const float* wq_panel, const float* gates_act, const float* c_prev,
                          const float* c_new, const float* dy, const float* mask_out, float keep_out,
                          const int32_t* lens, int t, float* dc_state, float* dh_state, float* dg, int B, int D,
                          hipStream_t st);
int comic_input_grad_fused(const float* dg, const float* K, const float* mask, float keep, float* demb, float* datt,
                           float* dh, const int32_t* lens, int t, int carry, int B, int E, int A, int D,
                           hipStream_t st);

// the reference's other recurrent cells (cells.hip)
int comic_cell_ln_stride(int D);
int comic_ln_lstm_fwd(const float* g, int S, const float* ln, const float* c_prev, const float* h_prev, float* gates_act,
                      float* xhat, float* rstd, float* c_new, float* y, const float* mask_out, float keep_out,
                      const int32_t* lens, int t, float* c_state, float* h_state, int B, int D, float* xh_next, int xh_ld,
                      hipStream_t st);
int comic_ln_lstm_bwd(const float* gates_act, const float* xhat, const float* rstd, const float* ln, const float* c_prev,
                      const float* c_new, const float* dy, const float* dy_part, int S, const float* mask_out,
                      float keep_out, const int32_t* lens, int t, float* dc, float* dh, float* dg, float* pgrad, int B, int D,
                      hipStream_t st);
int comic_ln_lstm_scatter(const float* sums, float* cell_ln_grad, int D, float beta, hipStream_t st);
int comic_gru_gates_fwd(const float* g1, int S, const float* bias, const float* h_prev, const float* xh, int xh_ld, float* ru,
                        int ld_ru, float* xh2, int xh2_ld, int B, int D, int EA, hipStream_t st);
int comic_gru_out_fwd(const float* g2, int S, const float* bias, const float* ru, int ld_ru, const float* h_prev,
                      float* cand, int ld_cand, float* y, const float* mask_out, float keep_out, const int32_t* lens, int t,
                      float* h_state, float* xh_next, int xh_ld, int B, int D, hipStream_t st);
int comic_gru_bwd1(const float* dy, const float* dy_part, int S, const float* mask_out, float keep_out, const int32_t* lens,
                   int t, float* dh_state, const float* ru, int ld_ru, const float* cand, int ld_cand, const float* h_prev,
                   float* dpre, int ld_dpre, int B, int D, hipStream_t st);
int comic_gru_bwd2(float* dxh2, int ld, const float* ru, int ld_ru, const float* h_prev, float* dpre, int ld_dpre, int B,
                   int D, int EA, hipStream_t st);

namespace {

// Executor switches come with the call (comic_decoder_desc::flags, COMIC_DEC_*): the library reads no environment.
// The entry points latch them for the calling thread; the helpers below are what the executors consult.
thread_local uint32_t g_dec_flags = 0;
struct FlagScope {
  explicit FlagScope(const comic_decoder_desc* d) {
    g_dec_flags = d ? d->flags : 0u;
    // LN_LSTM / GRU run on the per-step launch chain: the fused / streaming / persistent kernels are BasicLSTMCell's
    if (d && d->cell != COMIC_CELL_LSTM) g_dec_flags |= COMIC_DEC_NO_FUSED_STEP | COMIC_DEC_NO_PERSIST | COMIC_DEC_NO_PERSIST_BWD;
    comic_persist_set_stamps((g_dec_flags & COMIC_DEC_STAMPS) != 0);
  }
  ~FlagScope() { comic_persist_set_stamps(false); }
};
bool split_attn_bwd_enabled() { return !(g_dec_flags & COMIC_DEC_NO_SPLIT_ATTN_BWD); }
bool fused_step_enabled() { return !(g_dec_flags & COMIC_DEC_NO_FUSED_STEP); }
bool persist_enabled() { return !(g_dec_flags & COMIC_DEC_NO_PERSIST); }
bool persist_bwd_enabled() { return !(g_dec_flags & COMIC_DEC_NO_PERSIST_BWD); }
bool beam_logits_enabled() { return !(g_dec_flags & COMIC_DEC_NO_BEAM_LOGITS); }
bool lstm_stream_enabled() { return !(g_dec_flags & COMIC_DEC_NO_LSTM_STREAM); }
bool group_gemm_enabled() { return !(g_dec_flags & (COMIC_DEC_NO_GROUP_GEMM | COMIC_DEC_EXACT_GEMM)); }

// A second stream inside the training executor: the weight-gradient products after the backward loop are independent
// chains of mid-sized GEMMs and small reductions; two lanes fill each other's tails and launch gaps
// (COMIC_DEC_ONE_LANE: one stream).  Fork / join with events, so a hipGraph capture of the step takes both lanes.
struct SideLane {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
SideLane* side_lane() {
  static SideLane lanes[64];
  if (g_dec_flags & COMIC_DEC_ONE_LANE) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideLane* L = &lanes[dev];
  if (!L->s) {
    if (hipStreamCreateWithFlags(&L->s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&L->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&L->join, hipEventDisableTiming) != hipSuccess) {
      L->s = nullptr;
      return nullptr;
    }
  }
  return L;
}

thread_local void* g_splitk_ws = nullptr;   // split-K scratch of the GEMM helpers below (one per lane)

// Fork / join of the side lane as a scope: whatever path leaves the scope (every RC(...) can return early), the lane
// is joined back into the caller's stream and the split-K scratch pointer is restored -- a capture of the step is
// never left with an unjoined stream, and a later step is never ordered behind the stragglers of a failed one.
struct LaneScope {
  SideLane* L;
  hipStream_t st;
  void* saved_ws;
  bool forked = false;
  int rc = 0;
  LaneScope(SideLane* lane, hipStream_t main, void* lane_ws) : L(lane), st(main), saved_ws(g_splitk_ws) {
    if (!L) return;
    if (hipEventRecord(L->fork, st) != hipSuccess || hipStreamWaitEvent(L->s, L->fork, 0) != hipSuccess) {
      comic_set_error("train_step: cannot fork the side lane");
      rc = 2;
      return;
    }
    forked = true;
    g_splitk_ws = lane_ws;
  }
  hipStream_t lane() const { return forked ? L->s : st; }
  void main_ws() { g_splitk_ws = saved_ws; }      // launches on the caller's stream from here on
  int join() {
    g_splitk_ws = saved_ws;
    if (!forked) return 0;
    forked = false;
    if (hipEventRecord(L->join, L->s) != hipSuccess || hipStreamWaitEvent(st, L->join, 0) != hipSuccess) {
      comic_set_error("train_step: cannot join the side lane");
      return 2;
    }
    return 0;
  }
  ~LaneScope() { (void)join(); }
};

#define RC(x)               \
  do {                      \
    int rc__ = (x);         \
    if (rc__) return rc__;  \
  } while (0)

struct Bump {
  char* base;
  size_t off, cap;
  bool ok;
  Bump(void* p, size_t c) : base((char*)p), off(0), cap(c), ok(true) {}
  template <typename T>
  T* take(size_t n) {
    const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
    if (base && off + bytes > cap) ok = false;
    T* r = base ? (T*)(base + off) : nullptr;
    off += bytes;
    return r;
  }
};

// xh_row = [ drop(x) ; drop(att) ; h ]   (cell_input_fn concat + DropoutWrapper input dropout)
__global__ void assemble_input_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                      const float* __restrict__ h, const float* __restrict__ mask, float keep,
                                      float* __restrict__ xh, int B, int E, int A, int D) {
  const int W = E + A + D;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, c = i % W;
  float v;
  if (c < E + A) {
    v = c < E ? x[(size_t)b * E + c] : att[(size_t)b * A + (c - E)];
    if (mask) v = (v / keep) * mask[(size_t)b * (E + A) + c];
  } else {
    v = h ? h[(size_t)b * D + (c - E - A)] : 0.f;
  }
  xh[i] = v;
}

// xh_all[t][b][0:E] = drop(emb[ids[t,b]]) for every step at once (embedding lookup hoisted out
// of the time loop; the attention / recurrent parts of the row are filled by the step kernels)
__global__ void embed_to_xh_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids_tb,
                                   const float* __restrict__ mask, float keep, float* __restrict__ xh, long rows,
                                   int E, int V, int EA, int Wd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * E) return;
  const long r = i / E;
  const int e = (int)(i % E);
  const int id = ids_tb[r];
  float v = (id >= 0 && id < V) ? table[(size_t)id * E + e] : 0.f;
  if (mask) v = (v / keep) * mask[(size_t)r * EA + e];
  xh[(size_t)r * Wd + e] = v;
}

// The operand rows before the time loop, one launch: x parts of EVERY step (embedding lookup + input dropout, ids read
// from the batch-major table and also written time-major for the embedding backward), and step 0's att part (zero;
// att_all[0] too) and h part (h0).
__global__ void embed_step0_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids_bt,
                                   int32_t* __restrict__ ids_tb, const float* __restrict__ mask, float keep,
                                   float* __restrict__ xh, float* __restrict__ att0, const float* __restrict__ h0, int Tp,
                                   int B, int T, int E, int A, int D, int V, float* __restrict__ xh_init,
                                   const float* __restrict__ cell_g, float* __restrict__ cell_gates,
                                   float* __restrict__ cell_cnew, float* __restrict__ c0, float* __restrict__ h0_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_x = (long)Tp * B * E;
  const int EA = E + A, Wd = E + A + D;
  if (i < n_x) {
    const long r = i / E;
    const int e = (int)(i % E), t = (int)(r / B), b = (int)(r % B);
    const int id = ids_bt[(size_t)b * T + t];
    if (e == 0) ids_tb[r] = id;
    float v = (id >= 0 && id < V) ? table[(size_t)id * E + e] : 0.f;
    if (mask) v = (v / keep) * mask[(size_t)r * EA + e];
    xh[(size_t)r * Wd + e] = v;
  } else if (i < n_x + (long)B * A) {
    const long j = i - n_x;
    att0[j] = 0.f;
    xh[(size_t)(j / A) * Wd + E + (j % A)] = 0.f;
  } else if (i < n_x + (long)B * A + (long)B * D) {
    const long j = i - n_x - (long)B * A;
    float hv;
    if (cell_g) {      // the rnn-init step's LSTM cell from a zero state (lstm_gates_fwd_kernel's arithmetic; the bias is in g)
      const int b = (int)(j / D), dd = (int)(j % D);
      const float* gr = cell_g + (size_t)b * 4 * D;
      const float si = sigmoidf_(gr[dd]), tj = tanhf(gr[D + dd]);
      const float sf = sigmoidf_(gr[2 * D + dd] + 1.0f), so = sigmoidf_(gr[3 * D + dd]);
      const float c2 = 0.f * sf + si * tj;
      hv = tanhf(c2) * so;
      float* ga = cell_gates + (size_t)b * 4 * D;
      ga[dd] = si; ga[D + dd] = tj; ga[2 * D + dd] = sf; ga[3 * D + dd] = so;
      cell_cnew[j] = c2;
      c0[j] = c2;
      h0_out[j] = hv;
    } else {
      hv = h0[j];
    }
    xh[(size_t)(j / D) * Wd + EA + (j % D)] = hv;
  } else if (xh_init && i < n_x + (long)B * A + 2L * B * D) {   // zero state of the init step: the h third of its operand rows
    const long j = i - n_x - (long)B * A - (long)B * D;
    xh_init[(size_t)(j / D) * Wd + EA + (j % D)] = 0.f;
  }
}

// inference step operand: xh[r] = [ emb[ids[r]] ; att[src(r)] ; h[src(r)] ], c_in[r] = c[src(r)] with
// src(r) = the beam-search parent of row r in the previous step (identity for greedy / step 0): the
// embedding lookup, the three state gathers and the concat of one step in a single pass.
__global__ void infer_prep_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                  const int32_t* __restrict__ parent, int W, const float* __restrict__ att,
                                  const float* __restrict__ h, const float* __restrict__ c, float* __restrict__ xh,
                                  float* __restrict__ c_in, int R, int E, int A, int D, int V,
                                  const int32_t* __restrict__ stop, int stop_t) {
  // after the loop has ended the previous step's ids / parents were never written: nothing to gather
  if (comic_stopped(stop, stop_t)) return;
  const int Wd = E + A + D, cols = Wd + D;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)R * cols) return;
  const int r = (int)(i / cols), k = (int)(i % cols);
  const int src = parent ? (r / W) * W + min(max(parent[r], 0), W - 1) : r;
  if (k < E) {
    const int id = ids[r];
    xh[(size_t)r * Wd + k] = (id >= 0 && id < V) ? table[(size_t)id * E + k] : 0.f;
  } else if (k < E + A) {
    xh[(size_t)r * Wd + k] = att[(size_t)src * A + (k - E)];
  } else if (k < Wd) {
    xh[(size_t)r * Wd + k] = h[(size_t)src * D + (k - E - A)];
  } else {
    c_in[(size_t)r * D + (k - Wd)] = c[(size_t)src * D + (k - Wd)];
  }
}

// context-layer path only: att_next = fin ? att_prev : att_cur ; xh_next[:, E:E+A] = drop(att_next)
__global__ void select_att_kernel(const float* __restrict__ prev, const float* __restrict__ cur,
                                  const int32_t* __restrict__ lens, int t, float* __restrict__ dst,
                                  float* __restrict__ xh_next, int xh_ld, const float* __restrict__ mask_next,
                                  int mask_ld, float keep, int B, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const int b = i / A, c = i % A;
  const float v = (lens && t >= lens[b]) ? prev[i] : cur[i];
  dst[i] = v;
  if (xh_next) {
    float x = v;
    if (mask_next) x = (x / keep) * mask_next[(size_t)b * mask_ld + c];
    xh_next[(size_t)b * xh_ld + c] = x;
  }
}

// dst = fin ? prev : cur      (impute_finished state select; lens NULL -> copy cur)
__global__ void select_rows_kernel(const float* __restrict__ prev, const float* __restrict__ cur,
                                   const int32_t* __restrict__ lens, int t, float* __restrict__ dst, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C;
  dst[i] = (lens && t >= lens[b]) ? prev[i] : cur[i];
}

// live = !(t >= lens[b]):  out_live = d*live ; d = d*(1-live)
__global__ void split_live_kernel(float* __restrict__ d, float* __restrict__ out_live,
                                  const int32_t* __restrict__ lens, int t, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C;
  const bool fin = lens && t >= lens[b];
  const float v = d[i];
  out_live[i] = fin ? 0.f : v;
  d[i] = fin ? v : 0.f;
}

// dxh [B, E+A+D] -> demb_t [B,E] = drop'(dxh[:, :E]); datt += drop'(dxh[:, E:E+A]); dh += dxh[:, E+A:]
// `carry`: datt holds d(att state after this step); only FINISHED rows carry it through to the
// state before the step (live rows' share went into the context inside attn_bwd).
__global__ void input_bwd_kernel(const float* __restrict__ dxh, const float* __restrict__ mask, float keep,
                                 float* __restrict__ demb, float* __restrict__ datt, float* __restrict__ dh,
                                 const int32_t* __restrict__ lens, int t, int carry, int B, int E, int A, int D,
                                 int S) {
  const int W = E + A + D;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, c = i % W;
  float v = dxh[i];
  for (int s = 1; s < S; ++s) v += dxh[(size_t)s * B * W + i];  // split-K partials of dg * K^T
  if (c < E + A) {
    if (mask) v = (v / keep) * mask[(size_t)b * (E + A) + c];
    if (c < E) {
      if (demb) demb[(size_t)b * E + c] = v;
    } else {
      float* p = datt + (size_t)b * A + (c - E);
      const bool fin = lens && t >= lens[b];
      *p = ((carry && !fin) ? 0.f : *p) + v;
    }
  } else {
    dh[(size_t)b * D + (c - E - A)] += v;
  }
}

// flat[b,t,m] = sum_h hist[t,b,h,m];  map_loss = mean((1-flat)^2)*scale;
// dmap[t,b,m] = 2*(flat-1)/(B*Tp*M)*scale.   Single workgroup (deterministic).
// two stages (fixed partition -> deterministic): per-workgroup partial sums, then one workgroup
__global__ __launch_bounds__(256) void maploss_part_kernel(const float* __restrict__ hist, float* __restrict__ dmap,
                                                           float* __restrict__ partial, int Tp, int B, int H, int M,
                                                           float scale) {
  __shared__ float red[256];
  const long n = (long)Tp * B * M;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (i < n) {
    const int m = (int)(i % M);
    const long tb = i / M;
    const float* p = hist + (size_t)tb * H * M + m;
    float f = 0.f;
    for (int h = 0; h < H; ++h) f += p[(size_t)h * M];
    const float d = 1.0f - f;
    acc = d * d;
    if (dmap) dmap[i] = 2.0f * (f - 1.0f) / (float)n * scale;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// the four attention-parameter gradients out of the column-summed [v | ln_g | ln_b | tau] row
__global__ void scatter_pgrad_kernel(const float* __restrict__ row, float* __restrict__ v, float* __restrict__ ln_g,
                                     float* __restrict__ ln_b, float* __restrict__ tau, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) {
    v[i] = row[i];
    ln_g[i] = row[D + i];
    ln_b[i] = row[2 * D + i];
  }
  if (i == 0) tau[0] = row[3 * D];
}

__global__ __launch_bounds__(256) void maploss_final_kernel(const float* __restrict__ partial, int nparts, long n,
                                                            float scale, float* __restrict__ map_loss) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) map_loss[0] = red[0] / (float)n * scale;
}

__global__ __launch_bounds__(1024) void maploss_kernel(const float* __restrict__ hist, float* __restrict__ dmap,
                                                       float* __restrict__ map_loss, int Tp, int B, int H, int M,
                                                       float scale) {
  __shared__ float red[1024];
  const long n = (long)Tp * B * M;
  float acc = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const int m = (int)(i % M);
    const long tb = i / M;
    const float* p = hist + (size_t)tb * H * M + m;
    float f = 0.f;
    for (int h = 0; h < H; ++h) f += p[(size_t)h * M];
    const float d = 1.0f - f;
    acc += d * d;
    if (dmap) dmap[i] = 2.0f * (f - 1.0f) / (float)n * scale;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && map_loss) map_loss[0] = red[0] / (float)n * scale;
}

__global__ void fill_kernel(float* p, float v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void fill_i32_kernel(int32_t* p, int32_t v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// transpose int32 [B,T] -> [T,B] (first Tp rows)
__global__ void transpose_ids_kernel(const int32_t* __restrict__ in_bt, int32_t* __restrict__ out_tb, int B, int T,
                                     int Tp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Tp * B) return;
  const int t = i / B, b = i % B;
  out_tb[i] = in_bt[(size_t)b * T + t];
}
// tile_batch: out[b*W + w, :] = in[b, :]
__global__ void tile_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int W, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long r = i / cols;
  out[i] = in[(size_t)(r / W) * cols + (i % cols)];
}
// greedy bookkeeping: first_eos[b] = min(first_eos[b], t) when ids[b] == end,
// in one workgroup, plus the loop end -- steps_done = t+1 at the first step after which
// every row has emitted EOS (dynamic_decode stops there; later steps' kernels return at once, see ComicStop)
__global__ void eos_track_done_kernel(const int32_t* __restrict__ ids, int32_t* __restrict__ first_eos, int t,
                                      int end_id, int B, int32_t* __restrict__ steps_done, int max_steps) {
  __shared__ int any_live;
  if (threadIdx.x == 0) any_live = 0;
  __syncthreads();
  if (steps_done[0] <= t) return;                 // loop already over: ids of this step were never produced
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int fe = first_eos[b];
    if (ids[b] == end_id && fe > t) {
      fe = t;
      first_eos[b] = t;
    }
    if (fe > t) any_live = 1;
  }
  __syncthreads();
  if (threadIdx.x == 0 && !any_live && steps_done[0] == max_steps) steps_done[0] = t + 1;
}
// beam bookkeeping: steps_executed = t+1 at the first step after which every beam is finished
__global__ void all_finished_kernel(const int32_t* __restrict__ finished, int32_t* __restrict__ steps_executed, int t,
                                    int n, int max_steps) {
  __shared__ int any_live;
  if (threadIdx.x == 0) any_live = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (!finished[i]) any_live = 1;
  __syncthreads();
  if (threadIdx.x == 0 && !any_live && steps_executed[0] == max_steps) steps_executed[0] = t + 1;
}

__global__ void beam_init_kernel(float* __restrict__ log_probs, int32_t* __restrict__ finished,
                                 int64_t* __restrict__ lengths, int R, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const bool first = (i % W) == 0;
  log_probs[i] = first ? 0.f : -INFINITY;
  finished[i] = first ? 0 : 1;
  lengths[i] = 0;
}

// out[r][c] = in[r][c] for c < cols, 0 in the padding columns (rows of `ld` >= cols elements)
__global__ void pad_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int cols, int ld, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / ld;
  const int c = (int)(i - r * ld);
  out[i] = c < cols ? in[r * cols + c] : 0.f;
}

inline int fill(float* p, float v, long n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p, v, n);
  COMIC_LAUNCH_CHECK("fill");
  return 0;
}
// split-K scratch of the running executor call (carved from the caller's workspace)
constexpr int64_t kSplitKBytes = 32ll << 20;

inline int gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                int ldc, int ta, int tb, float beta, hipStream_t st) {
  return comic_gemm_f32_ws(A, B, C, bias, M, N, K, lda, ldb, ldc, ta, tb, 1.0f, beta, g_splitk_ws,
                           g_splitk_ws ? kSplitKBytes : 0, st);
}

// The time-batched products (hundreds of rows: keys, logits, d logits * W_o^T, every weight gradient) go
// to the bf16 matrix cores with hi/lo-split operands (comic_gemm_f32_split3, product error ~2^-15); the
// per-step products keep exact fp32 MFMAs.  COMIC_DEC_EXACT_GEMM selects the exact kernels everywhere.
inline int gemm_big(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                    int ldc, int ta, int tb, float beta, hipStream_t st) {
  const bool on = !(g_dec_flags & COMIC_DEC_EXACT_GEMM);
  if (!on || (long)M * N < 64 * 256 || K < 64)
    return gemm(A, B, C, bias, M, N, K, lda, ldb, ldc, ta, tb, beta, st);
  return comic_gemm_bf16x3_impl(A, B, C, bias, M, N, K, lda, ldb, ldc, ta, tb, 1.0f, beta, g_splitk_ws,
                                g_splitk_ws ? kSplitKBytes : 0, st);
}

// The products of a training step that do not feed the recurrence, as ONE grouped launch (gemm_group.hip): problems are
// collected here and run together; `slab` / `tickets` come from the step's workspace (kGroupTickets counters, zeroed at
// the top of the step and left zero by every launch).
constexpr int kGroupTickets = 4096;
struct GemmGroupRun {
  ComicGemmGroup g{};
  ComicGemmProb* add(int type, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc) {
    if (g.n >= kGemmGroupMax) return nullptr;
    ComicGemmProb& p = g.p[g.n++];
    p = ComicGemmProb{};
    p.type = type; p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = 1.f; p.beta = 0.f; p.keep = 1.f;
    return &p;
  }
  // column sums of B [K][N] (ldb) -> C [N]
  ComicGemmProb* add_colsum(const float* B, float* C, int N, int K, int ldb) {
    ComicGemmProb* p = add(COMIC_GG_TN, nullptr, B, C, 1, N, K, 1, ldb, N);
    if (p) p->ones_a = 1;
    return p;
  }
  int run(void* slab, int64_t slab_cap, unsigned* tickets, hipStream_t st) {
    if (g.n == 0) return 0;
    int target = kGemmGroupTargetItems;
    for (;;) {
      int64_t need = 0;
      int nt = 0;
      const int wg = comic_gemm_group_plan(g, target, &need, &nt);
      if (wg < 0) return 2;
      if (need <= slab_cap && nt <= kGroupTickets - 8) {      // (the last words serve other launches' tickets)
        g.slab = (float*)slab;
        g.tickets = tickets;
        return comic_gemm_group_launch(g, wg, st);
      }
      COMIC_REQUIRE(target > 1, "gemm_group: split-K scratch too small");
      target = target / 2;
    }
  }
};

int check_desc(const comic_decoder_desc* d) {
  COMIC_REQUIRE(d, "decoder: null descriptor");
  COMIC_REQUIRE(d->D > 0 && d->E > 0 && d->A > 0 && d->V > 1 && d->C > 0 && d->Cg > 0 && d->H > 0 && d->M > 0,
                "decoder: bad dimensions");
  const int cv = d->fm_projection == 0 ? d->C : d->D;
  COMIC_REQUIRE(d->Cv == cv, "decoder: Cv must be %d for fm_projection %d", cv, d->fm_projection);
  const int a = (d->fm_projection == 0 && !d->context_layer) ? d->C : d->D;
  COMIC_REQUIRE(d->A == a, "decoder: attention size A must be %d", a);
  COMIC_REQUIRE(d->cell >= COMIC_CELL_LSTM && d->cell <= COMIC_CELL_GRU, "decoder: unknown cell %d", d->cell);
  return 0;
}
int check_cell_params(const comic_decoder_desc* d, const comic_decoder_params* p) {
  COMIC_REQUIRE(p && p->K, "decoder: null cell kernel");
  if (d->cell == COMIC_CELL_LSTM) COMIC_REQUIRE(p->b, "decoder: LSTM needs its bias");
  if (d->cell == COMIC_CELL_LN_LSTM) COMIC_REQUIRE(p->cell_ln, "decoder: LN_LSTM needs cell_ln");
  if (d->cell == COMIC_CELL_GRU) COMIC_REQUIRE(p->b && p->K_c && p->b_c, "decoder: GRU needs b, K_c, b_c");
  return 0;
}

comic_attn_desc attn_desc(const comic_decoder_desc* d, int rows) {
  comic_attn_desc a;
  a.B = rows; a.M = d->M; a.D = d->D; a.H = d->H; a.Cv = d->Cv;
  a.method = d->method; a.prob = d->prob; a.tied = d->fm_projection == 2;
  return a;
}

// keys / values for `rows` feature maps (ops_rnn.py:440-477)
int memory_projections(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm, int rows,
                       float* keys, float* values_buf, const float** values, hipStream_t st) {
  RC(gemm_big(fm, p->W_m, keys, nullptr, rows * d->M, d->D, d->C, d->C, d->D, d->D, 0, 0, 0.f, st));
  if (d->fm_projection == 2) {
    *values = keys;
  } else if (d->fm_projection == 1) {
    RC(gemm_big(fm, p->W_v, values_buf, nullptr, rows * d->M, d->D, d->C, d->C, d->D, d->D, 0, 0, 0.f, st));
    *values = values_buf;
  } else {
    *values = fm;
  }
  return 0;
}

struct InitBufs {
  float *x, *xh, *g, *gates, *c_new;
  float *lnx = nullptr, *lnr = nullptr;      // LN_LSTM, training: normalised rows [rows][5D] and 1/std [rows][8] of the init step
};

// _get_rnn_init (model_base.py:651-689) -> c0,h0 [rows,D]
int rnn_init_fwd(const comic_decoder_desc* d, const comic_decoder_params* p, const float* im_embed, int rows,
                 const float* mask_init, InitBufs& ib, float* c0, float* h0, hipStream_t st) {
  const int D = d->D, EA = d->E + d->A;
  if (d->init_method == 1) {
    RC(gemm(im_embed, p->W_init, h0, nullptr, rows, D, d->Cg, d->Cg, D, D, 0, 0, 0.f, st));
    RC(fill(c0, 0.f, (long)rows * D, st));
    return 0;
  }
  RC(gemm(im_embed, p->W_init, ib.x, nullptr, rows, EA, d->Cg, d->Cg, EA, EA, 0, 0, 0.f, st));
  RC(comic_dropout_apply(ib.x, mask_init, d->keep_in, ib.xh, (int64_t)rows * EA, (void*)st));
  // zero initial state: only the first E+A rows of the cell's kernel(s) contribute
  if (d->cell == COMIC_CELL_LN_LSTM) {
    RC(gemm(ib.xh, p->K, ib.g, nullptr, rows, 4 * D, EA, EA, 4 * D, 4 * D, 0, 0, 0.f, st));
    return comic_ln_lstm_fwd(ib.g, 1, p->cell_ln, nullptr, nullptr, ib.gates, ib.lnx, ib.lnr, ib.c_new, nullptr, nullptr, 1.f,
                             nullptr, 0, c0, h0, rows, D, nullptr, 0, st);
  }
  if (d->cell == COMIC_CELL_GRU) {           // r*h = 0: the candidate sees [x ; 0]; gates / candidate kept in ib.gates
    float* g2 = ib.g + (size_t)rows * 2 * D;
    RC(gemm(ib.xh, p->K, ib.g, nullptr, rows, 2 * D, EA, EA, 2 * D, 2 * D, 0, 0, 0.f, st));
    RC(comic_gru_gates_fwd(ib.g, 1, p->b, nullptr, ib.xh, EA, ib.gates, 4 * D, nullptr, 0, rows, D, EA, st));
    RC(gemm(ib.xh, p->K_c, g2, nullptr, rows, D, EA, EA, D, D, 0, 0, 0.f, st));
    RC(comic_gru_out_fwd(g2, 1, p->b_c, ib.gates, 4 * D, nullptr, ib.gates + 2 * D, 4 * D, nullptr, nullptr, 1.f, nullptr, 0,
                         h0, nullptr, 0, rows, D, st));
    return fill(c0, 0.f, (long)rows * D, st);
  }
  RC(gemm(ib.xh, p->K, ib.g, p->b, rows, 4 * D, EA, EA, 4 * D, 4 * D, 0, 0, 0.f, st));
  RC(comic_lstm_gates_fwd(ib.g, nullptr, nullptr, ib.gates, ib.c_new, nullptr, nullptr, nullptr, 1.f, nullptr, 0, c0,
                          h0, rows, D, (void*)st));
  return 0;
}

// one wrapper step without dropout / imputing (inference)
struct StepBufs {
  float *xh, *g, *y, *q, *alpha, *ctx, *c2, *h2, *att2;
  float* xh2 = nullptr;                       // GRU: [x ; att ; r*h]
};
int infer_step(const comic_decoder_desc* d, const comic_decoder_params* p, const comic_attn_desc& ad,
               const float* keys, const float* values, const float* x, const float* c, const float* h,
               const float* att, StepBufs& sb, float* alpha_d_out, int rows, hipStream_t st) {
  const int D = d->D, E = d->E, A = d->A, Wd = E + A + D;
  hipLaunchKernelGGL(assemble_input_kernel, dim3(cdiv(rows * Wd, 256)), dim3(256), 0, st, x, att, h, nullptr, 1.f,
                     sb.xh, rows, E, A, D);
  COMIC_LAUNCH_CHECK("assemble_input");
  if (d->cell == COMIC_CELL_LN_LSTM) {
    RC(gemm(sb.xh, p->K, sb.g, nullptr, rows, 4 * D, Wd, Wd, 4 * D, 4 * D, 0, 0, 0.f, st));
    RC(comic_ln_lstm_fwd(sb.g, 1, p->cell_ln, c, h, nullptr, nullptr, nullptr, nullptr, sb.y, nullptr, 1.f, nullptr, 0, sb.c2,
                         sb.h2, rows, D, nullptr, 0, st));
  } else if (d->cell == COMIC_CELL_GRU) {
    float* ru = sb.g + (size_t)rows * 2 * D;             // sb.g: [rows][2D] product, then [rows][2D] r | u
    RC(gemm(sb.xh, p->K, sb.g, nullptr, rows, 2 * D, Wd, Wd, 2 * D, 2 * D, 0, 0, 0.f, st));
    RC(comic_gru_gates_fwd(sb.g, 1, p->b, h, sb.xh, Wd, ru, 2 * D, sb.xh2, Wd, rows, D, E + A, st));
    RC(gemm(sb.xh2, p->K_c, sb.q, nullptr, rows, D, Wd, Wd, D, D, 0, 0, 0.f, st));      // sb.q: free until the query product
    RC(comic_gru_out_fwd(sb.q, 1, p->b_c, ru, 2 * D, h, nullptr, 0, sb.y, nullptr, 1.f, nullptr, 0, sb.h2, nullptr, 0, rows, D,
                         st));
    RC(fill(sb.c2, 0.f, (long)rows * D, st));            // the state is h alone; c rides along as zeros
  } else {
    RC(gemm(sb.xh, p->K, sb.g, p->b, rows, 4 * D, Wd, Wd, 4 * D, 4 * D, 0, 0, 0.f, st));
    RC(comic_lstm_gates_fwd(sb.g, c, h, nullptr, nullptr, nullptr, sb.y, nullptr, 1.f, nullptr, 0, sb.c2, sb.h2, rows,
                            D, (void*)st));
  }
  RC(gemm(sb.y, p->W_q, sb.q, nullptr, rows, D, D, D, D, D, 0, 0, 0.f, st));
  RC(comic_attn_step_fwd(&ad, keys, values, sb.q, p->ln_g, p->ln_b, p->v, p->tau, nullptr, 1.f, sb.alpha,
                         alpha_d_out, sb.ctx, (void*)st));
  if (d->context_layer) {
    RC(gemm(sb.ctx, p->W_a, sb.att2, nullptr, rows, D, d->Cv, d->Cv, D, D, 0, 0, 0.f, st));
  }
  return 0;
}

// The same wrapper step on the fused kernels: operand prep (embedding + parent gather + concat),
// LSTM product + gates, query product left as split-K partials for the attention kernel.
// Reads the previous step's raw outputs (c_src, h_src, att_src) through `parent`.
// operands of the streaming step kernels (lstm_stream.hip): packed LSTM kernel, the step's operand rows and outputs as
// hi / lo fragments, packed W_q (null: exact split-K product)
struct StreamBufs {
  const void* kfrag;
  void* xfrag;
  void* yfrag;
  const void* wqfrag;
  int skip_prep = 0;       // the operand rows of this step were prepared by the previous step's beam merge
  // a second product over the same y, launched with the query projection (vocabulary projection at a small V):
  const void* wofrag = nullptr;   // packed W_o, or null
  int wo_N = 0;
  float* wo_part = nullptr;       // out: its K-slice partials (second half of the split-K scratch)
  int wo_S = 0;                   // out: their count
  int q_S = 0;                    // > 0: the query partials are already in the split-K scratch (they rode another launch)
};
// first half: operand prep + LSTM product + cell -> c2, h2, y (and y as fragments on the streaming path)
int infer_step_lstm(const comic_decoder_desc* d, const comic_decoder_params* p, const float* kpanel, const int32_t* ids,
                    const int32_t* parent, int W, const float* c_src, const float* h_src, const float* att_src,
                    StepBufs& sb, float* c_in, int rows, hipStream_t st, const StreamBufs* sm) {
  const int D = d->D, E = d->E, A = d->A, Wd = E + A + D;
  if (sm) {         // many rows: the kernel streamed once for all of them (lstm_stream.hip)
    RC(comic_lstm_stream_step(p->emb, ids, parent, W, att_src, h_src, c_src, sm->kfrag, p->b, sm->xfrag, c_in,
                              (float*)g_splitk_ws, kSplitKBytes, sb.c2, sb.h2, sb.y, sm->yfrag, rows, E, A, D, d->V, sm->skip_prep, st));
  } else {
    const long n = (long)rows * (Wd + D);
    hipLaunchKernelGGL(infer_prep_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p->emb, ids, parent, W,
                       att_src, h_src, c_src, sb.xh, c_in, rows, E, A, D, d->V, g_comic_stop.p, g_comic_stop.t);
    COMIC_LAUNCH_CHECK("infer_prep");
    RC(comic_lstm_step_fused(sb.xh, Wd, kpanel, p->b, c_in, nullptr, nullptr, nullptr, sb.y, nullptr, 1.f, nullptr, 0,
                             sb.c2, sb.h2, nullptr, 0, rows, D, Wd, st));
  }
  return 0;
}
// second half: query projection (split-K partials in g_splitk_ws) + attention -> alpha, context (the next step's
// attention state); reads y, writes nothing the vocabulary projection of the step looks at
int infer_step_attend(const comic_decoder_desc* d, const comic_decoder_params* p, const comic_attn_desc& ad,
                      const float* keys, const float* values, StepBufs& sb, float* alpha_d_out, int rows, hipStream_t st,
                      StreamBufs* sm, int mem_div) {
  const int D = d->D;
  int S = 1;
  float* part = (float*)g_splitk_ws;
  if (sm && sm->q_S > 0) {
    S = sm->q_S;
  } else if (sm && sm->wqfrag && sm->wofrag) {
    sm->wo_part = (float*)((char*)g_splitk_ws + kSplitKBytes / 2);
    RC(comic_stream_gemm2(sm->yfrag, sm->wqfrag, part, D, &S, sm->wofrag, sm->wo_part, sm->wo_N, &sm->wo_S, kSplitKBytes / 2,
                          rows, D, st));
  } else if (sm && sm->wqfrag) RC(comic_stream_gemm(sm->yfrag, sm->wqfrag, part, kSplitKBytes, rows, D, D, &S, st));
  else RC(comic_gemm_f32_partial(sb.y, p->W_q, rows, D, D, D, D, 0, part, kSplitKBytes, &S, st));
  // large memories (Inception-V1 Mixed_4f: M = 196): the attention step in its split form; its [rows][H][M] scratch is the
  // pre-activation gate buffer, which the fused LSTM step never materialises
  float* attn_ws = ((long)d->H * d->M <= 4L * D) ? sb.g : nullptr;
  RC(comic_attn_fwd_ex(&ad, keys, values, part, p->ln_g, p->ln_b, p->v, p->tau, nullptr, 1.f, sb.alpha, alpha_d_out,
                       sb.ctx, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, 0, 1.f, S, nullptr, attn_ws, st, mem_div));
  if (d->context_layer) {
    RC(gemm(sb.ctx, p->W_a, sb.att2, nullptr, rows, D, d->Cv, d->Cv, D, D, 0, 0, 0.f, st));
  }
  return 0;
}
int infer_step_fused(const comic_decoder_desc* d, const comic_decoder_params* p, const comic_attn_desc& ad,
                     const float* keys, const float* values, const float* kpanel, const int32_t* ids,
                     const int32_t* parent, int W, const float* c_src, const float* h_src, const float* att_src,
                     StepBufs& sb, float* c_in, float* alpha_d_out, int rows, hipStream_t st,
                     StreamBufs* sm = nullptr, int mem_div = 1) {
  RC(infer_step_lstm(d, p, kpanel, ids, parent, W, c_src, h_src, att_src, sb, c_in, rows, st, sm));
  return infer_step_attend(d, p, ad, keys, values, sb, alpha_d_out, rows, st, sm, mem_div);
}

}  // namespace

thread_local int g_train_path = 0;
extern "C" int comic_decoder_train_path(void) { return g_train_path; }
// fault injection for the tests of the end-of-step gate: the next comic_decoder_train_step that runs a persistent loop
// behaves as if one of its bounded waits had expired (one shot)
thread_local int g_greedy_path = 0;
extern "C" int comic_decoder_greedy_path(void) { return g_greedy_path; }
thread_local int g_beam_path = 0;
extern "C" int comic_decoder_beam_path(void) { return g_beam_path; }

extern "C" int64_t comic_decoder_train_workspace(const comic_decoder_desc* d, int B, int T) {
  if (!d) return -1;
  Bump w(nullptr, 0);
  const long D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H, Cv = d->Cv, Wd = E + A + D;
  const long TB = (long)T * B;
  w.take<float>(B * M * D); w.take<float>(B * M * D);              // keys, values
  w.take<float>(B * (E + A)); w.take<float>(B * (E + A));          // init x, xh
  w.take<float>(B * 4 * D); w.take<float>(B * 4 * D); w.take<float>(B * D);  // init g, gates, c_new
  w.take<float>(TB * E); w.take<int32_t>(TB);                      // emb_all, ids_tb
  w.take<float>((TB + B) * Wd); w.take<float>(B * 4 * D);          // xh_all (+ the init step's operand rows), g_tmp
  w.take<float>(TB * 4 * D);                                       // gates_act
  w.take<float>((TB + B) * D); w.take<float>((TB + B) * D);        // cs, hs
  w.take<float>(TB * D); w.take<float>(TB * D); w.take<float>(TB * D);  // c_new, y, q
  w.take<float>(TB * H * M);                                       // alpha
  w.take<float>(TB * Cv); w.take<float>(B * D); w.take<float>((TB + B) * A);  // ctx, att_new, att
  w.take<float>(TB * ((V + 3) / 4 * 4));                           // dlogits (rows padded to a multiple of 4 columns)
  w.take<float>(D * ((V + 3) / 4 * 4));                            // W_o with padded rows (grouped GEMM launches)
  w.take<float>(TB * D); w.take<float>(TB * D); w.take<float>((TB + B) * 4 * D);    // dy_all, dq_all, dg_all (+ the init step's rows)
  w.take<float>(B * Wd); w.take<float>(B * D); w.take<float>(B * D);          // dxh, dc, dh
  w.take<float>(B * A); w.take<float>(B * A); w.take<float>(B * Cv);          // datt, datt_live, dctx
  w.take<float>(TB * E); w.take<float>(B * M * D); w.take<float>(B * M * Cv); // demb, dkeys, dvalues
  w.take<float>(TB * (3 * D + 1)); w.take<float>(TB * M);                      // pgrad rows, dmap
  w.take<float>(B * (E + A));                                                  // dx_init
  w.take<char>(kSplitKBytes);                                                  // split-K partials
  w.take<char>(kSplitKBytes);                                                  // ... of the second gradient lane
  w.take<float>(comic_lstm_panel_floats((int)D, (int)Wd, 0));                  // LSTM kernel panels (fused step)
  w.take<float>(comic_lstm_panel_floats((int)D, (int)Wd, 1));
  w.take<float>(D * D);                                                        // W_q panel
  w.take<unsigned>(kPersistSyncWords + kGroupTickets);                         // persistent loops: error word; grouped GEMM: tile tickets
  w.take<float>(TB * 4 * D); w.take<float>((long)T * ((B + 15) / 16) * 16 * 4 * D);  // persistent backward: d q partials, d gates (blocked)
  w.take<float>((long)T * ((B + 15) / 16) * 16 * D);                                  // summed d q (blocked)
  w.take<float>(TB * 2 * D);                                                   // d att | d h
  w.take<float>(4 * B * (3 * D + 1));                                          // its parameter-gradient rows
  w.take<float>(TB * 64);                                                      // ... per-head dot products of the row quarters (M > 28)
  if (comic_persist_fwd_bigm((int)M, d->fm_projection == 2)) w.take<float>(TB * 8 * M);   // forward loop, channel-quarter form: LayerNorm sums of the quarters
  if (d->cell == COMIC_CELL_LN_LSTM) {           // normalised rows, 1/std and LayerNorm gradient rows of every step + the init step
    w.take<float>((TB + B) * 5 * D); w.take<float>((TB + B) * 8); w.take<float>((TB + B) * 10 * D); w.take<float>(10 * D);
  } else if (d->cell == COMIC_CELL_GRU) {        // [x ; att ; r*h] of every step, the two d-operand products of a step, bias sums
    w.take<float>(TB * Wd); w.take<float>(2 * B * Wd); w.take<float>(4 * D);
  }
  return (int64_t)w.off;
}

extern "C" int comic_decoder_train_step(const comic_decoder_desc* d, const comic_decoder_params* p,
                                        const comic_decoder_params* gr, const float* fm, const float* im_embed,
                                        const int32_t* inputs_bt, const int32_t* targets_bt, const float* wmask_bt,
                                        const float* coef_bt, const int32_t* lens, int B, int T, int Tp,
                                        const float* mask_init_in, const float* mask_in, const float* mask_out,
                                        const float* mask_alpha, float* logits_tb, int32_t* ids_tb, float* attn_hist,
                                        float* loss_rows, float* map_loss, float* dfm, float* dim_embed,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
  RC(check_desc(d));
  FlagScope flag_scope__(d);
  COMIC_REQUIRE(p && gr && fm && im_embed && inputs_bt && targets_bt && wmask_bt && coef_bt && lens,
                "train_step: null input");
  COMIC_REQUIRE(logits_tb && ids_tb && attn_hist && loss_rows && map_loss && workspace, "train_step: null output");
  COMIC_REQUIRE(B > 0 && T > 0 && Tp > 0 && Tp <= T, "train_step: bad B/T/Tp (%d %d %d)", B, T, Tp);
  COMIC_REQUIRE(workspace_bytes >= comic_decoder_train_workspace(d, B, T), "train_step: workspace too small");
  COMIC_REQUIRE(d->keep_in >= 1.f || (mask_in && (d->init_method == 1 || mask_init_in)),
                "train_step: input dropout enabled but no mask given");
  COMIC_REQUIRE(d->keep_out >= 1.f || mask_out, "train_step: output dropout enabled but no mask given");
  COMIC_REQUIRE(d->keep_alpha >= 1.f || mask_alpha, "train_step: attention dropout enabled but no mask given");
  hipStream_t st = (hipStream_t)stream;
  const int D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H, Cv = d->Cv, EA = E + A, Wd = E + A + D;
  const long TB = (long)T * B;
  const bool drop_in = d->keep_in < 1.f, drop_out = d->keep_out < 1.f, drop_al = d->keep_alpha < 1.f;
  Bump w(workspace, (size_t)workspace_bytes);
  float* keys = w.take<float>((long)B * M * D);
  float* values_buf = w.take<float>((long)B * M * D);
  InitBufs ib;
  ib.x = w.take<float>((long)B * EA); ib.xh = w.take<float>((long)B * EA);
  ib.g = w.take<float>((long)B * 4 * D); ib.gates = w.take<float>((long)B * 4 * D); ib.c_new = w.take<float>((long)B * D);
  float* emb_all = w.take<float>(TB * E);
  int32_t* in_tb = w.take<int32_t>(TB);
  float* xh_all = w.take<float>((TB + B) * Wd);
  float* g_tmp = w.take<float>((long)B * 4 * D);
  float* gates_all = w.take<float>(TB * 4 * D);
  float* cs = w.take<float>((TB + B) * D);
  float* hs = w.take<float>((TB + B) * D);
  float* cnew_all = w.take<float>(TB * D);
  float* y_all = w.take<float>(TB * D);
  float* q_all = w.take<float>(TB * D);
  float* alpha_all = w.take<float>(TB * H * M);
  float* ctx_all = w.take<float>(TB * Cv);
  float* att_new = w.take<float>((long)B * D);
  float* att_all = w.take<float>((TB + B) * A);
  const int Vp = (V + 3) / 4 * 4;                 // d logits / W_o rows padded to 16 bytes: every product loads them 16 bytes at a time
  float* dlogits = w.take<float>(TB * Vp);
  float* wo_pad = w.take<float>((long)D * Vp);
  float* dy_all = w.take<float>(TB * D);
  float* dq_all = w.take<float>(TB * D);
  float* dg_all = w.take<float>((TB + B) * 4 * D);
  float* dxh = w.take<float>((long)B * Wd);
  float* dc = w.take<float>((long)B * D);
  float* dh = w.take<float>((long)B * D);
  float* datt = w.take<float>((long)B * A);
  float* datt_live = w.take<float>((long)B * A);
  float* dctx = w.take<float>((long)B * Cv);
  float* demb = w.take<float>(TB * E);
  float* dkeys = w.take<float>((long)B * M * D);
  float* dvalues_buf = w.take<float>((long)B * M * Cv);
  float* pgrad = w.take<float>(TB * (3 * D + 1));   // one attention-parameter gradient row per (step, batch row)
  float* dmap = w.take<float>(TB * M);
  float* dx_init = w.take<float>((long)B * EA);
  g_splitk_ws = w.take<char>(kSplitKBytes);
  void* splitk_ws_b = w.take<char>(kSplitKBytes);
  float* kpanel_f = w.take<float>(comic_lstm_panel_floats(D, Wd, 0));
  float* kpanel_b = w.take<float>(comic_lstm_panel_floats(D, Wd, 1));
  float* wq_panel = w.take<float>((long)D * D);
  unsigned* persist_sync = w.take<unsigned>(kPersistSyncWords + kGroupTickets);
  unsigned* gg_tickets = persist_sync + kPersistSyncWords;
  const long TB16 = (long)T * ((B + 15) / 16) * 16;
  float* dq_part = w.take<float>(TB * 4 * D);
  float* dg_blk = w.take<float>(TB16 * 4 * D);
  float* dq_sum = w.take<float>(TB16 * D);
  float* dstate = w.take<float>(TB * 2 * D);
  float* pgrad4 = w.take<float>((long)4 * B * (3 * D + 1));
  float* dotp = w.take<float>(TB * 64);
  float* statp = comic_persist_fwd_bigm(M, d->fm_projection == 2) ? w.take<float>(TB * 8 * M) : nullptr;
  const int cell = d->cell;
  float *lnx_all = nullptr, *lnr_all = nullptr, *lnpg = nullptr, *cell_tmp = nullptr, *xh2_all = nullptr, *gru_dxh = nullptr;
  if (cell == COMIC_CELL_LN_LSTM) {
    lnx_all = w.take<float>((TB + B) * 5 * D); lnr_all = w.take<float>((TB + B) * 8);
    lnpg = w.take<float>((TB + B) * 10 * D); cell_tmp = w.take<float>(10L * D);
    ib.lnx = lnx_all + TB * 5 * D; ib.lnr = lnr_all + TB * 8;      // the init step's rows sit behind the time steps'
  } else if (cell == COMIC_CELL_GRU) {
    xh2_all = w.take<float>(TB * Wd); gru_dxh = w.take<float>(2L * B * Wd); cell_tmp = w.take<float>(4L * D);
  }
  COMIC_REQUIRE(w.ok, "train_step: workspace overflow");
  RC(check_cell_params(d, p));
  // COMIC_DEC_PHASE_FWD / _BWD: the step in two calls over the same workspace -- everything up to the logits (no loss
  // coefficient enters it), then loss + backward.  The SCST step runs the first under the host's reward computation.
  const bool do_fwd = !(d->flags & COMIC_DEC_PHASE_BWD), do_bwd = !(d->flags & COMIC_DEC_PHASE_FWD);
  COMIC_REQUIRE(do_fwd || do_bwd, "train_step: both phase flags set");

  const comic_attn_desc ad = attn_desc(d, B);
  const float* values = d->fm_projection == 2 ? keys : d->fm_projection == 1 ? values_buf : fm;   // (= what memory_projections reports)
  const bool fused = fused_step_enabled() && comic_fused_step_supported(D, Wd);
  const bool fused_q = fused && D % 16 == 0;
  // the time loops as persistent launches (decoder_persist.hip, decoder_persist_bwd.hip) when the shape allows it
  const bool persist = fused && persist_enabled() &&
                       comic_persist_fwd_supported(B, D, E, A, M, H, Cv, d->method, d->context_layer, ad.tied) &&
                       comic_persist_fits_device(B);
  const bool persist_b = persist && persist_bwd_enabled() &&
                         comic_persist_bwd_supported(B, D, E, A, M, H, Cv, d->method, d->prob, d->context_layer, ad.tied);
  g_train_path = (persist ? 1 : 0) | (persist_b ? 2 : 0);
  // scratch of the split attention kernels (large memories: comic_attn_splits workgroups per batch row): the d q
  // partials of the persistent backward loop, free whenever the per-step kernels run ([Tp][B][4][D] >= 2 x [B][H][M])
  float* attn_ws = (!persist_b && comic_attn_splits(B, M) > 1 &&
                    TB * 4 * D >= (long)B * H * M + comic_attn_bwd_scratch(B, H, M, D)) ? dq_part : nullptr;
  bool prologue_rides = false;
  if (persist && do_fwd) {   // every hand-off buffer of the step starts as "not written yet"; the error word as zero
    ComicPersistRanges pr{};
    const long n16 = (long)Tp * ((B + 15) / 16) * 16 * D;
    pr.p[0] = xh_all; pr.n[0] = (long)Tp * B * Wd;
    pr.p[1] = y_all; pr.n[1] = (long)Tp * B * D;
    pr.p[2] = q_all; pr.n[2] = (long)Tp * B * D;
    if (statp) {                       // the quarters' partial LayerNorm sums (decoder_persist.hip, BIGM): [Tp][B][4][M/2][4]
      pr.p[8] = statp; pr.n[8] = (long)Tp * B * 8 * M;
    }
    if (persist_b) {
      pr.p[3] = dq_part; pr.n[3] = (long)Tp * B * 4 * D;
      pr.p[4] = dg_blk; pr.n[4] = 4 * n16;
      pr.p[5] = dstate; pr.n[5] = (long)Tp * B * 2 * D;
      pr.p[6] = dq_sum; pr.n[6] = n16;
      pr.p[7] = dotp; pr.n[7] = (long)Tp * B * 64;
    }
    // grouped path: the forward panel of the LSTM kernel and the padded W_o ride on the same launch
    ComicPrologueExtra px{};
    prologue_rides = group_gemm_enabled() && cell == COMIC_CELL_LSTM && fused;
    if (prologue_rides) {
      px.K = p->K; px.panel = kpanel_f; px.D = D; px.Wd = Wd; px.n_pack = comic_lstm_panel_floats(D, Wd, 0);
      if (Vp != V) { px.W_o = p->W_o; px.wo_pad = wo_pad; px.V = V; px.Vp = Vp; px.n_pad = (long)D * Vp; }
    }
    RC(comic_persist_prepare(pr, persist_sync, kPersistSyncWords + kGroupTickets, st, prologue_rides ? &px : nullptr));
  }
  // ------------------------------------------------------------------ forward ------------
  // grp: the products outside the time loops as grouped launches (gemm_group.hip).  The rnn init step then keeps its
  // operand rows [drop(x_init) ; 0] and its d gates in row block Tp of xh_all / dg_all, so that d K and d b are ONE
  // product over (Tp + 1) * B rows.
  // (products with a short reduction -- a handful of rows in all -- keep the separate launches, whose small shapes run the
  // exact-fp32 kernels: gemm_big's rule)
  const bool grp = group_gemm_enabled() && cell == COMIC_CELL_LSTM && (long)Tp * B >= 64 && (long)B * M >= 64 && B >= 16;
  const int ldl = grp ? Vp : V;                                // row stride of d logits
  const float* wo_g = (grp && Vp != V) ? wo_pad : p->W_o;      // W_o with rows of ldl floats
  float* xh_init = xh_all + (size_t)Tp * B * Wd;
  float* dg_init = dg_all + (size_t)Tp * B * 4 * D;
  void* const gg_slab = g_splitk_ws;                 // both lanes' split-K blocks: consecutive in the workspace
  const int64_t gg_slab_cap = 2 * kSplitKBytes;
  SideLane* L = grp ? nullptr : side_lane();
  if (do_fwd) {
  if (grp) {
    if (!persist) COMIC_REQUIRE(hipMemsetAsync(gg_tickets, 0, sizeof(unsigned) * kGroupTickets, st) == hipSuccess, "train_step: memset");
    if (fused && !(prologue_rides && persist_b)) RC(comic_pack_lstm_panels(p->K, prologue_rides ? nullptr : kpanel_f, persist_b ? nullptr : kpanel_b, D, Wd, st));
    if (fused_q && !persist_b) RC(comic_pack_wq_panel(p->W_q, wq_panel, D, st));
    if (Vp != V && !prologue_rides) {
      const long n = (long)D * Vp;
      hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p->W_o, wo_pad, V, Vp, n);
      COMIC_LAUNCH_CHECK("pad W_o");
    }
    GemmGroupRun g1;
    g1.add(COMIC_GG_NN, fm, p->W_m, keys, B * M, D, d->C, d->C, D, D);
    if (d->fm_projection == 1) g1.add(COMIC_GG_NN, fm, p->W_v, values_buf, B * M, D, d->C, d->C, D, D);
    if (d->init_method == 1) {
      g1.add(COMIC_GG_NN, im_embed, p->W_init, hs, B, D, d->Cg, d->Cg, D, D);
    } else {
      ComicGemmProb* q = g1.add(COMIC_GG_NN, im_embed, p->W_init, xh_init, B, EA, d->Cg, d->Cg, EA, Wd);
      if (drop_in) { q->mask = mask_init_in; q->ld_mask = EA; q->keep = d->keep_in; }
    }
    RC(g1.run(gg_slab, gg_slab_cap, gg_tickets, st));
    if (d->init_method == 1) {
      RC(fill(cs, 0.f, (long)B * D, st));
    } else {
      // zero initial state: only the first E+A rows of the cell's kernel contribute
      GemmGroupRun g2;
      g2.add(COMIC_GG_NN, xh_init, p->K, ib.g, B, 4 * D, EA, Wd, 4 * D, 4 * D)->bias = p->b;
      RC(g2.run(gg_slab, gg_slab_cap, gg_tickets, st));      // (the cell itself: inside the operand-row launch below)
    }
  } else {
    LaneScope lane(L, st, splitk_ws_b);
    RC(lane.rc);
    hipStream_t sl = lane.lane();
    // weight panels of the fused step kernels (the persistent backward reads K and W_q in place): needed by the time
    // loop only, so they are packed on the side lane too
    if (fused) RC(comic_pack_lstm_panels(p->K, kpanel_f, persist_b ? nullptr : kpanel_b, D, Wd, sl));
    if (fused_q && !persist_b) RC(comic_pack_wq_panel(p->W_q, wq_panel, D, sl));
    RC(rnn_init_fwd(d, p, im_embed, B, drop_in ? mask_init_in : nullptr, ib, cs, hs, sl));
    lane.main_ws();
    RC(memory_projections(d, p, fm, B, keys, values_buf, &values, st));
    RC(lane.join());
  }
  // operand rows before the loop: the x part of every step (embedding lookup + input dropout, hoisted) and step 0's
  // att part (zero: dropout of 0 is 0) and h part (h0)
  {
    const bool zi = grp && d->init_method != 1;
    const long n = (long)Tp * B * E + (long)B * A + (long)B * D * (zi ? 2 : 1);
    hipLaunchKernelGGL(embed_step0_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p->emb, inputs_bt, in_tb,
                       drop_in ? mask_in : nullptr, d->keep_in, xh_all, att_all, hs, Tp, B, T, E, A, D, V,
                       zi ? xh_init : (float*)nullptr, zi ? ib.g : (const float*)nullptr, ib.gates, ib.c_new, cs, hs);
    COMIC_LAUNCH_CHECK("embed_step0");
  }
  if (persist) {
    ComicPersistFwdArgs pa{};
    pa.K_panel = kpanel_f; pa.bias = p->b; pa.W_q = p->W_q; pa.keys = keys; pa.values = values;
    pa.ln_g = p->ln_g; pa.ln_b = p->ln_b; pa.v = p->v; pa.tau = p->tau; pa.lens = lens;
    pa.mask_in = drop_in ? mask_in : nullptr; pa.mask_out = drop_out ? mask_out : nullptr;
    pa.mask_alpha = drop_al ? mask_alpha : nullptr;
    pa.keep_in = d->keep_in; pa.keep_out = d->keep_out; pa.keep_alpha = d->keep_alpha;
    pa.xh_all = xh_all; pa.gates_all = gates_all; pa.cnew_all = cnew_all; pa.y_all = y_all; pa.q_all = q_all;
    pa.cs = cs; pa.hs = hs; pa.att_all = att_all; pa.alpha_all = alpha_all; pa.attn_hist = attn_hist;
    pa.ctx_all = ctx_all; pa.sync = persist_sync;
    pa.statp = statp;
    pa.B = B; pa.D = D; pa.E = E; pa.Wd = Wd; pa.M = M; pa.H = H; pa.Tp = Tp;
    pa.method = d->method; pa.prob = d->prob; pa.tied = ad.tied;
    const int n_grp = (B + 15) / 16;                     // a launch serves up to four 16-row groups (256 CUs)
    for (int g0 = 0; g0 < n_grp; g0 += 4) {
      pa.grp0 = g0;
      pa.n_groups = std::min(4, n_grp - g0);
      RC(comic_persist_fwd_launch(pa, st));
    }
  }
  for (int t = 0; t < (persist ? 0 : Tp); ++t) {
    float* xh_t = xh_all + (size_t)t * B * Wd;
    float* xh_n = (t + 1 < Tp) ? xh_all + (size_t)(t + 1) * B * Wd : nullptr;
    const float* c_prev = cs + (size_t)t * B * D;
    const float* h_prev = hs + (size_t)t * B * D;
    const float* att_prev = att_all + (size_t)t * B * A;
    float* att_next = att_all + (size_t)(t + 1) * B * A;
    const float* mask_n = (drop_in && xh_n) ? mask_in + (size_t)(t + 1) * B * EA + E : nullptr;
    int S1 = 1, S2 = 1;
    float* part = (float*)g_splitk_ws;
    float* y_t = y_all + (size_t)t * B * D;
    if (fused) {
      RC(comic_lstm_step_fused(xh_t, Wd, kpanel_f, p->b, c_prev, h_prev, gates_all + (size_t)t * B * 4 * D,
                               cnew_all + (size_t)t * B * D, y_t, drop_out ? mask_out + (size_t)t * B * D : nullptr,
                               d->keep_out, lens, t, cs + (size_t)(t + 1) * B * D, hs + (size_t)(t + 1) * B * D,
                               xh_n ? xh_n + EA : nullptr, Wd, B, D, Wd, st));
    } else if (cell == COMIC_CELL_LN_LSTM) {
      RC(comic_gemm_f32_partial(xh_t, p->K, B, 4 * D, Wd, Wd, 4 * D, 0, part, kSplitKBytes, &S1, st));
      RC(comic_ln_lstm_fwd(part, S1, p->cell_ln, c_prev, h_prev, gates_all + (size_t)t * B * 4 * D,
                           lnx_all + (size_t)t * B * 5 * D, lnr_all + (size_t)t * B * 8, cnew_all + (size_t)t * B * D, y_t,
                           drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t,
                           cs + (size_t)(t + 1) * B * D, hs + (size_t)(t + 1) * B * D, B, D, xh_n ? xh_n + EA : nullptr, Wd, st));
    } else if (cell == COMIC_CELL_GRU) {         // gates_all[t]: r | u | candidate | -
      float* ga = gates_all + (size_t)t * B * 4 * D;
      float* xh2_t = xh2_all + (size_t)t * B * Wd;
      RC(comic_gemm_f32_partial(xh_t, p->K, B, 2 * D, Wd, Wd, 2 * D, 0, part, kSplitKBytes, &S1, st));
      RC(comic_gru_gates_fwd(part, S1, p->b, h_prev, xh_t, Wd, ga, 4 * D, xh2_t, Wd, B, D, EA, st));
      RC(comic_gemm_f32_partial(xh2_t, p->K_c, B, D, Wd, Wd, D, 0, part, kSplitKBytes, &S1, st));
      RC(comic_gru_out_fwd(part, S1, p->b_c, ga, 4 * D, h_prev, ga + 2 * D, 4 * D, y_t,
                           drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t,
                           hs + (size_t)(t + 1) * B * D, xh_n ? xh_n + EA : nullptr, Wd, B, D, st));
    } else {
      RC(comic_gemm_f32_partial(xh_t, p->K, B, 4 * D, Wd, Wd, 4 * D, 0, part, kSplitKBytes, &S1, st));
      RC(comic_lstm_gates_fwd_ex(part, c_prev, h_prev, gates_all + (size_t)t * B * 4 * D,
                                 cnew_all + (size_t)t * B * D, y_t,
                                 drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t,
                                 cs + (size_t)(t + 1) * B * D, hs + (size_t)(t + 1) * B * D, B, D,
                                 xh_n ? xh_n + EA : nullptr, Wd, S1, p->b, st));
    }
    float* q_t = q_all + (size_t)t * B * D;
    RC(comic_gemm_f32_partial(y_t, p->W_q, B, D, D, D, D, 0, part, kSplitKBytes, &S2, st));
    float* ctx_t = ctx_all + (size_t)t * B * Cv;
    const float* mal = drop_al ? mask_alpha + (size_t)t * B * H * M : nullptr;
    if (!d->context_layer) {
      RC(comic_attn_fwd_ex(&ad, keys, values, part, p->ln_g, p->ln_b, p->v, p->tau, mal, d->keep_alpha,
                           alpha_all + (size_t)t * B * H * M, attn_hist + (size_t)t * B * H * M, ctx_t, lens, t,
                           att_prev, att_next, xh_n ? xh_n + E : nullptr, Wd, mask_n, EA, d->keep_in, S2, q_t, attn_ws, st));
    } else {
      RC(comic_attn_fwd_ex(&ad, keys, values, part, p->ln_g, p->ln_b, p->v, p->tau, mal, d->keep_alpha,
                           alpha_all + (size_t)t * B * H * M, attn_hist + (size_t)t * B * H * M, ctx_t, nullptr, 0,
                           nullptr, nullptr, nullptr, 0, nullptr, 0, 1.f, S2, q_t, attn_ws, st));
      RC(gemm(ctx_t, p->W_a, att_new, nullptr, B, D, Cv, Cv, D, D, 0, 0, 0.f, st));
      hipLaunchKernelGGL(select_att_kernel, dim3(cdiv(B * A, 256)), dim3(256), 0, st, att_prev, att_new, lens, t,
                         att_next, xh_n ? xh_n + E : nullptr, Wd, mask_n, EA, d->keep_in, B, A);
      COMIC_LAUNCH_CHECK("select_att");
    }
  }
  // output projection for all executed steps, loss, d logits
  if (grp) {
    GemmGroupRun go;
    go.add(COMIC_GG_NN, y_all, wo_g, logits_tb, Tp * B, V, D, D, ldl, V)->bias = p->b_o;
    RC(go.run(gg_slab, gg_slab_cap, gg_tickets, st));
  } else {
    RC(gemm_big(y_all, p->W_o, logits_tb, p->b_o, Tp * B, V, D, D, V, V, 0, 0, 0.f, st));
  }
  }   // do_fwd
  if (!do_bwd) {
    COMIC_LAUNCH_CHECK("train_step (forward phase)");
    return 0;
  }
  if (grp) {       // sequence loss + attention-map loss: one launch (d logits rows are ldl floats apart)
    const int nparts = (int)cdiv64((long)Tp * B * M, 256);
    COMIC_REQUIRE(nparts <= B * Wd, "train_step: map-loss scratch too small");
    RC(comic_xent_maploss(logits_tb, targets_bt, coef_bt, wmask_bt, lens, loss_rows, dlogits, ldl, ids_tb, Tp, T, B, V, attn_hist,
                          dmap, dxh /* free until the backward loop */, map_loss, gg_tickets + kGroupTickets - 1, H, M,
                          d->map_loss_scale, st));
  } else {
    RC(comic_xent_ex(logits_tb, targets_bt, coef_bt, wmask_bt, lens, loss_rows, dlogits, ids_tb, Tp, T, B, V, st));
  }
  for (int t = Tp; t < T; ++t) {  // ops_rnn.py:235-241: pad by copying the last executed step
    (void)hipMemcpyAsync(logits_tb + (size_t)t * B * V, logits_tb + (size_t)(Tp - 1) * B * V, sizeof(float) * B * V,
                         hipMemcpyDeviceToDevice, st);
    (void)hipMemcpyAsync(ids_tb + (size_t)t * B, ids_tb + (size_t)(Tp - 1) * B, sizeof(int32_t) * B,
                         hipMemcpyDeviceToDevice, st);
    RC(fill(loss_rows + (size_t)t * B, 0.f, B, st));
  }
  if (!grp) {
    const long n = (long)Tp * B * M;
    const int nparts = (int)cdiv64(n, 256);
    float* partial = dxh;  // free until the backward loop; needs nparts floats
    COMIC_REQUIRE(nparts <= B * Wd, "train_step: map-loss scratch too small");
    hipLaunchKernelGGL(maploss_part_kernel, dim3(nparts), dim3(256), 0, st, attn_hist, dmap, partial, Tp, B, H, M,
                       d->map_loss_scale);
    hipLaunchKernelGGL(maploss_final_kernel, dim3(1), dim3(256), 0, st, partial, nparts, n, d->map_loss_scale,
                       map_loss);
    COMIC_LAUNCH_CHECK("maploss");
  }

  // ------------------------------------------------------------------ backward -----------
  const bool use_map = d->map_loss_scale > 0.f;
  const bool sep_values = d->fm_projection != 2;
  float* dvalues = sep_values ? dvalues_buf : dkeys;
  if (!persist_b) {
    RC(fill(dkeys, 0.f, (long)B * M * D, st));
    if (sep_values) RC(fill(dvalues_buf, 0.f, (long)B * M * Cv, st));
    RC(fill(dc, 0.f, (long)((datt + (long)B * A) - dc), st));    // dc | dh | datt: consecutive workspace blocks
  }
  // dy_all = dlogits * W_o^T (d W_o, d b_o: after the loop, on the gradient lanes)
  if (grp) {
    GemmGroupRun gy;
    gy.add(COMIC_GG_NT, dlogits, wo_g, dy_all, Tp * B, D, ldl, ldl, ldl, D);     // (K = padded columns: zeros on both sides)
    RC(gy.run(gg_slab, gg_slab_cap, gg_tickets, st));
  } else {
    RC(gemm_big(dlogits, p->W_o, dy_all, nullptr, Tp * B, D, V, V, V, D, 0, 1, 0.f, st));
  }
  if (d->context_layer) RC(fill(gr->W_a, 0.f, (long)Cv * D, st));
  // softmax attention: the backward kernel runs as two workgroups per batch row (half of the memory rows each), whose
  // d q / parameter-gradient contributions are added into zero-filled rows (comic_attn_bwd_ex, pgrad_overwrite 2)
  const int attn_bwd_mode = (d->prob == 0 && split_attn_bwd_enabled()) ? 2 : 1;
  if (persist_b) {
    // M > 64: the loop ADDS its rows' d keys into memory step by step (one writer per address, in step order)
    const bool own_rows = (d->flags & COMIC_DEC_BWD_OWN_ROWS) != 0;
    if (M > 64 || own_rows) RC(fill(dkeys, 0.f, (long)B * M * D, st));
    ComicPersistBwdArgs pb{};
    pb.own_rows = own_rows ? 1 : 0;
    pb.K = p->K; pb.W_q = p->W_q; pb.keys = keys;
    pb.ln_g = p->ln_g; pb.ln_b = p->ln_b; pb.v = p->v; pb.tau = p->tau; pb.lens = lens;
    pb.mask_in = drop_in ? mask_in : nullptr; pb.mask_out = drop_out ? mask_out : nullptr;
    pb.mask_alpha = drop_al ? mask_alpha : nullptr;
    pb.keep_in = d->keep_in; pb.keep_out = d->keep_out; pb.keep_alpha = d->keep_alpha;
    pb.q_all = q_all; pb.alpha_all = alpha_all; pb.gates_all = gates_all; pb.cs = cs; pb.cnew_all = cnew_all;
    pb.dy_all = dy_all; pb.dmap = use_map ? dmap : nullptr;
    pb.dq_part = dq_part; pb.dq_sum = dq_sum; pb.dg_blk = dg_blk; pb.dg_all = dg_all; pb.dstate = dstate;
    pb.dotp = dotp;
    pb.dq_all = dq_all; pb.dc = dc; pb.dh = dh; pb.dkeys = dkeys; pb.pgrad = pgrad4; pb.sync = persist_sync;
    pb.B = B; pb.E = E; pb.M = M; pb.H = H; pb.Tp = Tp; pb.method = d->method;
    const int n_grp = (B + 15) / 16;
    for (int g0 = 0; g0 < n_grp; g0 += 4) {
      pb.grp0 = g0;
      pb.n_groups = std::min(4, n_grp - g0);
      RC(comic_persist_bwd_launch(pb, st));
    }
  } else if (attn_bwd_mode == 2) {
    RC(fill(dq_all, 0.f, (long)Tp * B * D, st));
    RC(fill(pgrad, 0.f, (long)Tp * B * (3 * D + 1), st));
  }
  for (int t = persist_b ? -1 : Tp - 1; t >= 0; --t) {
    const float* ctx_t = ctx_all + (size_t)t * B * Cv;
    float* dq_t = dq_all + (size_t)t * B * D;
    const float* mal = drop_al ? mask_alpha + (size_t)t * B * H * M : nullptr;
    int carry;
    if (!d->context_layer) {
      // attn_bwd masks d(att state) by "live" itself; input_bwd keeps the finished rows' share
      RC(comic_attn_bwd_ex(&ad, keys, values, q_all + (size_t)t * B * D, p->ln_g, p->ln_b, p->v, p->tau,
                           alpha_all + (size_t)t * B * H * M, mal, d->keep_alpha, datt,
                           use_map ? dmap + (size_t)t * B * M : nullptr, dq_t, dkeys, dvalues,
                           pgrad + (size_t)t * B * (3 * D + 1), lens, t, st, attn_bwd_mode, attn_ws,
                           attn_ws ? attn_ws + (size_t)B * H * M : nullptr));
      carry = 1;
    } else {
      hipLaunchKernelGGL(split_live_kernel, dim3(cdiv(B * A, 256)), dim3(256), 0, st, datt, datt_live, lens, t, B, A);
      COMIC_LAUNCH_CHECK("split_live");
      RC(gemm(ctx_t, datt_live, gr->W_a, nullptr, Cv, D, B, Cv, D, D, 1, 0, 1.f, st));
      RC(gemm(datt_live, p->W_a, dctx, nullptr, B, Cv, D, D, D, Cv, 0, 1, 0.f, st));
      RC(comic_attn_bwd_ex(&ad, keys, values, q_all + (size_t)t * B * D, p->ln_g, p->ln_b, p->v, p->tau,
                           alpha_all + (size_t)t * B * H * M, mal, d->keep_alpha, dctx,
                           use_map ? dmap + (size_t)t * B * M : nullptr, dq_t, dkeys, dvalues,
                           pgrad + (size_t)t * B * (3 * D + 1), nullptr, 0, st, attn_bwd_mode, attn_ws,
                           attn_ws ? attn_ws + (size_t)B * H * M : nullptr));
      carry = 0;
    }
    float* dy_t = dy_all + (size_t)t * B * D;
    float* part = (float*)g_splitk_ws;
    int S3 = 1, S4 = 1;
    float* dg_t = dg_all + (size_t)t * B * 4 * D;
    if (fused_q) {
      RC(comic_lstm_grad_fused(dq_t, wq_panel, gates_all + (size_t)t * B * 4 * D, cs + (size_t)t * B * D,
                               cnew_all + (size_t)t * B * D, dy_t, drop_out ? mask_out + (size_t)t * B * D : nullptr,
                               d->keep_out, lens, t, dc, dh, dg_t, B, D, st));
    } else if (cell == COMIC_CELL_LN_LSTM) {
      RC(comic_gemm_f32_partial(dq_t, p->W_q, B, D, D, D, D, 1, part, kSplitKBytes, &S3, st));
      RC(comic_ln_lstm_bwd(gates_all + (size_t)t * B * 4 * D, lnx_all + (size_t)t * B * 5 * D, lnr_all + (size_t)t * B * 8,
                           p->cell_ln, cs + (size_t)t * B * D, cnew_all + (size_t)t * B * D, dy_t, part, S3,
                           drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t, dc, dh, dg_t,
                           lnpg + (size_t)t * B * 10 * D, B, D, st));
    } else if (cell == COMIC_CELL_GRU) {         // dg_t: d r_pre | d u_pre | d candidate_pre | -
      const float* ga = gates_all + (size_t)t * B * 4 * D;
      const float* h_prev = hs + (size_t)t * B * D;
      float* slice1 = gru_dxh + (size_t)B * Wd;
      RC(comic_gemm_f32_partial(dq_t, p->W_q, B, D, D, D, D, 1, part, kSplitKBytes, &S3, st));
      RC(comic_gru_bwd1(dy_t, part, S3, drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t, dh, ga, 4 * D,
                        ga + 2 * D, 4 * D, h_prev, dg_t, 4 * D, B, D, st));
      RC(gemm(dg_t + 2 * D, p->K_c, slice1, nullptr, B, Wd, D, 4 * D, D, Wd, 0, 1, 0.f, st));
      RC(comic_gru_bwd2(slice1, Wd, ga, 4 * D, h_prev, dg_t, 4 * D, B, D, EA, st));
      RC(gemm(dg_t, p->K, gru_dxh, nullptr, B, Wd, 2 * D, 4 * D, 2 * D, Wd, 0, 1, 0.f, st));
      hipLaunchKernelGGL(input_bwd_kernel, dim3(cdiv(B * Wd, 256)), dim3(256), 0, st, gru_dxh,
                         drop_in ? mask_in + (size_t)t * B * EA : nullptr, d->keep_in, demb + (size_t)t * B * E,
                         datt, dh, lens, t, carry, B, E, A, D, 2);
      COMIC_LAUNCH_CHECK("input_bwd");
      continue;
    } else {
      RC(comic_gemm_f32_partial(dq_t, p->W_q, B, D, D, D, D, 1, part, kSplitKBytes, &S3, st));
      RC(comic_lstm_gates_bwd_ex(gates_all + (size_t)t * B * 4 * D, cs + (size_t)t * B * D,
                                 cnew_all + (size_t)t * B * D, dy_t, part, S3,
                                 drop_out ? mask_out + (size_t)t * B * D : nullptr, d->keep_out, lens, t, dc, dh,
                                 dg_t, B, D, st));
    }
    if (fused) {
      RC(comic_input_grad_fused(dg_t, kpanel_b, drop_in ? mask_in + (size_t)t * B * EA : nullptr, d->keep_in,
                                demb + (size_t)t * B * E, datt, dh, lens, t, carry, B, E, A, D, st));
    } else {
      RC(comic_gemm_f32_partial(dg_t, p->K, B, Wd, 4 * D, 4 * D, 4 * D, 1, part, kSplitKBytes, &S4, st));
      hipLaunchKernelGGL(input_bwd_kernel, dim3(cdiv(B * Wd, 256)), dim3(256), 0, st, part,
                         drop_in ? mask_in + (size_t)t * B * EA : nullptr, d->keep_in, demb + (size_t)t * B * E,
                         datt, dh, lens, t, carry, B, E, A, D, S4);
      COMIC_LAUNCH_CHECK("input_bwd");
    }
  }
  // ---- gradients that do not feed the recurrence ---------------------------------------------------------------------
  float* dx_im = nullptr;  // gradient w.r.t. (im_embed * W_init)
  int n_init = 0;
  if (grp) {
    // ONE grouped launch: every weight gradient, the bias sums, the embedding third of d gates * K^T, d x_init
    const bool init_step = d->init_method != 1;
    const int rows_k = (Tp + (init_step ? 1 : 0)) * B;            // rows of the LSTM operand / d gates matrices
    if (init_step)                                               // d gates of the init step -> row block Tp of dg_all
      RC(comic_lstm_gates_bwd(ib.gates, nullptr, ib.c_new, nullptr, nullptr, 1.f, nullptr, 0, dc, dh, dg_init, B, D, (void*)st));
    GemmGroupRun g1, g2;
    g1.add(COMIC_GG_TN, xh_all, dg_all, gr->K, Wd, 4 * D, rows_k, Wd, 4 * D, 4 * D);
    g1.add(COMIC_GG_TN, fm, dkeys, gr->W_m, d->C, D, B * M, d->C, D, D);
    if (d->fm_projection == 1) g1.add(COMIC_GG_TN, fm, dvalues_buf, gr->W_v, d->C, D, B * M, d->C, D, D);
    if (persist_b) {   // the embedding third of d gates * K^T, all steps at once, and its input dropout
      ComicGemmProb* q = g1.add(COMIC_GG_NT, dg_all, p->K, demb, Tp * B, E, 4 * D, 4 * D, 4 * D, E);
      if (drop_in) { q->mask = mask_in; q->ld_mask = EA; q->keep = d->keep_in; }
    }
    g1.add(COMIC_GG_TN, y_all, dq_all, gr->W_q, D, D, Tp * B, D, D, D);
    g1.add(COMIC_GG_TN, y_all, dlogits, gr->W_o, D, V, Tp * B, D, ldl, V);
    if (dfm) g1.add(COMIC_GG_NT, dkeys, p->W_m, dfm, B * M, d->C, D, D, D, d->C);
    g1.add_colsum(dg_all, gr->b, 4 * D, rows_k, 4 * D);
    g1.add_colsum(dlogits, gr->b_o, V, Tp * B, ldl);
    if (d->method == 0) {      // pgrad rows are [v | ln_g | ln_b | tau]: column sums over the rows
      const float* pg = persist_b ? pgrad4 : pgrad;
      const int pr = persist_b ? 4 * B : Tp * B, ldp = 3 * D + 1;
      g1.add_colsum(pg, gr->v, D, pr, ldp);
      g1.add_colsum(pg + D, gr->ln_g, D, pr, ldp);
      g1.add_colsum(pg + 2 * D, gr->ln_b, D, pr, ldp);
      g1.add_colsum(pg + 3 * D, gr->tau, 1, pr, ldp);
    }
    if (init_step) {
      ComicGemmProb* q = g1.add(COMIC_GG_NT, dg_init, p->K, dx_init, B, EA, 4 * D, 4 * D, 4 * D, EA);
      if (drop_in) { q->mask = mask_init_in; q->ld_mask = EA; q->keep = d->keep_in; }
      dx_im = dx_init;
      n_init = EA;
      g2.add(COMIC_GG_TN, im_embed, dx_im, gr->W_init, d->Cg, n_init, B, d->Cg, n_init, n_init);
      if (dim_embed) g2.add(COMIC_GG_NT, dx_im, p->W_init, dim_embed, B, d->Cg, n_init, n_init, n_init, d->Cg);
    } else {
      dx_im = dh;
      n_init = D;
      g1.add(COMIC_GG_TN, im_embed, dx_im, gr->W_init, d->Cg, n_init, B, d->Cg, n_init, n_init);
      if (dim_embed) g1.add(COMIC_GG_NT, dx_im, p->W_init, dim_embed, B, d->Cg, n_init, n_init, n_init, d->Cg);
    }
    if (d->fm_projection == 1 && dfm) {
      ComicGemmProb* q = g2.add(COMIC_GG_NT, dvalues_buf, p->W_v, dfm, B * M, d->C, D, D, D, d->C);
      q->beta = 1.f;
    }
    RC(g1.run(gg_slab, gg_slab_cap, gg_tickets, st));
    RC(comic_embed_bwd_set(in_tb, demb, gr->emb, Tp * B, E, V, st));
    RC(g2.run(gg_slab, gg_slab_cap, gg_tickets, st));
    if (d->fm_projection == 0 && dfm) RC(comic_axpy(dfm, dvalues_buf, 1.f, (int64_t)B * M * Cv, (void*)st));
  } else {
  // ---- gradients that do not feed the recurrence, on two lanes (side_lane) ------------------------------------------
  LaneScope glane(L, st, splitk_ws_b);
  RC(glane.rc);
  hipStream_t sb = glane.lane();
  // lane B: output projection, embedding, query layer, memory projections, attention parameters
  RC(gemm_big(y_all, dlogits, gr->W_o, nullptr, D, V, Tp * B, D, V, V, 1, 0, 0.f, sb));
  if (persist_b) {   // the embedding third of d gates * K^T, all steps at once, and its input dropout
    RC(gemm_big(dg_all, p->K, demb, nullptr, Tp * B, E, 4 * D, 4 * D, 4 * D, E, 0, 1, 0.f, sb));
    if (drop_in) RC(comic_dropout_rows(demb, mask_in, d->keep_in, (long)Tp * B, E, EA, sb));
  }
  RC(fill(gr->emb, 0.f, (long)V * E, sb));
  RC(comic_embed_bwd(in_tb, demb, gr->emb, Tp * B, E, V, (void*)sb));
  RC(gemm_big(y_all, dq_all, gr->W_q, nullptr, D, D, Tp * B, D, D, D, 1, 0, 0.f, sb));
  RC(gemm_big(fm, dkeys, gr->W_m, nullptr, d->C, D, B * M, d->C, D, D, 1, 0, 0.f, sb));
  if (dfm) RC(gemm_big(dkeys, p->W_m, dfm, nullptr, B * M, d->C, D, D, D, d->C, 0, 1, 0.f, sb));
  if (d->fm_projection == 1) {
    RC(gemm_big(fm, dvalues_buf, gr->W_v, nullptr, d->C, D, B * M, d->C, D, D, 1, 0, 0.f, sb));
    if (dfm) RC(gemm_big(dvalues_buf, p->W_v, dfm, nullptr, B * M, d->C, D, D, D, d->C, 0, 1, 1.f, sb));
  } else if (d->fm_projection == 0 && dfm) {
    RC(comic_axpy(dfm, dvalues_buf, 1.f, (int64_t)B * M * Cv, (void*)sb));
  }
  if (d->method == 0) {
    // pgrad rows are [v | ln_g | ln_b | tau]: column sums over the batch, then scatter (g_tmp [B][4D] is free after
    // the loops and holds 3D + 1 floats at any batch size)
    float* tmp = g_tmp;
    if (persist_b) RC(comic_colsum_ws(pgrad4, tmp, 4 * B, 3 * D + 1, 0.f, (float*)g_splitk_ws, sb));
    else RC(comic_colsum_ws(pgrad, tmp, Tp * B, 3 * D + 1, 0.f, (float*)g_splitk_ws, sb));
    hipLaunchKernelGGL(scatter_pgrad_kernel, dim3(cdiv(D, 256)), dim3(256), 0, sb, tmp, gr->v, gr->ln_g, gr->ln_b, gr->tau, D);
  }
  glane.main_ws();
  // lane A: output bias, LSTM kernel and bias, then the rnn init (which accumulates into both)
  RC(comic_colsum_ws(dlogits, gr->b_o, Tp * B, V, 0.f, (float*)g_splitk_ws, st));
  if (cell == COMIC_CELL_GRU) {
    RC(gemm_big(xh_all, dg_all, gr->K, nullptr, Wd, 2 * D, Tp * B, Wd, 4 * D, 2 * D, 1, 0, 0.f, st));
    RC(gemm_big(xh2_all, dg_all + 2 * D, gr->K_c, nullptr, Wd, D, Tp * B, Wd, 4 * D, D, 1, 0, 0.f, st));
    RC(comic_colsum_ws(dg_all, cell_tmp, Tp * B, 4 * D, 0.f, (float*)g_splitk_ws, st));
    COMIC_REQUIRE(hipMemcpyAsync(gr->b, cell_tmp, sizeof(float) * 2 * D, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                      hipMemcpyAsync(gr->b_c, cell_tmp + 2 * D, sizeof(float) * D, hipMemcpyDeviceToDevice, st) == hipSuccess,
                  "train_step: bias gradient copy");
  } else {
    RC(gemm_big(xh_all, dg_all, gr->K, nullptr, Wd, 4 * D, Tp * B, Wd, 4 * D, 4 * D, 1, 0, 0.f, st));
    if (cell == COMIC_CELL_LSTM) RC(comic_colsum_ws(dg_all, gr->b, Tp * B, 4 * D, 0.f, (float*)g_splitk_ws, st));
  }
  if (d->init_method == 1) {
    dx_im = dh;
    n_init = D;
  } else if (cell == COMIC_CELL_LN_LSTM) {
    RC(comic_ln_lstm_bwd(ib.gates, ib.lnx, ib.lnr, p->cell_ln, nullptr, ib.c_new, nullptr, nullptr, 0, nullptr, 1.f, nullptr, 0,
                         dc, dh, ib.g, lnpg + (size_t)Tp * B * 10 * D, B, D, st));
    RC(gemm(ib.xh, ib.g, gr->K, nullptr, EA, 4 * D, B, EA, 4 * D, 4 * D, 1, 0, 1.f, st));
    RC(gemm(ib.g, p->K, dx_init, nullptr, B, EA, 4 * D, 4 * D, 4 * D, EA, 0, 1, 0.f, st));
    if (drop_in) RC(comic_dropout_apply(dx_init, mask_init_in, d->keep_in, dx_init, (int64_t)B * EA, (void*)st));
    dx_im = dx_init;
    n_init = EA;
  } else if (cell == COMIC_CELL_GRU) {        // zero state: d r_pre = 0
    RC(fill(ib.g, 0.f, (long)B * 4 * D, st));
    RC(comic_gru_bwd1(nullptr, nullptr, 0, nullptr, 1.f, nullptr, 0, dh, ib.gates, 4 * D, ib.gates + 2 * D, 4 * D, nullptr,
                      ib.g, 4 * D, B, D, st));
    RC(gemm(ib.xh, ib.g, gr->K, nullptr, EA, 2 * D, B, EA, 4 * D, 2 * D, 1, 0, 1.f, st));
    RC(gemm(ib.xh, ib.g + 2 * D, gr->K_c, nullptr, EA, D, B, EA, 4 * D, D, 1, 0, 1.f, st));
    RC(comic_colsum(ib.g, cell_tmp, B, 4 * D, 0.f, (void*)st));
    RC(comic_axpy(gr->b, cell_tmp, 1.f, 2 * D, (void*)st));
    RC(comic_axpy(gr->b_c, cell_tmp + 2 * D, 1.f, D, (void*)st));
    RC(gemm(ib.g, p->K, dx_init, nullptr, B, EA, 2 * D, 4 * D, 2 * D, EA, 0, 1, 0.f, st));
    RC(gemm(ib.g + 2 * D, p->K_c, dx_init, nullptr, B, EA, D, 4 * D, D, EA, 0, 1, 1.f, st));
    if (drop_in) RC(comic_dropout_apply(dx_init, mask_init_in, d->keep_in, dx_init, (int64_t)B * EA, (void*)st));
    dx_im = dx_init;
    n_init = EA;
  } else {
    RC(comic_lstm_gates_bwd(ib.gates, nullptr, ib.c_new, nullptr, nullptr, 1.f, nullptr, 0, dc, dh, ib.g, B, D,
                            (void*)st));
    RC(gemm(ib.xh, ib.g, gr->K, nullptr, EA, 4 * D, B, EA, 4 * D, 4 * D, 1, 0, 1.f, st));
    RC(comic_colsum(ib.g, gr->b, B, 4 * D, 1.f, (void*)st));
    RC(gemm(ib.g, p->K, dx_init, nullptr, B, EA, 4 * D, 4 * D, 4 * D, EA, 0, 1, 0.f, st));
    if (drop_in) RC(comic_dropout_apply(dx_init, mask_init_in, d->keep_in, dx_init, (int64_t)B * EA, (void*)st));
    dx_im = dx_init;
    n_init = EA;
  }
  if (cell == COMIC_CELL_LN_LSTM) {           // LayerNorm gains / shifts: column sums of the per-row gradient rows
    const int rows = Tp * B + (d->init_method == 1 ? 0 : B);      // (the init step wrote its rows behind step Tp - 1's)
    RC(comic_colsum_ws(lnpg, cell_tmp, rows, 10 * D, 0.f, (float*)g_splitk_ws, st));
    RC(comic_ln_lstm_scatter(cell_tmp, gr->cell_ln, D, 0.f, st));
  }
  RC(gemm_big(im_embed, dx_im, gr->W_init, nullptr, d->Cg, n_init, B, d->Cg, n_init, n_init, 1, 0, 0.f, st));
  if (dim_embed) RC(gemm(dx_im, p->W_init, dim_embed, nullptr, B, d->Cg, n_init, n_init, n_init, d->Cg, 0, 1, 0.f, st));
  RC(glane.join());
  }
  if (persist) {
    // a persistent loop that timed out leaves garbage everywhere: NaN losses and zero gradients (no host check needed
    // for the optimiser step that follows to be harmless; the host raises at its next look at the loss)
    ComicGateRanges gr_{};
    int k = 0;
    auto add = [&](float* ptr, long n) {
      if (ptr && n > 0 && k < 16) { gr_.p[k] = ptr; gr_.n[k] = n; ++k; }
    };
    add(gr->W_init, (long)d->Cg * n_init); add(gr->K, (long)Wd * 4 * D); add(gr->b, 4L * D);      // (LSTM only: persist)
    add(gr->W_m, (long)d->C * D);
    if (d->fm_projection == 1) add(gr->W_v, (long)d->C * D);
    add(gr->W_q, (long)D * D);
    if (d->method == 0) { add(gr->v, D); add(gr->ln_g, D); add(gr->ln_b, D); add(gr->tau, 1); }
    if (d->context_layer) add(gr->W_a, (long)Cv * D);
    add(gr->W_o, (long)D * V); add(gr->b_o, V); add(gr->emb, (long)V * E);
    add(dfm, (long)B * M * d->C); add(dim_embed, (long)B * d->Cg);
    if (d->flags & COMIC_DEC_INJECT_TIMEOUT)     // fault injection of THIS call: raise the loops' error word by hand
      COMIC_REQUIRE(hipMemsetAsync(persist_sync, 0xFF, sizeof(unsigned), st) == hipSuccess, "train_step: memset");
    RC(comic_persist_gate(persist_sync, loss_rows, map_loss, gr_, gr->status, p->status, st));
  }
  COMIC_LAUNCH_CHECK("train_step");
  return 0;
}

// columns of the W_o scratch: V rounded up to whole chunks of any of its packed forms (128, 112 or 64 columns)
static inline long wo_pad_cols(long V) { return (V + 127) / 128 * 128 + 128; }
constexpr int kBeamCntSteps = 4096;       // decode steps the in-kernel completion counters of the beam step cover

extern "C" int64_t comic_decoder_infer_workspace(const comic_decoder_desc* d, int rows, int max_steps) {
  if (!d) return -1;
  Bump w(nullptr, 0);
  const long D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H, Cv = d->Cv, Wd = E + A + D, R = rows;
  w.take<float>(R * M * d->C); w.take<float>(R * d->Cg);           // tiled fm, im_embed
  w.take<float>(R * M * D); w.take<float>(R * M * D);              // keys, values
  w.take<float>(R * (E + A)); w.take<float>(R * (E + A)); w.take<float>(R * 4 * D); w.take<float>(R * 4 * D);
  w.take<float>(R * D);                                            // init bufs
  w.take<float>(R * Wd); w.take<float>(R * 4 * D); w.take<float>(R * D); w.take<float>(R * D);  // xh,g,y,q
  w.take<float>(R * H * M); w.take<float>(R * Cv);                 // alpha, ctx
  for (int i = 0; i < 4; ++i) w.take<float>(R * D);                // c,h ping-pong
  w.take<float>(R * D);                                            // att2 (context layer)
  for (int i = 0; i < 2; ++i) w.take<float>(R * A);                // att ping-pong
  w.take<float>(R * E); w.take<float>(R * V);                      // x, logits
  w.take<int32_t>(R); w.take<float>(R); w.take<int32_t>(R);        // ids, log_probs, parents
  w.take<float>(R * (2 * D + A));                                  // gather temp
  w.take<char>(kSplitKBytes);                                      // split-K partials
  w.take<float>(comic_lstm_panel_floats(d->D, d->E + d->A + d->D, 0));  // LSTM kernel panel (fused step)
  w.take<float>((D + 1) * wo_pad_cols(V));                         // W_o with 16-byte aligned rows / packed hi-lo fragments + bias
  w.take<float>(comic_lstm_stream_kfrag_floats(d->D, (int)Wd)); w.take<float>(comic_lstm_stream_xfrag_floats(rows, (int)Wd));  // streaming LSTM step
  w.take<float>(comic_stream_gemm_wfrag_floats(d->D, d->D)); w.take<float>(comic_lstm_stream_xfrag_floats(rows, d->D));     // ... W_q, y fragments
  w.take<unsigned long long>(kBeamCntSteps);                       // beam search: per-step completion counters
  if (d->cell == COMIC_CELL_GRU) w.take<float>(R * Wd);            // GRU: [x ; att ; r*h]
  if (rows <= 64) {                                                // persistent greedy loop: hand-off buffers of all steps
    const long S = std::max(1, max_steps);
    w.take<float>(S * R * Wd); w.take<float>(S * R * D); w.take<float>(S * R * D); w.take<float>(S * R * 132);
    w.take<unsigned>(kPersistSyncWords);
  }
  return (int64_t)w.off;
}

namespace {
struct InferBufs {
  float *fm_t, *im_t, *keys, *values_buf;
  InitBufs ib;
  StepBufs sb;
  float *c[2], *h[2], *att[2], *x, *logits, *log_probs, *gtmp, *kpanel, *wo_pad, *kfrag, *xfrag, *wqfrag, *yfrag;
  unsigned long long* beam_cnt = nullptr;
  int32_t *ids, *parents;
  float *p_xh = nullptr, *p_y = nullptr, *p_q = nullptr, *p_argp = nullptr;   // persistent greedy loop (rows <= 64)
  unsigned* p_sync = nullptr;
  bool ok;
};
InferBufs carve_infer(const comic_decoder_desc* d, int rows, void* ws, int64_t bytes, int max_steps = 0) {
  Bump w(ws, (size_t)bytes);
  const long D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H, Cv = d->Cv, Wd = E + A + D, R = rows;
  InferBufs b;
  b.fm_t = w.take<float>(R * M * d->C); b.im_t = w.take<float>(R * d->Cg);
  b.keys = w.take<float>(R * M * D); b.values_buf = w.take<float>(R * M * D);
  b.ib.x = w.take<float>(R * (E + A)); b.ib.xh = w.take<float>(R * (E + A));
  b.ib.g = w.take<float>(R * 4 * D); b.ib.gates = w.take<float>(R * 4 * D); b.ib.c_new = w.take<float>(R * D);
  b.sb.xh = w.take<float>(R * Wd); b.sb.g = w.take<float>(R * 4 * D); b.sb.y = w.take<float>(R * D);
  b.sb.q = w.take<float>(R * D); b.sb.alpha = w.take<float>(R * H * M); b.sb.ctx = w.take<float>(R * Cv);
  b.c[0] = w.take<float>(R * D); b.c[1] = w.take<float>(R * D);
  b.h[0] = w.take<float>(R * D); b.h[1] = w.take<float>(R * D);
  b.sb.att2 = w.take<float>(R * D);
  b.att[0] = w.take<float>(R * A); b.att[1] = w.take<float>(R * A);
  b.x = w.take<float>(R * E); b.logits = w.take<float>(R * V);
  b.ids = w.take<int32_t>(R); b.log_probs = w.take<float>(R); b.parents = w.take<int32_t>(R);
  b.gtmp = w.take<float>(R * (2 * D + A));
  g_splitk_ws = w.take<char>(kSplitKBytes);
  b.kpanel = w.take<float>(comic_lstm_panel_floats(d->D, d->E + d->A + d->D, 0));
  b.wo_pad = w.take<float>((D + 1) * wo_pad_cols(V));
  b.kfrag = w.take<float>(comic_lstm_stream_kfrag_floats(d->D, (int)Wd)); b.xfrag = w.take<float>(comic_lstm_stream_xfrag_floats(rows, (int)Wd));
  b.wqfrag = w.take<float>(comic_stream_gemm_wfrag_floats(d->D, d->D)); b.yfrag = w.take<float>(comic_lstm_stream_xfrag_floats(rows, d->D));
  b.beam_cnt = w.take<unsigned long long>(kBeamCntSteps);
  if (d->cell == COMIC_CELL_GRU) b.sb.xh2 = w.take<float>(R * Wd);
  if (rows <= 64 && max_steps > 0) {
    const long S = max_steps;
    b.p_xh = w.take<float>(S * R * Wd); b.p_y = w.take<float>(S * R * D); b.p_q = w.take<float>(S * R * D);
    b.p_argp = w.take<float>(S * R * 132);
    b.p_sync = w.take<unsigned>(kPersistSyncWords);
  }
  b.ok = w.ok;
  return b;
}

// The output projection is streamed once per decode step.  Its rows ([D][V], TensorFlow layout) are 16-byte aligned
// only when V % 4 == 0 (V = 25 599 for the word vocabulary, 258 for radix-256): a copy with padded rows lets the
// product kernels use 16-byte loads (4x fewer load instructions on the dominant operand).  Returns the matrix and
// its leading dimension to use for this call.
const float* aligned_w_o(const comic_decoder_desc* d, const comic_decoder_params* p, float* pad, int* ldb,
                         hipStream_t st) {
  const int V = d->V, Vp = (V + 3) / 4 * 4;
  if (V == Vp || !pad) {
    *ldb = V;
    return p->W_o;
  }
  const long n = (long)d->D * Vp;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p->W_o, pad, V, Vp, n);
  *ldb = Vp;
  return pad;
}
}  // namespace

int comic_argmax_rows_noise(const float* x, const float* noise, int32_t* idx, int rows, int V, hipStream_t st);

// greedy (gumbel_tb null) or sampled (gumbel_tb [max_steps][B][V]: ids = argmax(logits + noise)) decode loop
static int decoder_search(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                          const float* im_embed, int B, int max_steps, const float* gumbel_tb, int32_t* ids_tb,
                          float* logits_tb, float* attn_hist, int32_t* first_eos, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  RC(check_desc(d));
  FlagScope flag_scope__(d);
  COMIC_REQUIRE(p && fm && im_embed && ids_tb && attn_hist && first_eos && workspace, "greedy: null pointer");
  COMIC_REQUIRE(B > 0 && max_steps > 0, "greedy: bad shape");
  COMIC_REQUIRE(workspace_bytes >= comic_decoder_infer_workspace(d, B, max_steps), "greedy: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  InferBufs ws = carve_infer(d, B, workspace, workspace_bytes, max_steps);
  COMIC_REQUIRE(ws.ok, "greedy: workspace overflow");
  const int D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H, Cv = d->Cv;
  const comic_attn_desc ad = attn_desc(d, B);
  const float* values = nullptr;
  RC(memory_projections(d, p, fm, B, ws.keys, ws.values_buf, &values, st));
  RC(rnn_init_fwd(d, p, im_embed, B, nullptr, ws.ib, ws.c[0], ws.h[0], st));
  RC(fill(ws.att[0], 0.f, (long)B * A, st));
  hipLaunchKernelGGL(fill_i32_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, ws.ids, d->start_id, (long)B);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, first_eos, max_steps, (long)B);
  int32_t* steps_done = ws.parents;        // beam-only buffer: its first word is this loop's end marker
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, steps_done, max_steps, 1L);
  COMIC_LAUNCH_CHECK("greedy init");
  const bool fused = fused_step_enabled() && comic_fused_step_supported(D, E + A + D);
  if (fused) RC(comic_pack_lstm_panels(p->K, ws.kpanel, nullptr, D, E + A + D, st));
  // the whole loop as one persistent launch (decoder_persist.hip, GREEDY) when the shape allows it
  g_greedy_path = 0;
  if (!gumbel_tb && fused && persist_enabled() && ws.p_xh &&
      comic_persist_greedy_supported(B, D, E, A, M, H, Cv, V, d->method, d->context_layer, ad.tied) &&
      comic_persist_fits_device(B)) {
    const int Wd = E + A + D;
    ComicPersistRanges pr{};
    pr.p[0] = ws.p_xh; pr.n[0] = (long)max_steps * B * Wd;
    pr.p[1] = ws.p_y; pr.n[1] = (long)max_steps * B * D;
    pr.p[2] = ws.p_q; pr.n[2] = (long)max_steps * B * D;
    pr.p[3] = ws.p_argp; pr.n[3] = (long)max_steps * B * 132;
    RC(comic_persist_prepare(pr, ws.p_sync, kPersistSyncWords, st));
    // step 0 operand: att = 0, h = h0 (the x third comes from the embedding table inside the loop)
    {
      const long n = (long)B * A + (long)B * D;
      hipLaunchKernelGGL(embed_step0_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p->emb,
                         (const int32_t*)nullptr, (int32_t*)nullptr, (const float*)nullptr, 1.f, ws.p_xh, ws.att[0], ws.h[0],
                         0, B, 0, E, A, D, V, (float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                         (float*)nullptr, (float*)nullptr);
      COMIC_LAUNCH_CHECK("greedy step0");
    }
    ComicPersistFwdArgs pa{};
    pa.K_panel = ws.kpanel; pa.bias = p->b; pa.W_q = p->W_q; pa.keys = ws.keys; pa.values = values;
    pa.ln_g = p->ln_g; pa.ln_b = p->ln_b; pa.v = p->v; pa.tau = p->tau;
    pa.keep_in = pa.keep_out = pa.keep_alpha = 1.f;
    pa.xh_all = ws.p_xh; pa.y_all = ws.p_y; pa.q_all = ws.p_q; pa.cs = ws.c[0]; pa.hs = ws.h[0];
    pa.attn_hist = attn_hist; pa.sync = ws.p_sync;
    pa.B = B; pa.D = D; pa.E = E; pa.Wd = Wd; pa.M = M; pa.H = H; pa.Tp = max_steps;
    pa.method = d->method; pa.prob = d->prob; pa.tied = ad.tied;
    pa.grp0 = 0; pa.n_groups = (B + 15) / 16;
    pa.greedy = 1; pa.emb = p->emb; pa.W_o = p->W_o; pa.b_o = p->b_o; pa.V = V; pa.ld_wo = V;
    pa.start_id = d->start_id; pa.end_id = d->end_id;
    pa.argp = ws.p_argp; pa.ids_tb = ids_tb; pa.first_eos = first_eos; pa.logits_tb = logits_tb;
    RC(comic_persist_fwd_launch(pa, st));
    RC(comic_persist_check_greedy(ws.p_sync, first_eos, st));
    g_greedy_path = 1;
    return 0;
  }
  int ld_wo = V;
  const float* w_o = aligned_w_o(d, p, ws.wo_pad, &ld_wo, st);
  struct StopScope {          // whatever way this call returns, no later launch sees the flag
    ~StopScope() { g_comic_stop = ComicStop(); }
  } stop_scope;
  for (int t = 0; t < max_steps; ++t) {
    if (fused) {              // the step's kernels return at once when the loop ended earlier (fused path only)
      g_comic_stop.p = steps_done;
      g_comic_stop.t = t;
    }
    const int cur = t & 1, nxt = cur ^ 1;
    ws.sb.c2 = ws.c[nxt];
    ws.sb.h2 = ws.h[nxt];
    float* att_next = ws.att[nxt];
    StepBufs sb = ws.sb;
    if (!d->context_layer) sb.ctx = att_next;  // context written straight into the next attention state
    else sb.att2 = att_next;
    // the ids of step t-1 are read where argmax wrote them (ids_tb), no copy
    const int32_t* ids_in = t == 0 ? ws.ids : ids_tb + (size_t)(t - 1) * B;
    if (fused) {
      RC(infer_step_fused(d, p, ad, ws.keys, values, ws.kpanel, ids_in, nullptr, 1, ws.c[cur], ws.h[cur], ws.att[cur],
                          sb, ws.gtmp, attn_hist + (size_t)t * B * H * M, B, st));
    } else {
      RC(comic_embed_fwd(p->emb, ids_in, ws.x, B, E, V, (void*)st));
      RC(infer_step(d, p, ad, ws.keys, values, ws.x, ws.c[cur], ws.h[cur], ws.att[cur], sb,
                    attn_hist + (size_t)t * B * H * M, B, st));
    }
    float* lg = logits_tb ? logits_tb + (size_t)t * B * V : ws.logits;
    RC(gemm(sb.y, w_o, lg, p->b_o, B, V, D, D, ld_wo, V, 0, 0, 0.f, st));
    int32_t* ids_out = ids_tb + (size_t)t * B;
    RC(comic_argmax_rows_noise(lg, gumbel_tb ? gumbel_tb + (size_t)t * B * V : nullptr, ids_out, B, V, st));
    hipLaunchKernelGGL(eos_track_done_kernel, dim3(1), dim3(256), 0, st, (const int32_t*)ids_out, first_eos, t,
                       d->end_id, B, steps_done, max_steps);
    COMIC_LAUNCH_CHECK("eos_track");
  }
  (void)Cv; (void)M;
  return 0;
}

extern "C" int comic_decoder_greedy(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                                    const float* im_embed, int B, int max_steps, int32_t* ids_tb, float* logits_tb,
                                    float* attn_hist, int32_t* first_eos, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
  return decoder_search(d, p, fm, im_embed, B, max_steps, nullptr, ids_tb, logits_tb, attn_hist, first_eos, workspace,
                        workspace_bytes, stream);
}
extern "C" int comic_decoder_sample(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                                    const float* im_embed, int B, int max_steps, const float* gumbel_tb, int32_t* ids_tb,
                                    float* logits_tb, float* attn_hist, int32_t* first_eos, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  COMIC_REQUIRE(gumbel_tb, "sample: null noise");
  return decoder_search(d, p, fm, im_embed, B, max_steps, gumbel_tb, ids_tb, logits_tb, attn_hist, first_eos, workspace,
                        workspace_bytes, stream);
}

// ---- one beam step from the decoder outputs, as an operator (C-ABI) -----------------------------------------------------
// logits = y W_o + b_o, log_softmax, _mask_probs, top-k over beam x V, bookkeeping ([TF-1.9] _beam_search_step as used by
// rnn_decoder_beam_search, ops_rnn.py:49-112) with the kernels comic_decoder_beam picks for the shape: the streaming
// projection + per-chunk top-k + merge (V >= 4096, D % 128 == 0), the register-resident small step (V <= 1024), the GEMM +
// comic_beam_step chain otherwise.  The same state arrays as comic_beam_step.
static int64_t beam_dense_region(int B, int W, int D, int V) {        // floats: logits / partials
  const int64_t R = (int64_t)B * W;
  return std::max<int64_t>(R * V, comic_beam_logits_supported(D, V, (int)R, W) ? comic_beam_logits_partial_floats(D, V, (int)R, W, 1) : 0) + 64;
}
extern "C" int64_t comic_beam_step_dense_workspace(int B, int W, int D, int V) {
  if (B <= 0 || W <= 0 || D <= 0 || V <= 0) return -1;
  return ((int64_t)(D + 1) * wo_pad_cols(V) + beam_dense_region(B, W, D, V)) * 4 + kSplitKBytes + 4096;
}
extern "C" int comic_beam_step_dense(const float* y, const float* W_o, const float* b_o, float* log_probs, int32_t* finished,
                                     int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores, int B, int W,
                                     int D, int V, int end_id, void* workspace, int64_t workspace_bytes, void* stream) {
  COMIC_REQUIRE(y && W_o && b_o && log_probs && finished && lengths && word_ids && parent_ids && scores && workspace,
                "beam_step_dense: null pointer");
  COMIC_REQUIRE(B > 0 && W > 0 && W <= 64 && D > 0 && V >= W, "beam_step_dense: bad shape");
  COMIC_REQUIRE(workspace_bytes >= comic_beam_step_dense_workspace(B, W, D, V), "beam_step_dense: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Bump w(workspace, (size_t)workspace_bytes);
  float* wo = w.take<float>((D + 1) * wo_pad_cols(V));
  float* region = w.take<float>(beam_dense_region(B, W, D, V));
  void* splitk = w.take<char>(kSplitKBytes);
  int32_t* steps = w.take<int32_t>(16);
  unsigned long long* cnt = (unsigned long long*)w.take<int64_t>(16);
  COMIC_REQUIRE(w.ok, "beam_step_dense: workspace overflow");
  const int R = B * W;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, steps, 1, 1L);
  if (comic_beam_logits_supported(D, V, R, W)) {
    RC(comic_beam_pack_wo(W_o, b_o, V, wo, D, V, st));
    RC(comic_beam_logits_begin(region, B, W, V, 1, st));
    return comic_beam_logits_step(y, nullptr, wo, region, log_probs, finished, lengths, word_ids, parent_ids, scores, steps, 0, 1,
                                  B, W, D, V, end_id, nullptr, st);
  }
  RC(comic_gemm_bf16x3_impl(y, W_o, region, b_o, R, V, D, D, V, V, 0, 0, 1.f, 0.f, splitk, kSplitKBytes, st));
  if (comic_beam_step_small_supported(V, W)) {
    RC(comic_beam_counters_zero(cnt, 1, st));
    return comic_beam_step_small(region, nullptr, 1, V, 0, log_probs, finished, lengths, word_ids, parent_ids, scores, B, W, V,
                                 end_id, cnt, steps, 0, 1, nullptr, st);
  }
  return comic_beam_step_ws(region, log_probs, finished, lengths, word_ids, parent_ids, scores, B, W, V, end_id, splitk,
                            kSplitKBytes, st);
}

extern "C" int comic_decoder_beam(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                                  const float* im_embed, int B, int W, int max_steps, int32_t* step_ids,
                                  int32_t* parent_ids, float* scores, int64_t* lengths, int32_t* finished,
                                  float* attn_hist, int32_t* steps_executed, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  RC(check_desc(d));
  FlagScope flag_scope__(d);
  COMIC_REQUIRE(p && fm && im_embed && step_ids && parent_ids && scores && lengths && finished && attn_hist &&
                    steps_executed && workspace,
                "beam: null pointer");
  COMIC_REQUIRE(B > 0 && W > 0 && W <= 64 && max_steps > 0, "beam: bad shape");
  const int R = B * W;
  COMIC_REQUIRE(workspace_bytes >= comic_decoder_infer_workspace(d, R, max_steps), "beam: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  InferBufs ws = carve_infer(d, R, workspace, workspace_bytes);
  COMIC_REQUIRE(ws.ok, "beam: workspace overflow");
  const int D = d->D, E = d->E, A = d->A, V = d->V, M = d->M, H = d->H;
  // tile_batch BEFORE keys are computed (model_base.py:127-131).  The beams of an entry attend to the same memory: the
  // fused step's attention kernel reads row b / W of keys / values held ONCE per entry (same values as the tiled copy's)
  const bool fused = fused_step_enabled() && comic_fused_step_supported(D, E + A + D);
  const int mem_div = fused ? W : 1;
  {
    const long n1 = (long)R * M * d->C, n2 = (long)R * d->Cg;
    if (mem_div == 1)
      hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)cdiv64(n1, 256)), dim3(256), 0, st, fm, ws.fm_t, n1, W,
                         M * d->C);
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)cdiv64(n2, 256)), dim3(256), 0, st, im_embed, ws.im_t, n2, W,
                       d->Cg);
    COMIC_LAUNCH_CHECK("tile_rows");
  }
  const comic_attn_desc ad = attn_desc(d, R);
  const float* values = nullptr;
  if (mem_div == 1) RC(memory_projections(d, p, ws.fm_t, R, ws.keys, ws.values_buf, &values, st));
  else RC(memory_projections(d, p, fm, B, ws.keys, ws.values_buf, &values, st));
  RC(rnn_init_fwd(d, p, ws.im_t, R, nullptr, ws.ib, ws.c[0], ws.h[0], st));
  RC(fill(ws.att[0], 0.f, (long)R * A, st));
  // initial beam state: log_probs [0,-inf,...], finished [0,1,...], lengths 0
  hipLaunchKernelGGL(beam_init_kernel, dim3(cdiv(R, 256)), dim3(256), 0, st, ws.log_probs, finished, lengths, R, W);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(cdiv(R, 256)), dim3(256), 0, st, ws.ids, d->start_id, (long)R);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(1), dim3(64), 0, st, steps_executed, max_steps, 1L);
  COMIC_LAUNCH_CHECK("beam init");
  const bool stream_lstm = fused && lstm_stream_enabled() && comic_lstm_stream_supported(D, E, A, R) &&
                           comic_lstm_stream_part_bytes(D, E + A + D, R) <= kSplitKBytes;
  StreamBufs sm{ws.kfrag, ws.xfrag, ws.yfrag, nullptr};
  if (stream_lstm) {
    RC(comic_lstm_stream_pack(p->K, ws.kfrag, D, E + A + D, st));
    if (comic_stream_gemm_supported(D, D, R) && comic_stream_gemm_part_bytes(D, D, R) <= kSplitKBytes) {
      RC(comic_stream_gemm_pack(p->W_q, ws.wqfrag, D, D, st));
      sm.wqfrag = ws.wqfrag;
    }
  } else if (fused) RC(comic_pack_lstm_panels(p->K, ws.kpanel, nullptr, D, E + A + D, st));
  // large vocabularies: projection + per-chunk top-k as one streaming launch over a packed W_o (beam_logits.hip)
  // (a length penalty ranks by score, not by log probability: its step runs the one-workgroup-per-entry kernel)
  const float lpw = d->length_penalty_weight;
  const bool stream_logits = fused && lpw == 0.f && beam_logits_enabled() && comic_beam_logits_supported(D, V, R, W) &&
                             comic_beam_logits_pack_bytes(D, V) <= (int64_t)(D + 1) * wo_pad_cols(V) * 4 &&
                             comic_beam_logits_partial_floats(D, V, R, W, max_steps) <= (int64_t)R * V;
  // small vocabularies (radix-256): the entry's beam step with a beam's logits in a wave's registers; with the step's y at
  // hand as fragments (streaming LSTM step) the projection goes through the streaming kernel too
  const bool small_step = fused && lpw == 0.f && !stream_logits && beam_logits_enabled() && comic_beam_step_small_supported(V, W) &&
                          max_steps <= kBeamCntSteps;
  const bool stream_wo = small_step && stream_lstm && comic_stream_gemm_supported(D, V, R) &&
                         comic_stream_gemm_part_bytes(D, V, R) <= kSplitKBytes &&
                         comic_stream_gemm_wfrag_floats(D, V) <= (int64_t)(D + 1) * wo_pad_cols(V);
  int ld_wo = V;
  const float* w_o = nullptr;
  if (stream_logits) {
    RC(comic_beam_pack_wo(p->W_o, p->b_o, V, ws.wo_pad, D, V, st));
    RC(comic_beam_logits_begin(ws.logits, B, W, V, max_steps, st));
  } else if (stream_wo) {
    RC(comic_stream_gemm_pack(p->W_o, ws.wo_pad, D, V, st));
    // (not with the context layer: its product after the attention may use the whole split-K scratch)
    if (sm.wqfrag && !d->context_layer && comic_stream_gemm_part_bytes(D, D, R) <= kSplitKBytes / 2 &&
        comic_stream_gemm_part_bytes(D, V, R) <= kSplitKBytes / 2) {
      sm.wofrag = ws.wo_pad;       // one launch for the query and the vocabulary projection
      sm.wo_N = V;
    }
  } else {
    w_o = aligned_w_o(d, p, ws.wo_pad, &ld_wo, st);
  }
  if (small_step) RC(comic_beam_counters_zero(ws.beam_cnt, max_steps, st));
  g_beam_path = (stream_logits ? 1 : 0) | (stream_lstm ? 2 : 0) | (small_step ? 4 : 0);
  int cur = 0;
  struct StopScope {          // whatever way this call returns, no later launch sees the flag
    ~StopScope() { g_comic_stop = ComicStop(); }
  } stop_scope;
  for (int t = 0; t < max_steps; ++t) {
    // the launches of step t return at once when the loop ended at an earlier step (see common.h, ComicStop);
    // fused path only: every kernel of its step honours the flag (the unfused path's gathers index through
    // the parents of the step before, which a skipped step leaves unwritten)
    if (fused) {
      g_comic_stop.p = steps_executed;
      g_comic_stop.t = t;
    }
    int32_t* word = step_ids + (size_t)t * R;
    int32_t* parent = parent_ids + (size_t)t * R;
    const int nxt = cur ^ 1;
    if (fused) {
      // raw step outputs ping-pong in c/h/att[]; the NEXT step's operand prep gathers them through the
      // parents chosen here (no gather / copy / embedding kernels in between)
      StepBufs sb = ws.sb;
      sb.c2 = ws.c[nxt];
      sb.h2 = ws.h[nxt];
      if (!d->context_layer) sb.ctx = ws.att[nxt];
      else sb.att2 = ws.att[nxt];
      const int32_t* ids_in = t == 0 ? ws.ids : step_ids + (size_t)(t - 1) * R;
      const int32_t* par_in = t == 0 ? nullptr : parent_ids + (size_t)(t - 1) * R;
      sm.skip_prep = (stream_lstm && (stream_logits || small_step) && t > 0) ? 1 : 0;
      sm.wo_part = nullptr;
      StreamBufs* smp = stream_lstm ? &sm : nullptr;
      RC(infer_step_lstm(d, p, ws.kpanel, ids_in, par_in, W, ws.c[cur], ws.h[cur], ws.att[cur], sb, ws.gtmp, R, st, smp));
      // (the two chains that hang off y -- query projection + attention, vocabulary projection + top-k -- measured
      // slower on two lanes than back to back: 98.5 vs 94.8 us per step, the fork / join of a 20 us branch costs more
      // than the overlap returns)
      if (stream_logits && stream_lstm && sm.wqfrag) {
        // projection launch first, with the query projection's workgroups riding on the CUs its chunks leave idle; the
        // attention step (which needs q) and the merge (which gathers the attention output) follow
        const float* att_new = d->context_layer ? sb.att2 : sb.ctx;
        LstmPrepArgs prep{p->emb, att_new, sb.h2, sb.c2, (uint4*)ws.xfrag, ws.gtmp, E, A, D, V, (E + A + D + 31) / 32};
        int n_q = 0, q_lds = 0;
        const LstmStreamArgs qa = comic_stream_gemm_args(ws.yfrag, ws.wqfrag, (float*)g_splitk_ws, R, D, D, &sm.q_S, &n_q, &q_lds,
                                                         std::max(8, 256 - comic_beam_logits_chunks(V)));
        RC(comic_beam_logits_launch(sb.y, ws.yfrag, ws.wo_pad, ws.logits, max_steps, B, W, D, V, &qa, n_q, q_lds, st));
        RC(infer_step_attend(d, p, ad, ws.keys, values, sb, attn_hist + (size_t)t * R * H * M, R, st, smp, mem_div));
        RC(comic_beam_merge_launch(ws.logits, ws.log_probs, finished, lengths, word, parent, scores + (size_t)t * R,
                                   steps_executed, t, max_steps, B, W, V, d->end_id, &prep, st));
        cur = nxt;
        continue;
      }
      RC(infer_step_attend(d, p, ad, ws.keys, values, sb, attn_hist + (size_t)t * R * H * M, R, st, smp, mem_div));
      if (stream_logits) {
        // (with the streaming LSTM step the merge also gathers the next step's operand rows: raw c / h / attention
        // outputs of this step through the parents it has just chosen)
        const float* att_new = d->context_layer ? sb.att2 : sb.ctx;
        LstmPrepArgs prep{p->emb, att_new, sb.h2, sb.c2, (uint4*)ws.xfrag, ws.gtmp, E, A, D, V, (E + A + D + 31) / 32};
        RC(comic_beam_logits_step(sb.y, stream_lstm ? ws.yfrag : nullptr, ws.wo_pad, ws.logits, ws.log_probs, finished, lengths, word, parent,
                                  scores + (size_t)t * R, steps_executed, t, max_steps, B, W, D, V, d->end_id,
                                  stream_lstm ? &prep : nullptr, st));
      } else if (small_step) {
        // small vocabulary: the entry's whole step in one workgroup, which also keeps steps_executed and (with the
        // streaming LSTM step) gathers the next step's operand rows
        const float* att_new = d->context_layer ? sb.att2 : sb.ctx;
        LstmPrepArgs prep{p->emb, att_new, sb.h2, sb.c2, (uint4*)ws.xfrag, ws.gtmp, E, A, D, V, (E + A + D + 31) / 32};
        if (stream_wo) {     // vocabulary projection through the streaming kernel: K-slice partials, summed by the step kernel
          int S = sm.wo_S;
          const int ldp = (V + 63) / 64 * 64;
          const float* lp_part = sm.wo_part;               // launched together with the query projection ...
          if (!lp_part) {                                  // ... or on its own
            RC(comic_stream_gemm(ws.yfrag, ws.wo_pad, (float*)g_splitk_ws, kSplitKBytes, R, D, V, &S, st));
            lp_part = (const float*)g_splitk_ws;
          }
          RC(comic_beam_step_small(lp_part, p->b_o, S, ldp, (long)R * ldp, ws.log_probs, finished, lengths,
                                   word, parent, scores + (size_t)t * R, B, W, V, d->end_id, ws.beam_cnt + t, steps_executed,
                                   t, max_steps, &prep, st));
        } else {
          RC(gemm_big(sb.y, w_o, ws.logits, p->b_o, R, V, D, D, ld_wo, V, 0, 0, 0.f, st));
          RC(comic_beam_step_small(ws.logits, nullptr, 1, V, 0, ws.log_probs, finished, lengths, word, parent,
                                   scores + (size_t)t * R, B, W, V, d->end_id, ws.beam_cnt + t, steps_executed, t, max_steps,
                                   stream_lstm ? &prep : nullptr, st));
        }
      } else {
        RC(gemm_big(sb.y, w_o, ws.logits, p->b_o, R, V, D, D, ld_wo, V, 0, 0, 0.f, st));
        if (lpw != 0.f)
          RC(comic_beam_step_lp(ws.logits, ws.log_probs, finished, lengths, word, parent, scores + (size_t)t * R, B, W, V,
                                d->end_id, lpw, st));
        else
          RC(comic_beam_step_ws(ws.logits, ws.log_probs, finished, lengths, word, parent, scores + (size_t)t * R, B, W, V,
                                d->end_id, g_splitk_ws, kSplitKBytes, st));
      }
    } else {
      if (t > 0) (void)hipMemcpyAsync(ws.ids, step_ids + (size_t)(t - 1) * R, sizeof(int32_t) * R,
                                      hipMemcpyDeviceToDevice, st);
      RC(comic_embed_fwd(p->emb, ws.ids, ws.x, R, E, V, (void*)st));
      StepBufs sb = ws.sb;
      sb.c2 = ws.gtmp;
      sb.h2 = ws.gtmp + (size_t)R * D;
      float* att_new = ws.gtmp + (size_t)2 * R * D;
      if (!d->context_layer) sb.ctx = att_new;
      else sb.att2 = att_new;
      RC(infer_step(d, p, ad, ws.keys, values, ws.x, ws.c[cur], ws.h[cur], ws.att[cur], sb,
                    attn_hist + (size_t)t * R * H * M, R, st));
      RC(gemm_big(sb.y, w_o, ws.logits, p->b_o, R, V, D, D, ld_wo, V, 0, 0, 0.f, st));
      if (lpw != 0.f)
        RC(comic_beam_step_lp(ws.logits, ws.log_probs, finished, lengths, word, parent, scores + (size_t)t * R, B, W, V,
                              d->end_id, lpw, st));
      else
        RC(comic_beam_step_ws(ws.logits, ws.log_probs, finished, lengths, word, parent, scores + (size_t)t * R, B, W, V,
                              d->end_id, g_splitk_ws, kSplitKBytes, st));
      RC(comic_gather_rows(sb.c2, parent, ws.c[nxt], R, W, D, (void*)st));
      RC(comic_gather_rows(sb.h2, parent, ws.h[nxt], R, W, D, (void*)st));
      RC(comic_gather_rows(att_new, parent, ws.att[nxt], R, W, A, (void*)st));
    }
    if (!stream_logits && !small_step) {       // (the streaming step's merge / the small step keep steps_executed themselves)
      hipLaunchKernelGGL(all_finished_kernel, dim3(1), dim3(256), 0, st, finished, steps_executed, t, R, max_steps);
      COMIC_LAUNCH_CHECK("all_finished");
    }
    cur = nxt;
  }
  return 0;
}
