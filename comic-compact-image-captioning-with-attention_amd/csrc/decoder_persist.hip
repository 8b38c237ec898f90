// Persistent forward time loop of the attention-LSTM decoder (training, fp32) for gfx950.
//
// Replaces the per-step launch chain of comic_decoder_train_step's forward loop -- BasicLSTMCell + DropoutWrapper
// (src/model_base.py:606-648), the query layer and MultiHeadAddLN / MultiHeadDot + MultiHeadAttentionWrapperV3.call
// (common/ops_rnn.py:531-565, :611-632, :660-755), TrainingHelper / impute_finished of rnn_decoder_training
// (common/ops_rnn.py:183-243) -- with ONE launch that runs all T' steps.
//
// Why: the recurrence is a chain of three dependent ~10 us kernels per step whose bodies are bound by what a CU can
// pull from L2 (about 70 GB/s): every step re-streams the 10.5 MB LSTM kernel, W_q and the 3.3 MB of keys.  Here the
// operands that do not change over time never leave the CU:
//   * a workgroup keeps ITS 32 gate columns of the LSTM kernel (160 KB) in the registers of its eight waves, ITS
//     eight columns of W_q (16 KB) and the keys (= values when tied) of ITS batch row (M*D*4 = 51 KB) in LDS;
//   * the c / h state and the previous attention vector of the elements a workgroup owns stay in registers.
// Per step only activations cross workgroups: [x ; att ; h] rows (LSTM operand), y (query operand), q.
//
// Decomposition.  The batch splits into independent groups of 16 rows; a group is served by 64 workgroups (one per
// CU; four groups fill the chip at batch 64) that never exchange anything with another group.  Workgroup i of a
// group runs, per step:
//   L  units [8i, 8i+8) x 16 rows:  gates = [x ; att ; h] * K + b (exact fp32 MFMA, the eight waves split K, fixed-
//      order combine) -> cell, output dropout, finished-row select       writes y_t, h part of the next operand row
//   Q  query columns [8i, 8i+8) x 16 rows:  q = y * W_q                                                  writes q_t
//   A  batch row (i % 16), channel quarter (i / 16):  LayerNorm statistics of keys + q over all D (from the
//      resident keys), tanh / v scores of the quarter's heads, softmax (or sigmoid norm) over M, dropout, context of
//      its channels, finished-row select, input dropout                   writes the att part of the next operand row
// The x and h thirds of the next step's LSTM product do not depend on the attention: they run between Q and A,
// under the latency of the q hand-off; only the att third sits on the critical path.
//
// Hand-offs carry no flag and need no barrier: THE DATA IS THE FLAG.  Every handed-off buffer (operand rows, y, q)
// is time-major, so each 4-byte word is written exactly once per launch; the caller fills them with a NaN pattern no
// computation produces (kSentinel) before the launch, producers write them with sc1 (write-through) stores, and a
// consumer re-reads with sc1 loads (registers only, never L1) until none of its words holds the pattern -- the
// granule form of MI355X_MICROARCH.md's hand-off recipes ("Persistent kernels: synchronisation and hand-off price
// list", handoff-1to1 / allgather; cdna_hip_programming.md Guideline 16, R2) with a 4-byte granule: an aligned dword
// store is never torn and a word needs no ordering against any other word.  A counter barrier cost 3.0 us of the
// 24.7 us step three times over (measured, profiles/r02_persist_phases.txt); a validated load costs one more round
// trip only when the data is late.  Data that only later kernels read (saved gates, alpha, ...) uses plain stores.
// Every spin is bounded: a wave that polls more than kSpinLimit times raises the error word, every wave that sees it
// stops waiting, and comic_persist_gate (end of the step) turns the step's losses into NaN and its gradients into zeros.  One workgroup
// per CU (the launch reserves more than half of the LDS): with 256 CUs all workgroups of a launch are resident.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "decoder_persist_dev.h"

#include "decoder_math.h"

namespace {

// diagnostic (COMIC_PERSIST_STAMPS=1): workgroup 0 records the 100 MHz clock at the phase edges of every step
__device__ __forceinline__ void stamp(unsigned long long* st, int t, int i) {
  if (st && blockIdx.x == 0 && threadIdx.x == 0) st[t * 8 + i] = __builtin_amdgcn_s_memrealtime();
}

// A wave's k16-blocks of the operand row [x ; att ; h]: NX of the x third (block wave + 8 i), then four of the att third
// and four of the h third (pairs of adjacent blocks: the two loads of a row share a 128-B line).
// GREEDY: the same loop as the greedy decode of rnn_decoder_search (common/ops_rnn.py:115-180; GreedyEmbeddingHelper): no
// teacher forcing, no dropout, no finished-row select; the Q phase also forms this workgroup's columns of the logits
// y W_o + b_o and their row maxima, every workgroup reduces the 64 partial maxima of its rows to the step's token ids
// and the x third of the next operand row is read straight from the embedding table.
// BIGM (64 < M <= 256, tied keys / values; the reference CLI's default map, Inception-V1 Mixed_4f: M = 196): the keys of
// a batch row (M*D*4 = 401 KB) do not fit a CU, so a workgroup keeps only ITS CHANNEL QUARTER of them ([M][128], 100 KB).
// The LayerNorm statistics of keys + q run over all D channels: every quarter forms the partial sums (sum z, sum z^2) of
// its 128 channels for all M rows and the four workgroups of a batch row exchange them through one more sentinel-checked
// hand-off (M/2 pieces of 16 bytes per quarter); mean and variance come from the combined sums (var = E[z^2] - mean^2).
template <int NX, bool WQ_LDS, bool GREEDY, bool BIGM = false>
__global__ __launch_bounds__(kThreads) void decoder_fwd_persistent_kernel(ComicPersistFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = kD;
  const int M = a.M, H = a.H, Wd = a.Wd, E = a.E, EA = a.E + D, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = a.grp0 + blockIdx.x / kGroupWgs, wi = blockIdx.x % kGroupWgs;   // group index over the whole batch
  const int row0 = grp * kGroupRows;
  Waiter wt{a.sync, false};

  // ---- LDS carve-up ---------------------------------------------------------------------------------------------
  constexpr int KP = BIGM ? D / 4 : D;                         // floats per resident key row (BIGM: the quarter's channels)
  constexpr int SCP = BIGM ? 256 : 64;                         // floats per head row of the score buffer
  float* keys_l = (float*)smem;                                // [M][KP] keys of this workgroup's attention row
  float* vals_l = (a.tied || BIGM) ? keys_l : keys_l + M * D;  // [M][D]  values (independent projection)
  float4* red = (float4*)(vals_l + M * KP);                    // [8 waves][2 tiles][64 lanes]   cross-wave combine
  float* q_l = (float*)(red + kWaves * 2 * 64);                // [D]      q row of the attention phase
  float* sc_l = q_l + D;                                       // [<= 4 heads][SCP]
  float* st_l = sc_l + 4 * SCP;                                // BIGM: [256][2] mean, 1/std of the rows; [3][128] gamma, beta, v
  float* lnq_l = st_l + 512;
  float* wq_l = sc_l + 4 * SCP + (BIGM ? 512 + 3 * (D / 4) : 0);   // [D][8] + 4 floats per 16 rows: eight W_q columns
  float* wo_l = wq_l + (WQ_LDS ? 8 * D + (D / 16) * 4 : 0);     // GREEDY: [D][8] + pad, this workgroup's logit columns of W_o
  int* ids_l = (int*)(wo_l + 8 * D + (D / 16) * 4);            // GREEDY: [16] token ids of the group's rows, [16] first-EOS steps

  const __amdgpu_buffer_rsrc_t xh_r = make_rsrc(a.xh_all, (long)a.Tp * B * Wd * 4);
  const __amdgpu_buffer_rsrc_t y_r = make_rsrc(a.y_all, (long)a.Tp * B * D * 4);
  const __amdgpu_buffer_rsrc_t q_r = make_rsrc(a.q_all, (long)a.Tp * B * D * 4);
  const __amdgpu_buffer_rsrc_t ap_r = make_rsrc(a.argp, GREEDY ? (long)a.Tp * B * kArgRow * 4 : 0);
  const __amdgpu_buffer_rsrc_t sp_r = make_rsrc(a.statp, BIGM ? (long)a.Tp * B * 8 * M * 4 : 0);   // [t][row][quarter][M/2][4]
  const int ncol = GREEDY ? (a.V + kGroupWgs - 1) / kGroupWgs : 0;   // logit columns per workgroup (<= 8)

  // ---- attention-phase identity: batch row + channel quarter --------------------------------------------------------
  const int ab = row0 + (wi & 15), aq = wi >> 4;
  const bool a_live = ab < B;
  const int cq0 = aq * (D / 4);                                // first channel of the quarter
  const int dh = D / H, hq = (D / 4) / dh, lph = dh / 2;       // head width, heads per quarter, lanes per head
  {
    const int arow = a_live ? ab : 0;
    const float4* ks = (const float4*)(a.keys + (size_t)arow * M * D);
    if constexpr (BIGM) {
      for (int i = tid; i < M * (KP / 4); i += kThreads)
        ((float4*)keys_l)[i] = *(const float4*)(a.keys + ((size_t)arow * M + (i >> 5)) * D + (wi >> 4) * (D / 4) + 4 * (i & 31));
      for (int i = tid; i < 3 * (D / 4); i += kThreads) {
        const int w = i / (D / 4), c = (wi >> 4) * (D / 4) + i % (D / 4);
        lnq_l[i] = a.method == 0 ? (w == 0 ? a.ln_g[c] : w == 1 ? a.ln_b[c] : a.v[c]) : 0.f;
      }
    } else {
      for (int i = tid; i < M * D / 4; i += kThreads) ((float4*)keys_l)[i] = ks[i];
    }
    if (!a.tied && !BIGM) {
      const float4* vs = (const float4*)(a.values + (size_t)arow * M * D);
      for (int i = tid; i < M * D / 4; i += kThreads) ((float4*)vals_l)[i] = vs[i];
    }
    if (WQ_LDS) {   // row k at 8k + 4(k >> 4) floats: a thread's 16 rows are 528 B from its neighbour's (no bank conflict)
      for (int k = tid; k < D; k += kThreads) {
        float* dst = wq_l + 8 * k + 4 * (k >> 4);
        *(float4*)dst = *(const float4*)(a.W_q + (size_t)k * D + 8 * wi);
        *(float4*)(dst + 4) = *(const float4*)(a.W_q + (size_t)k * D + 8 * wi + 4);
      }
    }
    if constexpr (GREEDY) {   // W_o[k][ncol wi + c] for c < ncol (0 past V), same padded layout as the W_q columns
      for (int i = tid; i < D * 8; i += kThreads) {
        const int k = i >> 3, c = i & 7, col = ncol * wi + c;
        wo_l[8 * k + 4 * (k >> 4) + c] = (c < ncol && col < a.V) ? a.W_o[(size_t)k * a.ld_wo + col] : 0.f;
      }
      if (tid < 16) {
        ids_l[tid] = a.start_id;
        ids_l[16 + tid] = a.Tp;
      }
      if (tid == 0) ids_l[32] = ids_l[33] = 0;
    }
  }

  // ---- L phase residents: this wave's k-blocks of the two unit tiles' weight panels -----------------------------------
  const int r16 = lane & 15, kq = lane >> 4;
  const int KB = (Wd + 15) >> 4;
  constexpr int NB = NX + 8;
  const int xb = E >> 4;                                       // k16-blocks of the x third
  float4 wreg[NB][2];
  unsigned kb_off[NB];                                         // byte offset of the (clamped) block in an operand row
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = i - NX;
    const int kb = i < NX ? wave + kWaves * i : xb + (j >> 2) * (D / 16) + 2 * (wave + kWaves * ((j >> 1) & 1)) + (j & 1);
    const bool real = i >= NX || kb < xb;                      // an x block past the third has zero weights
    kb_off[i] = (unsigned)(real ? kb : 0) * 64u;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const float* panel = a.K_panel + ((size_t)(2 * wi + jt) * KB * 16 + r16) * 16 + 4 * kq;
      wreg[i][jt] = real ? *(const float4*)(panel + (size_t)kb * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // epilogue lanes (wave j = unit tile j): state of (row, unit) lives in registers across the steps
  const int e_row = row0 + r16, e_d = 8 * wi + 4 * wave + kq;
  const bool e_lane = wave < 2 && e_row < B;
  float e_c = 0.f, e_h = 0.f, e_b[4] = {0.f, 0.f, 0.f, 0.f};
  int e_len = 0;
  if (e_lane) {
    e_c = a.cs[(size_t)e_row * D + e_d];
    e_h = a.hs[(size_t)e_row * D + e_d];
    e_b[0] = a.bias[e_d]; e_b[1] = a.bias[D + e_d]; e_b[2] = a.bias[2 * D + e_d]; e_b[3] = a.bias[3 * D + e_d];
    e_len = GREEDY ? 0x7fffffff : a.lens[e_row];
  }
  const int l_row = min(row0 + r16, B - 1);                    // operand row of this lane (clamped: results unused)
  // attention epilogue threads (tid < D/4): previous attention state of (row, channel)
  const int a_c = cq0 + tid;
  float att_prev = 0.f;
  const int a_len = GREEDY ? 0x7fffffff : (a_live ? a.lens[ab] : 0);
  float lnp[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};     // gamma, beta, v of this lane's two quarter channels
  if (a.method == 0) {
    const int c = cq0 + 2 * lane;
    lnp[0][0] = a.ln_g[c]; lnp[0][1] = a.ln_g[c + 1];
    lnp[1][0] = a.ln_b[c]; lnp[1][1] = a.ln_b[c + 1];
    lnp[2][0] = a.v[c]; lnp[2][1] = a.v[c + 1];
  }
  const float inv_scale = 1.0f / (a.method == 0 ? a.tau[0] : sqrtf((float)dh));
  __syncthreads();

  // The LSTM product of a step in two parts: blocks [I0, I1) of the wave.
  f32x4_t acc[2];
  auto lstm_part = [&](int t, auto i0_, auto i1_) {
    constexpr int I0 = decltype(i0_)::value, I1 = decltype(i1_)::value, N = I1 - I0;
    const unsigned xo = (unsigned)((((size_t)t * B + l_row) * Wd + 4 * kq) * 4);
    float4 xa[N];
    unsigned off[N];
    if constexpr (GREEDY && I0 == 0) {    // x third: the embedding row of the token this row emitted (plain loads: a table)
      const float* er = a.emb + (size_t)ids_l[r16] * E + 4 * kq;
#pragma unroll
      for (int i = 0; i < N; ++i) xa[i] = *(const float4*)(er + kb_off[I0 + i] / 4);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        off[i] = kb_off[I0 + i];
        xa[i] = load16_sc1(xh_r, xo + off[i]);
      }
      wait_written<N>(xa, xh_r, xo, off, (1u << N) - 1u, wt);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[I0 + i][j].x, xa[i].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[I0 + i][j].y, xa[i].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[I0 + i][j].z, xa[i].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[I0 + i][j].w, xa[i].w, acc[j], 0, 0, 0);
    }
  };
  using std::integral_constant;
  const integral_constant<int, 0> c0;
  const integral_constant<int, NX> cx;
  const integral_constant<int, NX + 4> ca;
  const integral_constant<int, NX + 8> ch;
  acc[0] = acc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  lstm_part(0, c0, cx);
  lstm_part(0, ca, ch);

  for (int t = 0; t < a.Tp; ++t) {
    // masks of this step: fetched ahead of the phases that use them
    float m_out = 1.f, m_in = 1.f, m_al = 1.f;
    const size_t e_i = ((size_t)t * B + e_row) * D + e_d;
    if (a.mask_out && e_lane) m_out = a.mask_out[e_i];
    if (a.mask_in && a_live && tid < D / 4 && t + 1 < a.Tp) m_in = a.mask_in[((size_t)(t + 1) * B + ab) * EA + E + a_c];
    float m_al4[4] = {1.f, 1.f, 1.f, 1.f};                     // BIGM: a lane holds rows lane + 64 k
    if constexpr (BIGM) {
      if (a.mask_alpha && a_live && wave < hq) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lane + 64 * k < M) m_al4[k] = a.mask_alpha[(((size_t)t * B + ab) * H + aq * hq + wave) * M + lane + 64 * k];
      }
    } else if (a.mask_alpha && a_live && wave < hq && lane < M) {
      m_al = a.mask_alpha[(((size_t)t * B + ab) * H + aq * hq + wave) * M + lane];
    }
    // =============================================================== L: LSTM cell (att third) ========================
    stamp(a.stamps, t, 0);
    lstm_part(t, cx, ca);
#pragma unroll
    for (int j = 0; j < 2; ++j) red[(wave * 2 + j) * 64 + lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    __syncthreads();
    stamp(a.stamps, t, 1);
    if (wave < 2) {
      float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {                        // fixed order: deterministic
        const float4 pp = red[(w * 2 + wave) * 64 + lane];
        g[0] += pp.x; g[1] += pp.y; g[2] += pp.z; g[3] += pp.w;
      }
      const float si = sigmoidf_(g[0] + e_b[0]), tj = tanhf(g[1] + e_b[1]);
      const float sf = sigmoidf_(g[2] + e_b[2] + 1.0f), so = sigmoidf_(g[3] + e_b[3]);   // forget_bias = 1
      const float c2 = e_c * sf + si * tj;
      const float h2 = tanhf(c2) * so;
      const float yv = a.mask_out ? (h2 / a.keep_out) * m_out : h2;
      const bool fin = t >= e_len;
      e_c = fin ? e_c : c2;
      e_h = fin ? e_h : h2;
      // the four units of a row sit in lanes r16 + 16 kq: gather them into the kq == 0 lane, one 16-byte store each
      float y4[4], h4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y4[k] = __shfl(yv, r16 + 16 * k, 64);
        h4[k] = __shfl(e_h, r16 + 16 * k, 64);
      }
      if (kq == 0 && e_row < B) {
        const size_t i4 = ((size_t)t * B + e_row) * D + 8 * wi + 4 * wave;
        store16_sc1(y_r, (unsigned)(i4 * 4), make_float4(y4[0], y4[1], y4[2], y4[3]));
        if (t + 1 < a.Tp)
          store16_sc1(xh_r, (unsigned)((((size_t)(t + 1) * B + e_row) * Wd + EA + 8 * wi + 4 * wave) * 4),
                      make_float4(h4[0], h4[1], h4[2], h4[3]));
      }
      if (!GREEDY && e_lane) {      // saved for the backward loop
        float* ga = a.gates_all + ((size_t)t * B + e_row) * 4 * D;
        ga[e_d] = si; ga[D + e_d] = tj; ga[2 * D + e_d] = sf; ga[3 * D + e_d] = so;
        a.cnew_all[e_i] = c2;
        a.cs[((size_t)(t + 1) * B + e_row) * D + e_d] = e_c;
        a.hs[((size_t)(t + 1) * B + e_row) * D + e_d] = e_h;
      }
    }
    __syncthreads();   // the six other waves start polling y only after this workgroup's own stores are on their way
    stamp(a.stamps, t, 2);
    // =============================================================== Q: query layer =================================
    {
      if (!WQ_LDS) asm volatile("" ::: "memory");               // W_q from L2 every step: 128 registers would not fit
      const int rl = tid >> 5, part = tid & 31;                 // row of the group, k range [16 part, 16 part + 16)
      const int row = min(row0 + rl, B - 1);
      const unsigned yo = (unsigned)((((size_t)t * B + row) * D + 16 * part) * 4);
      const unsigned yoff[4] = {0u, 16u, 32u, 48u};
      float4 yv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] = load16_sc1(y_r, yo + yoff[i]);
      wait_written<4>(yv, y_r, yo, yoff, 15u, wt);
      float q8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ys[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w};
        if (!WQ_LDS) asm volatile("" ::: "memory");             // at most eight W_q loads in flight (registers)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 16 * part + 4 * i + e;
          float4 w0, w1;
          if (WQ_LDS) {
            const float* src = wq_l + 8 * k + 4 * part;
            w0 = *(const float4*)src;
            w1 = *(const float4*)(src + 4);
          } else {
            w0 = *(const float4*)(a.W_q + (size_t)k * D + 8 * wi);
            w1 = *(const float4*)(a.W_q + (size_t)k * D + 8 * wi + 4);
          }
          q8[0] = fmaf(ys[e], w0.x, q8[0]); q8[1] = fmaf(ys[e], w0.y, q8[1]);
          q8[2] = fmaf(ys[e], w0.z, q8[2]); q8[3] = fmaf(ys[e], w0.w, q8[3]);
          q8[4] = fmaf(ys[e], w1.x, q8[4]); q8[5] = fmaf(ys[e], w1.y, q8[5]);
          q8[6] = fmaf(ys[e], w1.z, q8[6]); q8[7] = fmaf(ys[e], w1.w, q8[7]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q8[e] = group_sum_dpp(q8[e], 16);
        q8[e] += __shfl_xor(q8[e], 16, 64);
      }
      if (part == 0 && row0 + rl < B) {
        const unsigned qo = (unsigned)((((size_t)t * B + row0 + rl) * D + 8 * wi) * 4);
        store16_sc1(q_r, qo, make_float4(q8[0], q8[1], q8[2], q8[3]));
        store16_sc1(q_r, qo + 16, make_float4(q8[4], q8[5], q8[6], q8[7]));
      }
      if constexpr (GREEDY) {       // this workgroup's columns of the logits, and their maximum per row
        float l8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ys[4] = {yv[i].x, yv[i].y, yv[i].z, yv[i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = 16 * part + 4 * i + e;
            const float* src = wo_l + 8 * k + 4 * part;
            const float4 w0 = *(const float4*)src, w1 = *(const float4*)(src + 4);
            l8[0] = fmaf(ys[e], w0.x, l8[0]); l8[1] = fmaf(ys[e], w0.y, l8[1]);
            l8[2] = fmaf(ys[e], w0.z, l8[2]); l8[3] = fmaf(ys[e], w0.w, l8[3]);
            l8[4] = fmaf(ys[e], w1.x, l8[4]); l8[5] = fmaf(ys[e], w1.y, l8[5]);
            l8[6] = fmaf(ys[e], w1.z, l8[6]); l8[7] = fmaf(ys[e], w1.w, l8[7]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          l8[e] = group_sum_dpp(l8[e], 16);
          l8[e] += __shfl_xor(l8[e], 16, 64);
        }
        // Leaving the loop: workgroup 0 of the group decides and publishes the decision with its partial of this step
        // (word 128 of the group's first row), so that all 64 workgroups leave after the SAME step.  It says "stop"
        // once its rows had all emitted EOS before this step and it has seen the same word of every other group of
        // the launch: no row stops before the step the host trims to; a late sighting costs a step.
        if (wi == 0 && tid == 0) {
          bool every = true;
          for (int r = 0; r < kGroupRows && row0 + r < B; ++r) every &= ids_l[16 + r] < t;
          for (int g = a.grp0; every && g < a.grp0 + a.n_groups; ++g)
            every = g == grp || __hip_atomic_load(a.sync + 32 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
          store4_sc1(ap_r, (unsigned)((((size_t)t * B + row0) * kArgRow + 128) * 4), every ? 1.f : 0.f);
        }
        if (part == 0 && row0 + rl < B) {
          float best = -INFINITY;
          int bi = 0x7fffffff;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int col = ncol * wi + e;
            if (e < ncol && col < a.V) {
              const float v = l8[e] + a.b_o[col];
              if (a.logits_tb) a.logits_tb[((size_t)t * B + row0 + rl) * a.V + col] = v;
              if (v > best) {                                   // first maximum: lowest column wins a tie
                best = v;
                bi = col;
              }
            }
          }
          // (value, column) of this workgroup for the row; a workgroup without columns publishes (-inf, INT_MAX)
          const unsigned po = (unsigned)((((size_t)t * B + row0 + rl) * kArgRow + 2 * wi) * 4);
          store4_sc1(ap_r, po, best);
          store4_sc1(ap_r, po + 4, __int_as_float(bi));
        }
      }
    }
    stamp(a.stamps, t, 3);
    // ============================================ next step's x and h thirds, under the q hand-off ====================
    acc[0] = acc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (GREEDY) {
      if (t + 1 < a.Tp) lstm_part(t + 1, ca, ch);              // h third first: the ids are still on their way
      __syncthreads();                                          // ids_l of the previous step is no longer read
      // the step's token ids: 64 partial (value, column) maxima per row, two per thread
      {
        const int rl = tid >> 5, j = tid & 31;
        const int row = min(row0 + rl, B - 1);
        const unsigned po = (unsigned)((((size_t)t * B + row) * kArgRow + 4 * j) * 4);
        const unsigned zoff[1] = {0u};
        float4 pv[1] = {load16_sc1(ap_r, po)};
        wait_written<1>(pv, ap_r, po, zoff, 1u, wt);
        float bv = pv[0].x;
        int bi = __float_as_int(pv[0].y);
        if (pv[0].z > bv) {                                     // columns grow with the workgroup index: ties keep the lower
          bv = pv[0].z;
          bi = __float_as_int(pv[0].w);
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
          }
        }
        if (j == 0) {
          ids_l[rl] = bi;
          if (bi == a.end_id && ids_l[16 + rl] > t) ids_l[16 + rl] = t;
          if (wi == 0 && row0 + rl < B) {
            a.ids_tb[(size_t)t * B + row0 + rl] = bi;
            if (ids_l[16 + rl] == t) a.first_eos[row0 + rl] = t;
          }
        }
        if (tid == 0) {                                         // workgroup 0's "stop after this step"
          const unsigned fo = (unsigned)((((size_t)t * B + row0) * kArgRow + 128) * 4);
          const unsigned z4[1] = {0u};
          float4 fv[1] = {load16_sc1(ap_r, fo)};                // words 128..131 of the row: only 128 is written
          unsigned spins = 0;
          while (__float_as_uint(fv[0].x) == kSentinel && !wt.spin(spins)) fv[0] = load16_sc1(ap_r, fo);
          (void)z4;
          ids_l[32] = fv[0].x == 1.f ? 1 : 0;
        }
      }
      __syncthreads();
      if (wi == 0 && tid == 0 && ids_l[33] == 0) {              // all rows of this group have emitted EOS: say so, once
        bool mine = true;
        for (int r = 0; r < kGroupRows && row0 + r < B; ++r) mine &= ids_l[16 + r] <= t;
        if (mine) {
          __hip_atomic_store(a.sync + 32 + grp, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ids_l[33] = 1;
        }
      }
      if (t + 1 < a.Tp) lstm_part(t + 1, c0, cx);              // x third from the embedding table
    } else if (t + 1 < a.Tp) {
      lstm_part(t + 1, c0, cx);
      lstm_part(t + 1, ca, ch);
    }
    // =============================================================== A: attention ===================================
    if (a_live) {
      if (wave < 2) {
        const unsigned qo = (unsigned)((((size_t)t * B + ab) * D + 4 * tid) * 4);
        const unsigned zoff[1] = {0u};
        float4 qv[1] = {load16_sc1(q_r, qo)};
        wait_written<1>(qv, q_r, qo, zoff, 1u, wt);
        *(float4*)(q_l + 4 * tid) = qv[0];
      }
      __syncthreads();
      stamp(a.stamps, t, 4);
      // scores of this quarter's heads; a wave owns memory rows m = wave, wave + 8, ... and takes two per pass (the
      // reductions of the two rows interleave); the phase is VALU-bound (about 150 wave instructions per row)
      if constexpr (BIGM) {
        if (a.method == 0) {
          // (i) partial statistics of this quarter's 128 channels: thread (row tid >> 1, channel half tid & 1); the walk
          // over the 64 channels starts at a per-thread rotation so that the lanes of a wave hit distinct LDS banks
          const int pm = tid >> 1, ph = tid & 1;
          float s1 = 0.f, s2 = 0.f;
          if (pm < M) {
            const float* kr = keys_l + pm * KP + 64 * ph;
            const float* qq = q_l + cq0 + 64 * ph;
            const int rot = pm + 16 * ph;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) {                      // two channels a read (8-byte LDS reads, 32 distinct pairs)
              const int c = 2 * ((j + rot) & 31);
              const float2 kv = *(const float2*)(kr + c), qv = *(const float2*)(qq + c);
              const float z0 = kv.x + qv.x, z1 = kv.y + qv.y;
              s1 += z0; t1 += z1;
              s2 = fmaf(z0, z0, s2); t2 = fmaf(z1, z1, t2);
            }
            s1 += t1;
            s2 += t2;
          }
          s1 += dpp_move<0xB1>(0.f, s1);                        // the two halves of a row: adjacent lanes
          s2 += dpp_move<0xB1>(0.f, s2);
          const float n1 = __shfl_down(s1, 2, 64), n2 = __shfl_down(s2, 2, 64);   // the next row's sums
          const unsigned sbase = (unsigned)(((size_t)t * B + ab) * 8 * M * 4);    // bytes; quarter qq at qq * 2M floats
          if ((tid & 3) == 0 && pm < M) store16_sc1(sp_r, sbase + (unsigned)((aq * 2 * M + 2 * pm) * 4), make_float4(s1, s2, n1, n2));
          __syncthreads();                                      // poll only after the own stores are on their way
          // (ii) the four quarters' sums of a row pair -> mean, 1/std (combined in quarter order: the same in all four)
          if (wave < 2) {
            const int pr = min(tid, M / 2 - 1);
            const unsigned po = sbase + (unsigned)(4 * pr * 4);
            const unsigned qoff[4] = {0u, (unsigned)(2 * M * 4), (unsigned)(4 * M * 4), (unsigned)(6 * M * 4)};
            float4 pv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = load16_sc1(sp_r, po + qoff[k]);
            wait_written<4>(pv, sp_r, po, qoff, 15u, wt);
            const float S0 = (pv[0].x + pv[1].x) + (pv[2].x + pv[3].x), Q0 = (pv[0].y + pv[1].y) + (pv[2].y + pv[3].y);
            const float S1 = (pv[0].z + pv[1].z) + (pv[2].z + pv[3].z), Q1 = (pv[0].w + pv[1].w) + (pv[2].w + pv[3].w);
            const float mean0 = S0 * (1.0f / (float)D), mean1 = S1 * (1.0f / (float)D);
            const float var0 = fmaxf(Q0 * (1.0f / (float)D) - mean0 * mean0, 0.f);
            const float var1 = fmaxf(Q1 * (1.0f / (float)D) - mean1 * mean1, 0.f);
            if (tid < M / 2)
              *(float4*)(st_l + 4 * tid) = make_float4(mean0, __frsqrt_rn(var0 + kLnEps), mean1, __frsqrt_rn(var1 + kLnEps));
          }
          __syncthreads();
        }
        // (iii) scores: one (row, head of the quarter) per thread over the head's dh channels
        for (int idx = tid; idx < M * hq; idx += kThreads) {
          const int m = idx / hq, hl = idx - m * hq;
          const float* kr = keys_l + m * KP + hl * dh;
          const float* qq = q_l + cq0 + hl * dh;
          float acc_s = 0.f;
          if (a.method == 0) {
            const float mean = st_l[2 * m], rstd = st_l[2 * m + 1];
            const float* lg = lnq_l + hl * dh;
            float acc_t = 0.f;
            // tanh(x) = 1 - 2 / (1 + e^(2x)) on v_exp_f32 / v_rcp_f32: |error| < 2e-7 absolute (what a score, a sum of
            // tanh * v, needs); M * 128 of them per step make this loop the phase's cost
            auto th = [](float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); };
#pragma unroll 4
            for (int j = 0; j < dh / 2; ++j) {
              const int c = 2 * ((j + idx) & (dh / 2 - 1));
              const float2 kv = *(const float2*)(kr + c), qv = *(const float2*)(qq + c);
              const float2 gv = *(const float2*)(lg + c), bv = *(const float2*)(lg + D / 4 + c), vv = *(const float2*)(lg + 2 * (D / 4) + c);
              const float i0 = rstd * gv.x, i1 = rstd * gv.y;
              const float zh0 = (kv.x + qv.x) * i0 + (bv.x - mean * i0);   // tf.nn.batch_normalization form
              const float zh1 = (kv.y + qv.y) * i1 + (bv.y - mean * i1);
              acc_s = fmaf(th(zh0), vv.x, acc_s);
              acc_t = fmaf(th(zh1), vv.y, acc_t);
            }
            acc_s += acc_t;
          } else {
#pragma unroll 4
            for (int j = 0; j < dh / 2; ++j) {
              const int c = 2 * ((j + idx) & (dh / 2 - 1));
              const float2 kv = *(const float2*)(kr + c), qv = *(const float2*)(qq + c);
              acc_s = fmaf(kv.x, qv.x, fmaf(kv.y, qv.y, acc_s));
            }
          }
          sc_l[hl * SCP + m] = acc_s * inv_scale;
        }
      } else {
      auto score_rows = [&](int m0, auto nr_) {
        constexpr int NR = decltype(nr_)::value;
        const int c = cq0 + 2 * lane;                           // this lane's two channels of the quarter
        const float* kr[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) kr[r] = keys_l + (m0 + r * kWaves) * D;
        float part0[NR];
        if (a.method == 0) {
          constexpr int EPL = 8;                                // statistics over all D = 64 lanes x 8 channels
          float z[NR][EPL], s[NR];
          const float4 qa = *(const float4*)(q_l + lane * EPL), qb = *(const float4*)(q_l + lane * EPL + 4);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float4 ka = *(const float4*)(kr[r] + lane * EPL), kb = *(const float4*)(kr[r] + lane * EPL + 4);
            z[r][0] = ka.x + qa.x; z[r][1] = ka.y + qa.y; z[r][2] = ka.z + qa.z; z[r][3] = ka.w + qa.w;
            z[r][4] = kb.x + qb.x; z[r][5] = kb.y + qb.y; z[r][6] = kb.z + qb.z; z[r][7] = kb.w + qb.w;
            s[r] = ((z[r][0] + z[r][1]) + (z[r][2] + z[r][3])) + ((z[r][4] + z[r][5]) + (z[r][6] + z[r][7]));
          }
          float mean[NR], s2[NR], rstd[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) mean[r] = wave_sum(s[r]) * (1.0f / (float)D);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            s2[r] = 0.f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
              const float cc = z[r][i] - mean[r];
              s2[r] = fmaf(cc, cc, s2[r]);
            }
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) rstd[r] = __frsqrt_rn(wave_sum(s2[r]) * (1.0f / (float)D) + kLnEps);
          const float2 qc = *(const float2*)(q_l + c);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float2 kc = *(const float2*)(kr[r] + c);
            const float zz[2] = {kc.x + qc.x, kc.y + qc.y};
            part0[r] = 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float inv = rstd[r] * lnp[0][e];
              const float zh = zz[e] * inv + (lnp[1][e] - mean[r] * inv);   // tf.nn.batch_normalization form
              part0[r] += fast_tanh(zh) * lnp[2][e];
            }
          }
        } else {
          const float2 qc = *(const float2*)(q_l + c);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float2 kc = *(const float2*)(kr[r] + c);
            part0[r] = kc.x * qc.x + kc.y * qc.y;
          }
        }
        // sum over the lanes of a head (16, 32 or 64): the total lands in the head's LAST lane
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          float hsum = group_sum_dpp(part0[r], 16);
          if (lph >= 32) hsum += dpp_move<0x142, 0xA>(0.f, hsum);   // row_bcast:15 into rows 1, 3
          if (lph == 64) hsum += dpp_move<0x143, 0xC>(0.f, hsum);   // row_bcast:31 into rows 2, 3
          if ((lane % lph) == lph - 1) sc_l[(lane / lph) * 64 + m0 + r * kWaves] = hsum * inv_scale;
        }
      };
      for (int m0 = wave; m0 < M; m0 += 2 * kWaves) {
        if (m0 + kWaves < M) score_rows(m0, integral_constant<int, 2>());
        else score_rows(m0, integral_constant<int, 1>());
      }
      }
      __syncthreads();
      stamp(a.stamps, t, 5);
      // probability fn per head (a wave per head of the quarter), dropout; sc <- alpha_d
      if (BIGM && wave < hq) {                                  // a lane holds rows lane + 64 k
        const int h = aq * hq + wave;
        float* srow = sc_l + wave * SCP;
        const size_t go = (((size_t)t * B + ab) * H + h) * M;
        float sv[4], al[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[k] = lane + 64 * k < M ? srow[lane + 64 * k] : -INFINITY;
        if (a.prob == 0) {
          const float mx = wave_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
#pragma unroll
          for (int k = 0; k < 4; ++k) al[k] = lane + 64 * k < M ? expf(sv[k] - mx) : 0.f;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) al[k] = lane + 64 * k < M ? sigmoidf_(sv[k]) : 0.f;
        }
        const float tot = wave_sum((al[0] + al[1]) + (al[2] + al[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = lane + 64 * k;
          if (m < M) {
            const float av = al[k] / tot;
            a.alpha_all[go + m] = av;
            const float ad = a.mask_alpha ? (av / a.keep_alpha) * m_al4[k] : av;
            a.attn_hist[go + m] = ad;
            srow[m] = ad;
          }
        }
      } else if (!BIGM && wave < hq) {
        const int h = aq * hq + wave;
        float* srow = sc_l + wave * 64;
        const size_t go = (((size_t)t * B + ab) * H + h) * M;
        const float sv = lane < M ? srow[lane] : -INFINITY;
        float al;
        if (a.prob == 0) {
          const float mx = wave_max(sv);
          const float ex = lane < M ? expf(sv - mx) : 0.f;
          al = ex / wave_sum(ex);
        } else {
          const float sg = lane < M ? sigmoidf_(sv) : 0.f;
          al = sg / wave_sum(sg);
        }
        if (lane < M) {
          if (!GREEDY) a.alpha_all[go + lane] = al;
          const float ad = a.mask_alpha ? (al / a.keep_alpha) * m_al : al;
          a.attn_hist[go + lane] = ad;
          srow[lane] = ad;
        }
      }
      __syncthreads();
      stamp(a.stamps, t, 6);
      if constexpr (BIGM) {      // context: the M rows in four ranges over the eight waves, combined in a fixed order
        const int ch = tid & 127, part = tid >> 7;
        const int mq = (M + 3) / 4, me = min(M, (part + 1) * mq);
        const float* al = sc_l + (ch / dh) * SCP;
        const float* vp = vals_l + ch;
        float c4[4] = {0.f, 0.f, 0.f, 0.f};
        int m = part * mq;
        for (; m + 4 <= me; m += 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) c4[k] = fmaf(al[m + k], vp[(m + k) * KP], c4[k]);
        }
        for (; m < me; ++m) c4[0] = fmaf(al[m], vp[m * KP], c4[0]);
        ((float*)red)[tid] = (c4[0] + c4[1]) + (c4[2] + c4[3]);
        __syncthreads();
      }
      if (wave < 2) {
        float cx;
        if constexpr (BIGM) {
          const float* rp = (const float*)red;
          cx = (rp[tid] + rp[128 + tid]) + (rp[256 + tid] + rp[384 + tid]);
        } else {
        const float* al = sc_l + (tid / dh) * SCP;
        const float* vp = vals_l + a_c;
        float c4[4] = {0.f, 0.f, 0.f, 0.f};                     // four interleaved partial sums (fixed order)
        int m = 0;
        for (; m + 4 <= M; m += 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) c4[k] = fmaf(al[m + k], vp[(m + k) * KP], c4[k]);
        }
        for (; m < M; ++m) c4[0] = fmaf(al[m], vp[m * KP], c4[0]);
        cx = (c4[0] + c4[1]) + (c4[2] + c4[3]);
        }
        if (!GREEDY) a.ctx_all[((size_t)t * B + ab) * D + a_c] = cx;
        const bool fin = t >= a_len;
        att_prev = fin ? att_prev : cx;
        if (!GREEDY) a.att_all[((size_t)(t + 1) * B + ab) * D + a_c] = att_prev;
        if (t + 1 < a.Tp) {
          const float xv = a.mask_in ? (att_prev / a.keep_in) * m_in : att_prev;
          const float x1 = __shfl_down(xv, 1, 64), x2 = __shfl_down(xv, 2, 64), x3 = __shfl_down(xv, 3, 64);
          if ((lane & 3) == 0)
            store16_sc1(xh_r, (unsigned)((((size_t)(t + 1) * B + ab) * Wd + E + a_c) * 4), make_float4(xv, x1, x2, x3));
        }
      }
      __syncthreads();   // as after the cell epilogue: poll the next operand only once our own att stores are issued
    }
    stamp(a.stamps, t, 7);
    if constexpr (GREEDY) {
      if (ids_l[32]) break;       // the same word in all 64 workgroups of the group (read in the ids gather of this step)
    }
  }
}

__global__ void sentinel_fill_kernel(ComicPersistRanges r, unsigned* sync, int n_zero, ComicPrologueExtra x) {
  const uint4 v = make_uint4(kSentinel, kSentinel, kSentinel, kSentinel);
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_zero) sync[i] = 0u;
#pragma unroll
  for (int k = 0; k < kPersistRanges; ++k) {
    const long n4 = r.n[k] >> 2;
    if (i < n4) {
      ((uint4*)r.p[k])[i] = v;
      return;
    }
    i -= n4;
  }
  // riders: the LSTM kernel's forward panel (the element order of pack_lstm_panels_kernel, mode 0) ...
  if (i < x.n_pack) {
    const int kk = (int)(i & 15), rr = (int)((i >> 4) & 15);
    const long blk = i >> 8;
    const int KB = (x.Wd + 15) >> 4;
    const int kb = (int)(blk % KB), tile = (int)(blk / KB);
    const int k = kb * 16 + kk, unit = tile * 4 + (rr >> 2);
    x.panel[i] = (k < x.Wd && unit < x.D) ? x.K[(size_t)k * 4 * x.D + (rr & 3) * x.D + unit] : 0.f;
    return;
  }
  i -= x.n_pack;
  // ... and W_o with padded rows
  if (i < x.n_pad) {
    const long row = i / x.Vp;
    const int c = (int)(i - row * x.Vp);
    x.wo_pad[i] = c < x.V ? x.W_o[row * x.V + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void persist_gate_kernel(const unsigned* err, float* loss_rows, float* map_loss,
                                                           ComicGateRanges r, float* step_flag, float* sticky) {
  const bool bad = err[0] != 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (step_flag) step_flag[0] = bad ? 1.f : 0.f;   // read by the gated optimiser entries (and summed by the all-reduce)
    if (bad && sticky) sticky[0] += 1.f;             // voided steps so far: the host reads it at its log points
  }
  if (!bad) return;                               // uniform over the launch: a healthy step ends here
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    loss_rows[0] = __int_as_float(0x7fc00000);
    map_loss[0] = __int_as_float(0x7fc00000);
  }
  const long stride = (long)gridDim.x * blockDim.x;
#pragma unroll 1
  for (int k = 0; k < 16; ++k)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < r.n[k]; i += stride) r.p[k][i] = 0.f;
}

constexpr int64_t kLdsMax = 160 * 1024;
constexpr int64_t kLdsMin = 96 * 1024;   // more than half of a CU's LDS: at most one workgroup per CU
int64_t lds_bytes(int M, int tied, bool wq_lds, bool greedy = false, bool bigm = false) {
  const int64_t cols = kD * 32 + (kD / 16) * 16;               // eight padded columns of a [D][D'] matrix
  const int64_t keys = bigm ? (int64_t)M * (kD / 4) * 4 : (int64_t)(tied ? 1 : 2) * M * kD * 4;
  const int64_t sc = bigm ? (4 * 256 + 512 + 3 * (kD / 4)) * 4 : 4 * 64 * 4;
  return keys + kWaves * 2 * 64 * 16 + kD * 4 + sc + (wq_lds ? cols : 0) + (greedy ? cols + 64 * 4 : 0);
}
// the channel-quarter form: memories that do not fit a CU whole (M > 64) and, among those that do, the ones that leave no
// room for the W_q columns beside them (tied M = 64: the query product read W_q from L2 every step, 8.7 us against 2.5)
inline bool big_m(int M, int tied) {
  if (!tied || M % 2 != 0 || M > 256) return false;
  return M > 64 || lds_bytes(M, tied, true) > kLdsMax;
}

template <int NX, bool WQ_LDS, bool GREEDY = false, bool BIGM = false>
int launch(const ComicPersistFwdArgs& a, int groups, int64_t lds, hipStream_t st) {
  auto kern = decoder_fwd_persistent_kernel<NX, WQ_LDS, GREEDY, BIGM>;
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax) != hipSuccess) {
      comic_set_error("persistent decoder: cannot reserve LDS");
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(groups * kGroupWgs), dim3(kThreads), (size_t)lds, st, a);
  COMIC_LAUNCH_CHECK("persistent decoder forward");
  return 0;
}

}  // namespace

bool comic_persist_fwd_supported(int B, int D, int E, int A, int M, int H, int Cv, int method, int context_layer,
                                 int tied) {
  if (context_layer || B < 1 || B > kMaxLaunches * kMaxGroups * kGroupRows) return false;
  if (D != kD || A != D || Cv != D || E < 16 || E % 16 != 0 || E > 512) return false;
  if (H != 4 && H != 8 && H != 16) return false;                  // a channel quarter holds 1, 2 or 4 whole heads
  if (M < 1) return false;
  if (method != 0 && method != 1) return false;
  // larger memories (Inception-V1 Mixed_4f, M = 196): a workgroup holds its channel quarter of the keys (BIGM)
  if (big_m(M, tied)) return lds_bytes(M, tied, false, false, true) <= kLdsMax;
  if (M > 64) return false;
  return lds_bytes(M, tied, false) <= kLdsMax;
}
bool comic_persist_fwd_bigm(int M, int tied) { return big_m(M, tied); }

// COMIC_DEC_STAMPS (comic_decoder_desc::flags; Python: COMIC_PERSIST_STAMPS=1): print the previous launch's mean phase times (a host synchronisation per launch: diagnostic).
// which = 0 forward loop, 1 backward loop; workgroup 0 stores eight 100 MHz clock values per step.
static thread_local bool g_stamps_on = false;
void comic_persist_set_stamps(bool on) { g_stamps_on = on; }
unsigned long long* comic_persist_stamps(int which, int Tp, hipStream_t st) {
  const bool on = g_stamps_on;
  static unsigned long long* dev[2] = {nullptr, nullptr};
  static int prev_tp[2] = {0, 0};
  static const char* names[2][8] = {
      {"att-wait+L", "L-epi", "y-wait+Q", "L-xh+q-wait", "scores", "prob", "ctx", "(masks)"},
      {"A'", "dq-sum", "dq-gather+dyq", "cell", "dg-stores", "sync", "dg-wait+I", "I-epi"}};
  if (!on) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;   // the read-back synchronises: never inside a graph capture
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  if (!dev[which] && hipMalloc((void**)&dev[which], 8 * 8 * 256) != hipSuccess) return nullptr;
  const int n = prev_tp[which];
  if (n > 1 && hipStreamSynchronize(st) == hipSuccess) {
    unsigned long long h[8 * 256];
    if (hipMemcpy(h, dev[which], sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      // steps are stored at index t; the backward loop runs t downwards, so "next step" is t - 1 there
      double sum[8] = {0};
      const int dir = which == 0 ? 1 : -1, first = which == 0 ? 0 : n - 1;
      for (int s = 0; s + 1 < n; ++s) {
        const int t = first + dir * s;
        for (int i = 0; i < 8; ++i) sum[i] += (double)(h[i < 7 ? t * 8 + i + 1 : (t + dir) * 8] - h[t * 8 + i]);
      }
      fprintf(stderr, "[persist stamps %s] per step (us):", which == 0 ? "fwd" : "bwd");
      for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.2f", names[which][i], sum[i] / (n - 1) / 100);
      const int last = first + dir * (n - 1);
      fprintf(stderr, " | step %.2f\n", (double)(h[last * 8] - h[first * 8]) / (n - 1) / 100);
    }
  }
  prev_tp[which] = Tp < 256 ? Tp : 0;
  return prev_tp[which] ? dev[which] : nullptr;
}

int comic_persist_fwd_launch(const ComicPersistFwdArgs& a_in, hipStream_t st) {
  ComicPersistFwdArgs a = a_in;
  a.stamps = a.grp0 == 0 ? comic_persist_stamps(0, a.Tp, st) : nullptr;
  const bool greedy = a.greedy != 0;
  const bool bigm = !greedy && big_m(a.M, a.tied);
  if (bigm && !a.statp) {
    comic_set_error("persistent decoder: M = %d needs tied keys / values, the statistics buffer, and is not a greedy loop", a.M);
    return 2;
  }
  const bool wq_lds = lds_bytes(a.M, a.tied, true, greedy, bigm) <= kLdsMax;
  int64_t lds = lds_bytes(a.M, a.tied, wq_lds, greedy, bigm);
  if (lds > kLdsMax) {
    comic_set_error("persistent decoder: %lld bytes of LDS needed", (long long)lds);
    return 2;
  }
  if (lds < kLdsMin) lds = kLdsMin;
  const int groups = a.n_groups;
  if (groups < 1 || groups > kMaxGroups || (a.grp0 + groups - 1) * kGroupRows >= a.B) {
    comic_set_error("persistent decoder: bad group range %d + %d at batch %d", a.grp0, groups, a.B);
    return 2;
  }
  const int nx = (a.E / 16 + kWaves - 1) / kWaves;                // x blocks per wave
  int rc = 2;
  if (greedy) {
    if (nx <= 1) rc = wq_lds ? launch<1, true, true>(a, groups, lds, st) : launch<1, false, true>(a, groups, lds, st);
    else if (nx == 2) rc = wq_lds ? launch<2, true, true>(a, groups, lds, st) : launch<2, false, true>(a, groups, lds, st);
    else if (nx <= 4) rc = wq_lds ? launch<4, true, true>(a, groups, lds, st) : launch<4, false, true>(a, groups, lds, st);
    else comic_set_error("persistent decoder: word size %d not supported", a.E);
    return rc;
  }
  if (bigm) {
    if (nx <= 1) rc = wq_lds ? launch<1, true, false, true>(a, groups, lds, st) : launch<1, false, false, true>(a, groups, lds, st);
    else if (nx == 2) rc = wq_lds ? launch<2, true, false, true>(a, groups, lds, st) : launch<2, false, false, true>(a, groups, lds, st);
    else if (nx <= 4) rc = wq_lds ? launch<4, true, false, true>(a, groups, lds, st) : launch<4, false, false, true>(a, groups, lds, st);
    else comic_set_error("persistent decoder: word size %d not supported", a.E);
    return rc;
  }
  if (nx <= 1) rc = wq_lds ? launch<1, true>(a, groups, lds, st) : launch<1, false>(a, groups, lds, st);
  else if (nx == 2) rc = wq_lds ? launch<2, true>(a, groups, lds, st) : launch<2, false>(a, groups, lds, st);
  else if (nx <= 4) rc = wq_lds ? launch<4, true>(a, groups, lds, st) : launch<4, false>(a, groups, lds, st);
  else comic_set_error("persistent decoder: word size %d not supported", a.E);
  return rc;
}

// greedy decode: as the training loop, plus the logit columns in LDS (V <= 512: radix / char vocabularies), one launch
bool comic_persist_greedy_supported(int B, int D, int E, int A, int M, int H, int Cv, int V, int method,
                                    int context_layer, int tied) {
  if (B > kMaxGroups * kGroupRows || V < 2 || V > 8 * kGroupWgs) return false;
  if (!comic_persist_fwd_supported(B, D, E, A, M, H, Cv, method, context_layer, tied)) return false;
  return lds_bytes(M, tied, false, true) <= kLdsMax;
}

__global__ void persist_check_greedy_kernel(const unsigned* err, int32_t* first_eos) {
  if (err[0] != 0u) first_eos[0] = -1;
}
int comic_persist_check_greedy(const unsigned* sync, int32_t* first_eos, hipStream_t st) {
  hipLaunchKernelGGL(persist_check_greedy_kernel, dim3(1), dim3(1), 0, st, sync, first_eos);
  COMIC_LAUNCH_CHECK("persistent decoder check");
  return 0;
}

int comic_persist_prepare(const ComicPersistRanges& r, unsigned* sync, int n_zero, hipStream_t st,
                          const ComicPrologueExtra* extra) {
  long n = 0;
  for (int k = 0; k < kPersistRanges; ++k) {
    COMIC_REQUIRE(r.n[k] % 4 == 0 && (r.n[k] == 0 || r.p[k]), "persistent decoder: bad hand-off range %d", k);
    n += r.n[k] / 4;
  }
  if (n_zero < kPersistSyncWords) n_zero = kPersistSyncWords;
  ComicPrologueExtra x{};
  if (extra) x = *extra;
  if (!x.panel) x.n_pack = 0;
  if (!x.wo_pad) x.n_pad = 0;
  n += x.n_pack + x.n_pad;
  if (n < n_zero) n = n_zero;
  hipLaunchKernelGGL(sentinel_fill_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, r, sync, n_zero, x);
  COMIC_LAUNCH_CHECK("persistent decoder prepare");
  return 0;
}

// The loops need every workgroup resident at once (one per CU): refuse devices with fewer CUs than the launch has
// workgroups (the per-step kernels run instead).  Cached per device.
bool comic_persist_fits_device(int B) {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    cus[dev] = n > 0 ? n : -1;
  }
  const int groups = (B + kGroupRows - 1) / kGroupRows;
  return cus[dev] >= (groups < kMaxGroups ? groups : kMaxGroups) * kGroupWgs;
}

int comic_persist_gate(const unsigned* sync, float* loss_rows, float* map_loss, const ComicGateRanges& r, float* step_flag,
                       float* sticky, hipStream_t st) {
  hipLaunchKernelGGL(persist_gate_kernel, dim3(128), dim3(256), 0, st, sync, loss_rows, map_loss, r, step_flag, sticky);
  COMIC_LAUNCH_CHECK("persistent decoder gate");
  return 0;
}
