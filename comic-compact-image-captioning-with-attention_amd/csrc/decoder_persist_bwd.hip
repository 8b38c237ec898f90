// Persistent backward time loop of the attention-LSTM decoder (training, fp32) for gfx950: the reverse of
// decoder_persist.hip's loop, one launch for all T' steps.
//
// Replaces the per-step launches of comic_decoder_train_step's backward loop (attn_bwd_kernel, lstm_grad_fused_kernel,
// input_grad_fused_kernel: 36 us of kernel time and three launch gaps per step) -- the gradient of
// MultiHeadAttentionWrapperV3.call (common/ops_rnn.py:660-755), of MultiHeadAddLN / MultiHeadDot (:531-565, :611-632),
// of BasicLSTMCell + DropoutWrapper (src/model_base.py:606-648) and of impute_finished (common/ops_rnn.py:183-243).
// tf.gradients builds the same chain in the reference; here it is written out by hand (oracle/decoder_ref.py
// train_backward is the CPU restatement the tests compare against).
//
// Same geometry and hand-off rule as the forward loop (decoder_persist_dev.h): groups of 16 batch rows x 64
// workgroups, one per CU; every handed-off buffer is time-major, sentinel-filled by the caller, written once with
// sc1 stores and read with validated sc1 loads.  Workgroup i of a group runs, per step t = T'-1 .. 0:
//   A' batch row (i % 16), memory rows m = (i / 16) mod 4 (one per wave): recomputes LayerNorm / tanh of its rows from
//      the resident keys and the saved q_t, d alpha of ALL rows (cheap), the probability backward, then tanh /
//      LayerNorm backward of its rows                      writes its quarter of d q_t (four partials per batch row)
//      d keys of its rows and the attention-parameter gradients accumulate in REGISTERS over all steps.
//   G  units [8i, 8i+8) x 16 rows:  d y = d y_logits + (sum of the four d q partials) * W_q^T (its W_q rows in LDS),
//      output-dropout and BasicLSTMCell backward; d c and the kept part of d h stay in registers
//                                                          writes d gates_t (also the operand of the d K GEMM)
//   I  operand features [16i, 16i+16) of the att and h thirds x 16 rows:  d gates_t * K^T (its 16 rows of K^T, 128 KB,
//      in registers, read in place; exact fp32 MFMA), input dropout       writes d att / d h of step t (read by A' / G of step t-1)
// The x third of d gates * K^T (the embedding gradient) does not feed the recurrence: one GEMM after the loop.
// What bounds a step: G and I each gather 128 KB per workgroup (3.6 us at the 70 GB/s a CU pulls, measured with
// tools/micro/gather_bench.hip); A' is VALU-bound (about 2 us).
#include <type_traits>

#include "decoder_persist_dev.h"

#include "decoder_math.h"

namespace {

// sum over the lph (2, 4, 8 or 16) consecutive lanes of a head: every lane of the head gets the total
__device__ __forceinline__ float head_total(float v, int lph) { return group_sum_dpp(v, lph); }

// demb[r][c] = (demb[r][c] / keep) * mask[r * ld + c]: the input dropout of the embedding third
__global__ void dropout_rows_kernel(float* __restrict__ x, const float* __restrict__ mask, float keep, long rows, int cols,
                                    int ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i % cols);
  x[i] = (x[i] / keep) * mask[r * ld + c];
}

__device__ __forceinline__ void stamp(unsigned long long* st, int t, int i) {
  if (st && blockIdx.x == 0 && threadIdx.x == 0) st[t * 8 + i] = __builtin_amdgcn_s_memrealtime();
}

// OWN (28 < M <= 64; the 8 x 8 map of 299-pixel inputs): the batch row's whole key matrix no longer fits beside the other
// residents, so a workgroup keeps only ITS memory rows m = quarter + 4 j (<= 16 of them, two per wave).  d alpha_d is then
// formed for the own rows only; the softmax backward needs the dot product sum_m alpha * d alpha over ALL rows of a head,
// which the four workgroups of a batch row assemble from their partial sums through one more sentinel-checked hand-off
// ([T'][B][4][16] floats).  The own rows' LayerNorm / tanh are recomputed in the last part instead of held in registers.
// MODE 2 (64 < M <= 256: the reference CLI's default map, Inception-V1 Mixed_4f, M = 196): the own-rows form with up to
// eight rows per wave.  Neither eight rows of d keys per wave (registers) nor the per-wave d v / d ln_g / d ln_b accumulators
// (48 KB of LDS beside 100 KB of keys) fit any more: d keys go to memory with no-return float atomics -- ONE writer per
// address, adds in step order, so the sums are reproducible -- into a zero-filled buffer, and the three parameter-gradient
// rows of a step are combined over the waves through the d q combine buffer into three registers per thread.
template <int MODE>
__global__ __launch_bounds__(kThreads) void decoder_bwd_persistent_kernel(ComicPersistBwdArgs a) {
  constexpr bool OWN = MODE != 0, BIG = MODE == 2;
  constexpr int NR = BIG ? 8 : 2;                               // own rows per wave (slots wave + 8 r)
  constexpr int SP = BIG ? 64 : 32;                             // slots per head row of the score buffers
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = kD, EPL = 8;
  const int M = a.M, H = a.H, E = a.E, EA = a.E + D, B = a.B, Tp = a.Tp, N4 = 4 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = a.grp0 + blockIdx.x / kGroupWgs, wi = blockIdx.x % kGroupWgs;   // group index over the whole batch
  const int row0 = grp * kGroupRows;
  Waiter wt{a.sync, false};

  // ---- LDS carve-up ---------------------------------------------------------------------------------------------
  float* keys_l = (float*)smem;                                // [M][D]   keys (= values) of the attention row (OWN: [16 slots][D], slot j = row quarter + 4 j)
  float* wq_l = keys_l + (BIG ? (M + 3) / 4 : OWN ? 16 : M) * D;   // [32 k16-blocks][8 units][4][4]  W_q rows of its units
  float4* red_i = (float4*)(wq_l + 8 * D);                     // [8 waves][64]  cross-wave combine of the G and I products
  float* red_q = (float*)(red_i + kWaves * 64);                // [8 waves][D]        d q combine (A')
  float* ss = red_q + kWaves * D;                              // [H][SP] scaled scores of own rows
  float* sd = ss + 16 * SP;                                    // [H][SP] d alpha_d, then d raw
  float* sa = sd + 16 * SP;                                    // [H][SP] alpha_d
  float* lnp_l = sa + 16 * SP;                                 // [3][D]  ln gamma | ln beta | v (read in A' only)
  float* pacc = lnp_l + 3 * D;                                 // [3][8 waves][D]  d v | d ln_g | d ln_b accumulators of the waves (not MODE 2)
  float* pd_l = pacc + (BIG ? 0 : 3 * kWaves * D);             // OWN: [16] this workgroup's partial dots, [16] the batch row's dots
  float* dot_l = pd_l + 16;

  // hand-off buffers in BLOCKED layouts: a k16-block of the 16 rows of a group is one contiguous KiB (16 rows x 64
  // bytes), which is exactly what one MFMA-operand load of a wave reads: whole 128-byte lines instead of 16 half lines
  //   dq_sum [t][group][k16-block 32][row 16][16]     dg_blk [t][group][k16-block 128][row 16][16]
  const int G = (B + kGroupRows - 1) / kGroupRows;             // groups of the whole batch
  const __amdgpu_buffer_rsrc_t dqp_r = make_rsrc(a.dq_part, (long)Tp * B * 4 * D * 4);     // [t][row][partial 4][D]
  const __amdgpu_buffer_rsrc_t dqs_r = make_rsrc(a.dq_sum, (long)Tp * G * 16 * D * 4);
  const __amdgpu_buffer_rsrc_t dg_r = make_rsrc(a.dg_blk, (long)Tp * G * 16 * N4 * 4);
  const int rows_here = min(kGroupRows, B - row0);              // rows of this group that exist
  const int r16c = min(lane & 15, rows_here - 1);               // MFMA-operand row of this lane (clamped: unused rows)
  const __amdgpu_buffer_rsrc_t ds_r = make_rsrc(a.dstate, (long)Tp * B * 2 * D * 4);
  const __amdgpu_buffer_rsrc_t dot_r = make_rsrc(a.dotp, OWN ? (long)Tp * B * 64 * 4 : 0);   // [t][row][quarter][16 heads]

  // ---- A' identity: batch row, owned memory rows ---------------------------------------------------------------------
  const int ab = row0 + (wi & 15), aq = wi >> 4;
  const bool a_live = ab < B;
  const int dh = D / H, lph = dh / EPL;                        // lanes per head (8 channels per lane)
  const int k0 = lane * EPL, head = k0 / dh;
  const int m_own = aq + 4 * wave;                             // this wave's memory row
  const bool has_own = a_live && m_own < M;
  const int m_own2 = aq + 4 * (wave + kWaves);                 // OWN: its second row (slot wave + 8)
  const bool has_own2 = OWN && a_live && m_own2 < M;           // (MODE 2 walks its rows by slot: row_ok)
  const int a_len = a_live ? a.lens[ab] : 0;
  {
    const int arow = a_live ? ab : 0;
    const float4* ks = (const float4*)(a.keys + (size_t)arow * M * D);
    if constexpr (OWN) {
      for (int i = tid; i < (BIG ? (M + 3) / 4 : 16) * (D / 4); i += kThreads) {
        const int m = aq + 4 * (i / (D / 4));
        ((float4*)keys_l)[i] = m < M ? ks[(size_t)m * (D / 4) + i % (D / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (tid < 32) pd_l[tid] = 0.f;
    } else {
      for (int i = tid; i < M * D / 4; i += kThreads) ((float4*)keys_l)[i] = ks[i];
    }
    for (int i = tid; i < 8 * D; i += kThreads) {              // W_q[8 wi + u][k] at ((k/16 * 8 + u) * 4 + (k%16)/4) * 4 + k%4
      const int u = i >> 9, k = i & (D - 1);
      wq_l[(((k >> 4) * 8 + u) * 4 + ((k & 15) >> 2)) * 4 + (k & 3)] = a.W_q[(size_t)(8 * wi + u) * D + k];
    }
    for (int i = tid; i < kWaves * D; i += kThreads) red_q[i] = 0.f;   // waves without a row never write theirs
    lnp_l[tid] = a.method == 0 ? a.ln_g[tid] : 0.f;
    lnp_l[D + tid] = a.method == 0 ? a.ln_b[tid] : 0.f;
    lnp_l[2 * D + tid] = a.method == 0 ? a.v[tid] : 0.f;
  }
  const float scale = a.method == 0 ? a.tau[0] : sqrtf((float)dh);
  const float inv_scale = 1.0f / scale;
  float datt_state[EPL], dk_acc[EPL], dk_acc2[EPL], dtau = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) datt_state[i] = dk_acc[i] = dk_acc2[i] = 0.f;
  if constexpr (!BIG)
    for (int i = tid; i < 3 * kWaves * D; i += kThreads) pacc[i] = 0.f;
  float pg_v = 0.f, pg_g = 0.f, pg_b = 0.f;                     // MODE 2: d v / d ln_g / d ln_b of channel tid, summed over waves and steps
  float* pa_v = pacc + wave * D + k0;                          // this lane's slices
  float* pa_g = pa_v + kWaves * D;
  float* pa_b = pa_g + kWaves * D;

  // ---- G identity: thread (row rl, k part) ; epilogue element (row rl, unit 8 wi + part) for part < 8 ------------------
  const int g_rl = tid >> 5, g_part = tid & 31;
  const int g_row = row0 + g_rl, g_d = 8 * wi + g_part;
  const bool g_elem = g_part < 8 && g_row < B;
  const int g_len = g_row < B ? a.lens[g_row] : 0;
  float g_dc = 0.f, g_dhk = 0.f;                               // d c state, kept part of d h state

  // ---- I identity: feature tile wi of the att | h thirds; this wave's sixteen k16-blocks of d gates --------------------
  const int r16 = lane & 15, kq = lane >> 4;
  constexpr int KBI = 4 * D / 16, NBI = KBI / kWaves;          // 128 blocks, 16 per wave
  float4 wreg[NBI];
  unsigned gb_off[NBI];
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int kb = 2 * (wave + kWaves * (i >> 1)) + (i & 1);
    gb_off[i] = (unsigned)kb * 1024u;
    wreg[i] = *(const float4*)(a.K + (size_t)(E + 16 * wi + r16) * N4 + 16 * kb + 4 * kq);   // K[feature][k], in place
  }
  __syncthreads();

  for (int t = Tp - 1; t >= 0; --t) {
    // ===================================================================== A': attention backward =====================
    stamp(a.stamps, t, 0);
    if (a_live) {
      const float live = t < a_len ? 1.f : 0.f;
      // saved forward values and masks of this step (plain loads: written by earlier launches)
      float qv[EPL];
      {
        const float4 q0 = *(const float4*)(a.q_all + ((size_t)t * B + ab) * D + k0);
        const float4 q1 = *(const float4*)(a.q_all + ((size_t)t * B + ab) * D + k0 + 4);
        qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
      }
      // d att of step t+1's operand: requested now, needed after the recomputation of the own row (which depends
      // only on saved forward values and covers the hand-off's latency)
      const unsigned dso = (unsigned)((((size_t)(t + 1) * B + ab) * 2 * D + k0) * 4);
      const unsigned off2[2] = {0u, 16u};
      float4 dv2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      if (t + 1 < Tp) {
        dv2[0] = load16_sc1(ds_r, dso);
        dv2[1] = load16_sc1(ds_r, dso + 16);
      }
      float gv[EPL], bv[EPL], vv[EPL];
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        gv[i] = lnp_l[k0 + i];
        bv[i] = lnp_l[D + k0 + i];
        vv[i] = lnp_l[2 * D + k0 + i];
      }
      if constexpr (OWN) {
        // the wave's own rows: slots wave + 8 r, rows m = quarter + 4 slot
        auto row_ok = [&](int r) { return a_live && aq + 4 * (wave + kWaves * r) < M; };
        // one own row's LayerNorm statistics of keys + q over the D channels (8 a lane)
        auto row_stats = [&](const float (&kk)[EPL], float& mean, float& rstd) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) s += kk[i] + qv[i];
          mean = wave_sum(s) / (float)D;
          float s2 = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float cc = kk[i] + qv[i] - mean;
            s2 += cc * cc;
          }
          rstd = 1.0f / sqrtf(wave_sum(s2) / (float)D + kLnEps);
        };
        // MODE 2 recomputes M / 4 * 512 tanh twice a step: 1 - 2 / (1 + e^(2x)) on v_exp_f32 / v_rcp_f32 (|error| < 2e-7)
        auto tanh_b = [](float x) {
          if constexpr (BIG) return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x));
          else return fast_tanh(x);
        };
        // (i) -- none: the forward's scaled scores s enter the backward only through d tau = -(sum_m ds_m s_m) / tau, and
        // with alpha = softmax(s), s_m = log alpha_m + c, sum_m ds_m = 0: the sum is sum_m ds_m log alpha_m, taken from the
        // SAVED probabilities below.  The rows' LayerNorm / tanh are computed once per step, in (iii).
        // d att state of step t: (finished at t+1 ? carried : 0) + d att of step t+1's operand
        if (t + 1 < Tp) {
          wait_written<2>(dv2, ds_r, dso, off2, 3u, wt);
          const float keepf = (t + 1 >= a_len) ? 1.f : 0.f;
          const float vin[EPL] = {dv2[0].x, dv2[0].y, dv2[0].z, dv2[0].w, dv2[1].x, dv2[1].y, dv2[1].z, dv2[1].w};
#pragma unroll
          for (int i = 0; i < EPL; ++i) datt_state[i] = datt_state[i] * keepf + vin[i];
        }
        float dcl[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) dcl[i] = datt_state[i] * live;
        // (ii) d alpha_d of the OWN rows: d ctx . values
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
          if (!row_ok(r)) break;
          const int slot = wave + kWaves * r;
          const float* kr = keys_l + slot * D + k0;
          const float4 ka = *(const float4*)kr, kb = *(const float4*)(kr + 4);
          float part = dcl[0] * ka.x;
          part = fmaf(dcl[1], ka.y, part); part = fmaf(dcl[2], ka.z, part); part = fmaf(dcl[3], ka.w, part);
          part = fmaf(dcl[4], kb.x, part); part = fmaf(dcl[5], kb.y, part); part = fmaf(dcl[6], kb.z, part);
          part = fmaf(dcl[7], kb.w, part);
          part = head_total(part, lph);
          if ((lane % lph) == 0)
            sd[head * SP + slot] = part + (a.dmap ? a.dmap[((size_t)t * B + ab) * M + aq + 4 * slot] : 0.f);
        }
        __syncthreads();
        // through the dropout; the softmax backward needs sum_m alpha * d alpha over ALL rows of a head: this workgroup's
        // part of it (lanes = own slots), heads wave and wave + 8
        float al_[2] = {0.f, 0.f}, da_[2] = {0.f, 0.f}, mk_[2] = {1.f, 1.f};
        const int mlane = aq + 4 * lane;
        const bool in = lane < SP && mlane < M;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int h = wave + kWaves * hh;
          if (h >= H) continue;
          const size_t go = (((size_t)t * B + ab) * H + h) * M;
          al_[hh] = in ? a.alpha_all[go + mlane] : 0.f;
          mk_[hh] = (in && a.mask_alpha) ? a.mask_alpha[go + mlane] : 1.f;
          float da = in ? sd[h * SP + lane] : 0.f;
          if (a.mask_alpha) da = (da / a.keep_alpha) * mk_[hh];
          da_[hh] = da;
          const float pd = wave_sum(al_[hh] * da);
          if (lane == 0) pd_l[h] = pd;
        }
        __syncthreads();
        const unsigned dbase = (unsigned)((((size_t)t * B + ab) * 64) * 4);
        if (wave == 0 && lane < 4)
          store16_sc1(dot_r, dbase + (unsigned)((aq * 16 + 4 * lane) * 4), *(const float4*)(pd_l + 4 * lane));
        __syncthreads();                                        // poll only after the own stores are on their way
        if (wave == 0) {                                        // lane l < 4: heads 4 l .. 4 l + 3 of the four quarters, in order
          const unsigned po = dbase + (unsigned)((4 * (lane & 3)) * 4);
          const unsigned qoff[4] = {0u, 64u, 128u, 192u};
          float4 pv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) pv[k] = load16_sc1(dot_r, po + qoff[k]);
          wait_written<4>(pv, dot_r, po, qoff, 15u, wt);
          if (lane < 4)
            *(float4*)(dot_l + 4 * lane) = make_float4((pv[0].x + pv[1].x) + (pv[2].x + pv[3].x), (pv[0].y + pv[1].y) + (pv[2].y + pv[3].y),
                                                       (pv[0].z + pv[1].z) + (pv[2].z + pv[3].z), (pv[0].w + pv[1].w) + (pv[2].w + pv[3].w));
        }
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int h = wave + kWaves * hh;
          if (h >= H) continue;
          const float dsv = al_[hh] * (da_[hh] - dot_l[h]);      // softmax backward (the launch requires prob == 0)
          if (in) {
            dtau -= dsv * (al_[hh] > 0.f ? logf(al_[hh]) : 0.f);
            sd[h * SP + lane] = dsv * inv_scale;
            sa[h * SP + lane] = a.mask_alpha ? (al_[hh] / a.keep_alpha) * mk_[hh] : al_[hh];
          }
        }
        __syncthreads();
        // (iii) own rows: through tanh / LayerNorm (or the dot product)
        float dqv[EPL], tv[EPL], tg[EPL], tb[EPL];                // d q and (MODE 2) this step's d v / d ln_g / d ln_b of the wave's rows
#pragma unroll
        for (int i = 0; i < EPL; ++i) dqv[i] = tv[i] = tg[i] = tb[i] = 0.f;
        auto lds_add8 = [](float* p, const float (&v)[EPL]) {
          float4 x = *(const float4*)p, y = *(const float4*)(p + 4);
          x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3]; y.x += v[4]; y.y += v[5]; y.z += v[6]; y.w += v[7];
          *(float4*)p = x;
          *(float4*)(p + 4) = y;
        };
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
          if (!row_ok(r)) break;
          const int slot = wave + kWaves * r;
          const float* kr = keys_l + slot * D + k0;
          const float4 ka = *(const float4*)kr, kb = *(const float4*)(kr + 4);
          const float kk[EPL] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
          const float draw = sd[head * SP + slot], adm = sa[head * SP + slot];
          float dkr[EPL];
          if (a.method == 0) {
            float mean, rstd;
            row_stats(kk, mean, rstd);
            float dxh[EPL], xh[EPL], s1 = 0.f, s2 = 0.f;
            {
              float uv[EPL], ug[EPL], ub[EPL];
#pragma unroll
              for (int i = 0; i < EPL; ++i) {
                const float zz = kk[i] + qv[i];
                const float inv = rstd * gv[i];
                const float th = tanh_b(zz * inv + (bv[i] - mean * inv));
                xh[i] = (zz - mean) * rstd;
                uv[i] = draw * th;
                const float dzh = draw * vv[i] * (1.f - th * th);
                ug[i] = dzh * xh[i];
                ub[i] = dzh;
                dxh[i] = dzh * gv[i];
                s1 += dxh[i];
                s2 += dxh[i] * xh[i];
              }
              if constexpr (BIG) {
#pragma unroll
                for (int i = 0; i < EPL; ++i) { tv[i] += uv[i]; tg[i] += ug[i]; tb[i] += ub[i]; }
              } else {       // the per-wave accumulators live in LDS, updated row by row
                lds_add8(pa_v, uv);
                lds_add8(pa_g, ug);
                lds_add8(pa_b, ub);
              }
            }
            const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
              const float dz = rstd * (dxh[i] - m1 - xh[i] * m2);
              dkr[i] = dz + adm * dcl[i];
              dqv[i] += dz;
            }
          } else {
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
              dkr[i] = draw * qv[i] + adm * dcl[i];
              dqv[i] += draw * kk[i];
            }
          }
          if constexpr (BIG) {       // one writer per address, adds in step order: reproducible sums (zero-filled by the caller)
            // through the wave's slice of the combine buffer, so that one atomic instruction covers 256 contiguous bytes
            // (lane = channel 64 i + lane) instead of four dwords in each of the row's sixteen lines
            float* tr = red_q + wave * D;
            *(float4*)(tr + k0) = make_float4(dkr[0], dkr[1], dkr[2], dkr[3]);
            *(float4*)(tr + k0 + 4) = make_float4(dkr[4], dkr[5], dkr[6], dkr[7]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float* dk = a.dkeys + ((size_t)ab * M + aq + 4 * slot) * D + lane;
#pragma unroll
            for (int i = 0; i < EPL; ++i) unsafeAtomicAdd(dk + 64 * i, tr[64 * i + lane]);
            __builtin_amdgcn_wave_barrier();
          } else {
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
              if (r == 0) dk_acc[i] += dkr[i];
              else dk_acc2[i] += dkr[i];
            }
          }
        }
        if constexpr (BIG) {       // the step's parameter-gradient rows over the eight waves, in wave order, into channel tid
          if (a.method == 0) {
            auto over_waves = [&](const float (&src)[EPL], float& acc) {
              *(float4*)(red_q + wave * D + k0) = make_float4(src[0], src[1], src[2], src[3]);
              *(float4*)(red_q + wave * D + k0 + 4) = make_float4(src[4], src[5], src[6], src[7]);
              __syncthreads();
              float sum = 0.f;
#pragma unroll
              for (int w = 0; w < kWaves; ++w) sum += red_q[w * D + tid];
              acc += sum;
              __syncthreads();
            };
            over_waves(tv, pg_v);
            over_waves(tg, pg_g);
            over_waves(tb, pg_b);
          }
        }
        *(float4*)(red_q + wave * D + k0) = make_float4(dqv[0], dqv[1], dqv[2], dqv[3]);
        *(float4*)(red_q + wave * D + k0 + 4) = make_float4(dqv[4], dqv[5], dqv[6], dqv[7]);
        __syncthreads();
        if (wave < 2) {                                           // this workgroup's partial of d q_t: 4 channels a thread
          float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w = 0; w < kWaves; ++w) {                      // fixed order: deterministic
            const float4 pp = *(const float4*)(red_q + w * D + 4 * tid);
            sum.x += pp.x; sum.y += pp.y; sum.z += pp.z; sum.w += pp.w;
          }
          store16_sc1(dqp_r, (unsigned)(((((size_t)t * B + ab) * 4 + aq) * D + 4 * tid) * 4), sum);
        }
        __syncthreads();   // polls of the next phase start after this workgroup's own stores are on their way
      } else {
      // (i) own row: LayerNorm / tanh recomputed, scaled scores of every head
      float th[EPL], xh[EPL], kro[EPL], rstd = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) th[i] = xh[i] = kro[i] = 0.f;
      if (has_own) {
        const float* kr = keys_l + m_own * D + k0;
        const float4 ka = *(const float4*)kr, kb = *(const float4*)(kr + 4);
        kro[0] = ka.x; kro[1] = ka.y; kro[2] = ka.z; kro[3] = ka.w; kro[4] = kb.x; kro[5] = kb.y; kro[6] = kb.z; kro[7] = kb.w;
        float part = 0.f;
        if (a.method == 0) {
          float z[EPL], s = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            z[i] = kro[i] + qv[i];
            s += z[i];
          }
          const float mean = wave_sum(s) / (float)D;
          float s2 = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float cc = z[i] - mean;
            s2 += cc * cc;
          }
          rstd = 1.0f / sqrtf(wave_sum(s2) / (float)D + kLnEps);
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float inv = rstd * gv[i];
            const float zh = z[i] * inv + (bv[i] - mean * inv);   // tf.nn.batch_normalization form
            th[i] = fast_tanh(zh);
            xh[i] = (z[i] - mean) * rstd;
            part += th[i] * vv[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < EPL; ++i) part += kro[i] * qv[i];
        }
        part = head_total(part, lph);
        if ((lane % lph) == 0) ss[head * 32 + m_own] = part * inv_scale;
      }
      // d att state of step t: (finished at t+1 ? carried : 0) + d att of step t+1's operand
      if (t + 1 < Tp) {
        wait_written<2>(dv2, ds_r, dso, off2, 3u, wt);
        const float keepf = (t + 1 >= a_len) ? 1.f : 0.f;
        const float vin[EPL] = {dv2[0].x, dv2[0].y, dv2[0].z, dv2[0].w, dv2[1].x, dv2[1].y, dv2[1].z, dv2[1].w};
#pragma unroll
        for (int i = 0; i < EPL; ++i) datt_state[i] = datt_state[i] * keepf + vin[i];
      }
      float dcl[EPL];
#pragma unroll
      for (int i = 0; i < EPL; ++i) dcl[i] = datt_state[i] * live;
      // (ii) d alpha_d of ALL memory rows (the probability backward needs whole rows of it): d ctx . values
      for (int m = wave; m < M; m += kWaves) {
        const float* kr = keys_l + m * D + k0;
        const float4 ka = *(const float4*)kr, kb = *(const float4*)(kr + 4);
        float part = dcl[0] * ka.x;
        part = fmaf(dcl[1], ka.y, part); part = fmaf(dcl[2], ka.z, part); part = fmaf(dcl[3], ka.w, part);
        part = fmaf(dcl[4], kb.x, part); part = fmaf(dcl[5], kb.y, part); part = fmaf(dcl[6], kb.z, part);
        part = fmaf(dcl[7], kb.w, part);
        part = head_total(part, lph);
        if ((lane % lph) == 0) sd[head * 32 + m] = part + (a.dmap ? a.dmap[((size_t)t * B + ab) * M + m] : 0.f);
      }
      __syncthreads();
      // through the dropout and the probability fn (a wave per head, lanes = memory rows); sd <- d raw, sa <- alpha_d
      for (int h = wave; h < H; h += kWaves) {
        const size_t go = (((size_t)t * B + ab) * H + h) * M;
        const bool in = lane < M;
        const float al = in ? a.alpha_all[go + lane] : 0.f;
        const float mk = (in && a.mask_alpha) ? a.mask_alpha[go + lane] : 1.f;
        float da = in ? sd[h * 32 + lane] : 0.f;
        if (a.mask_alpha) da = (da / a.keep_alpha) * mk;
        const bool own = in && (lane & 3) == aq;
        const float sown = own ? ss[h * 32 + lane] : 0.f;
        const float dot = wave_sum(al * da);                    // softmax backward (the launch requires prob == 0)
        const float dsv = al * (da - dot);
        if (own) dtau -= dsv * sown;
        if (in) {
          sd[h * 32 + lane] = dsv * inv_scale;
          sa[h * 32 + lane] = a.mask_alpha ? (al / a.keep_alpha) * mk : al;
        }
      }
      __syncthreads();
      // own row: through tanh / LayerNorm (or the dot product); d keys and parameter gradients stay in registers
      float dqv[EPL];
#pragma unroll
      for (int i = 0; i < EPL; ++i) dqv[i] = 0.f;
      if (has_own) {
        const float draw = sd[head * 32 + m_own], adm = sa[head * 32 + m_own];
        if (a.method == 0) {
          float dxh[EPL], s1 = 0.f, s2 = 0.f, av[EPL], ag[EPL], abb[EPL];
          *(float4*)av = *(const float4*)pa_v; *(float4*)(av + 4) = *(const float4*)(pa_v + 4);
          *(float4*)ag = *(const float4*)pa_g; *(float4*)(ag + 4) = *(const float4*)(pa_g + 4);
          *(float4*)abb = *(const float4*)pa_b; *(float4*)(abb + 4) = *(const float4*)(pa_b + 4);
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            av[i] += draw * th[i];
            const float dzh = draw * vv[i] * (1.f - th[i] * th[i]);
            ag[i] += dzh * xh[i];
            abb[i] += dzh;
            dxh[i] = dzh * gv[i];
            s1 += dxh[i];
            s2 += dxh[i] * xh[i];
          }
          *(float4*)pa_v = *(const float4*)av; *(float4*)(pa_v + 4) = *(const float4*)(av + 4);
          *(float4*)pa_g = *(const float4*)ag; *(float4*)(pa_g + 4) = *(const float4*)(ag + 4);
          *(float4*)pa_b = *(const float4*)abb; *(float4*)(pa_b + 4) = *(const float4*)(abb + 4);
          const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float dz = rstd * (dxh[i] - m1 - xh[i] * m2);
            dk_acc[i] += dz + adm * dcl[i];
            dqv[i] = dz;
          }
        } else {
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            dk_acc[i] += draw * qv[i] + adm * dcl[i];
            dqv[i] = draw * kro[i];
          }
        }
        *(float4*)(red_q + wave * D + k0) = make_float4(dqv[0], dqv[1], dqv[2], dqv[3]);
        *(float4*)(red_q + wave * D + k0 + 4) = make_float4(dqv[4], dqv[5], dqv[6], dqv[7]);
      }
      __syncthreads();
      if (wave < 2) {                                           // this workgroup's partial of d q_t: 4 channels a thread
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {                      // fixed order: deterministic
          const float4 pp = *(const float4*)(red_q + w * D + 4 * tid);
          sum.x += pp.x; sum.y += pp.y; sum.z += pp.z; sum.w += pp.w;
        }
        store16_sc1(dqp_r, (unsigned)(((((size_t)t * B + ab) * 4 + aq) * D + 4 * tid) * 4), sum);
      }
      __syncthreads();   // polls of the next phase start after this workgroup's own stores are on their way
      }
    }
    stamp(a.stamps, t, 1);
    // ===================================================================== G: query layer + LSTM cell backward =========
    {
      // saved forward values of the epilogue element (plain loads)
      float e_g[4] = {0.f, 0.f, 0.f, 0.f}, e_cp = 0.f, e_cn = 0.f, e_dy = 0.f, e_mk = 1.f;
      if (g_elem) {
        const size_t e = ((size_t)t * B + g_row) * D + g_d;
        const float* ga = a.gates_all + ((size_t)t * B + g_row) * N4;
        e_g[0] = ga[g_d]; e_g[1] = ga[D + g_d]; e_g[2] = ga[2 * D + g_d]; e_g[3] = ga[3 * D + g_d];
        e_cp = a.cs[e];
        e_cn = a.cnew_all[e];
        e_dy = a.dy_all[e];
        if (a.mask_out) e_mk = a.mask_out[e];
      }
      const int row = min(g_row, B - 1);
      // d h of step t+1's operand (h third), issued ahead of the d q gather: written one phase ago
      const unsigned hso = (unsigned)((((size_t)(t + 1) * B + row) * 2 * D + D + 8 * wi) * 4);
      const unsigned off2[2] = {0u, 16u};
      float4 hv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      if (t + 1 < Tp && g_part < 8) {
        hv[0] = load16_sc1(ds_r, hso);
        hv[1] = load16_sc1(ds_r, hso + 16);
      }
      // G1: this workgroup sums the four partials of ITS eight columns [8 wi, 8 wi + 8) of d q_t for the 16 rows (2 KB
      // gathered) and publishes them: the summed d q (operand of the d W_q GEMM after the loop, row-major) and the
      // blocked copy every workgroup of the group then gathers (32 KB instead of the 128 KB of all partials).
      if (wave < 2) {
        const int r = tid >> 3, hf = (tid >> 2) & 1, j = tid & 3;   // row, half of the eight columns, partial
        const int rr = min(row0 + r, B - 1);
        const unsigned po = (unsigned)(((((size_t)t * B + rr) * 4 + j) * D + 8 * wi + 4 * hf) * 4);
        const unsigned z1[1] = {0u};
        float4 pv[1] = {load16_sc1(dqp_r, po)};
        wait_written<1>(pv, dqp_r, po, z1, 1u, wt);
        float4 sm = pv[0];                                      // (p0 + p1) + (p2 + p3): fixed order
        sm.x += dpp_move<0xB1>(0.f, sm.x); sm.y += dpp_move<0xB1>(0.f, sm.y);
        sm.z += dpp_move<0xB1>(0.f, sm.z); sm.w += dpp_move<0xB1>(0.f, sm.w);
        sm.x += dpp_move<0x4E>(0.f, sm.x); sm.y += dpp_move<0x4E>(0.f, sm.y);
        sm.z += dpp_move<0x4E>(0.f, sm.z); sm.w += dpp_move<0x4E>(0.f, sm.w);
        if (j == 0 && row0 + r < B) {
          const int k = 8 * wi + 4 * hf;
          store16_sc1(dqs_r, (unsigned)((((((size_t)t * G + grp) * 32 + (k >> 4)) * 16 + r) * 16 + (k & 15)) * 4), sm);
          *(float4*)(a.dq_all + ((size_t)t * B + row0 + r) * D + k) = sm;
        }
      }
      __syncthreads();   // the gather below starts after this workgroup's own stores are on their way
      stamp(a.stamps, t, 2);
      // G2: d y_q = d q * W_q^T as an fp32 MFMA: lane (row r16, kq) of wave w loads k16-blocks 4 w .. 4 w + 3
      const unsigned qo = (unsigned)((((((size_t)t * G + grp) * 32 + 4 * wave) * 16 + r16c) * 16 + 4 * kq) * 4);
      const unsigned off4[4] = {0u, 1024u, 2048u, 3072u};
      float4 dq4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) dq4[i] = load16_sc1(dqs_r, qo + off4[i]);
      wait_written<4>(dq4, dqs_r, qo, off4, 15u, wt);
      {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 w4 = r16 < 8 ? *(const float4*)(wq_l + (((4 * wave + i) * 8 + r16) * 4 + kq) * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dq4[i].x, w4.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dq4[i].y, w4.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dq4[i].z, w4.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dq4[i].w, w4.w, acc, 0, 0, 0);
        }
        red_i[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);   // D[row 4 kq + i][unit r16]
      }
      __syncthreads();
      float dyq = 0.f;
      if (g_part < 8) {
        const float* rp = (const float*)red_i + (g_part + 16 * (g_rl >> 2)) * 4 + (g_rl & 3);
#pragma unroll
        for (int w = 0; w < kWaves; ++w) dyq += rp[w * 256];    // fixed order: deterministic
      }
      stamp(a.stamps, t, 3);
      // the cell backward of (row, unit)
      float dgv[4] = {0.f, 0.f, 0.f, 0.f};
      if (g_part < 8) {
        float vh = 0.f;
        if (t + 1 < Tp) {
          wait_written<2>(hv, ds_r, hso, off2, 3u, wt);
          const float h8[8] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w};
          vh = h8[0];
#pragma unroll
          for (int u = 1; u < 8; ++u) vh = g_part == u ? h8[u] : vh;
        }
        const float live = t < g_len ? 1.f : 0.f;
        const float dh_in = g_dhk + vh;
        const float si = e_g[0], tj = e_g[1], sf = e_g[2], so_ = e_g[3];
        const float tc = tanhf(e_cn);
        float dyv = e_dy + dyq;
        if (a.mask_out) dyv = (dyv / a.keep_out) * e_mk;
        const float dh2 = dh_in * live + dyv;
        float dc2 = g_dc * live;
        const float dso = dh2 * tc;
        dc2 += dh2 * so_ * (1.f - tc * tc);
        const float dsf = dc2 * e_cp, dsi = dc2 * tj, dtj = dc2 * si;
        dgv[0] = dsi * si * (1.f - si);
        dgv[1] = dtj * (1.f - tj * tj);
        dgv[2] = dsf * sf * (1.f - sf);
        dgv[3] = dso * so_ * (1.f - so_);
        g_dc = g_dc * (1.f - live) + dc2 * sf;
        g_dhk = dh_in * (1.f - live);
      }
      stamp(a.stamps, t, 4);
      // d gates: lanes part = 0..7 of a row hold 8 consecutive units of each gate: two 16-byte stores per gate
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float x1 = __shfl_down(dgv[g], 1, 64), x2 = __shfl_down(dgv[g], 2, 64), x3 = __shfl_down(dgv[g], 3, 64);
        if ((g_part == 0 || g_part == 4) && g_row < B) {
          const float4 v4 = make_float4(dgv[g], x1, x2, x3);
          const int k = g * D + g_d;
          store16_sc1(dg_r, (unsigned)((((((size_t)t * G + grp) * 128 + (k >> 4)) * 16 + g_rl) * 16 + (k & 15)) * 4), v4);
          *(float4*)(a.dg_all + ((size_t)t * B + g_row) * N4 + k) = v4;   // row-major copy for the GEMMs after the loop
        }
      }
      stamp(a.stamps, t, 5);
      __syncthreads();
    }
    stamp(a.stamps, t, 6);
    // ===================================================================== I: d gates * K^T (att and h thirds) ==========
    {
      // epilogue element of wave w < 4: row 4 kq + w, feature c; its input-dropout mask is fetched ahead
      const int c = 16 * wi + r16;                              // feature of the att | h thirds
      const int eb = row0 + 4 * kq + wave;
      float mk = 1.f;
      if (wave < 4 && a.mask_in && c < D && eb < B) mk = a.mask_in[((size_t)t * B + eb) * EA + E + c];
      const unsigned go = (unsigned)((((((size_t)t * G + grp) * 128) * 16 + r16c) * 16 + 4 * kq) * 4);
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      // the first four blocks come first (see the d q gather above), then the other twelve in one go
      float4 ga[NBI];
#pragma unroll
      for (int i = 0; i < 4; ++i) ga[i] = load16_sc1(dg_r, go + gb_off[i]);
      {
        float4 x4[4] = {ga[0], ga[1], ga[2], ga[3]};
        const unsigned o4[4] = {gb_off[0], gb_off[1], gb_off[2], gb_off[3]};
        wait_written<4>(x4, dg_r, go, o4, 15u, wt);
#pragma unroll
        for (int i = 0; i < 4; ++i) ga[i] = x4[i];
      }
#pragma unroll
      for (int i = 4; i < NBI; ++i) ga[i] = load16_sc1(dg_r, go + gb_off[i]);
#pragma unroll
      for (int c4 = 0; c4 < NBI; c4 += 4) {                     // validate and multiply four blocks at a time
        float4 x4[4] = {ga[c4], ga[c4 + 1], ga[c4 + 2], ga[c4 + 3]};
        const unsigned o4[4] = {gb_off[c4], gb_off[c4 + 1], gb_off[c4 + 2], gb_off[c4 + 3]};
        if (c4 > 0) wait_written<4>(x4, dg_r, go, o4, 15u, wt);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[i].x, wreg[c4 + i].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[i].y, wreg[c4 + i].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[i].z, wreg[c4 + i].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[i].w, wreg[c4 + i].w, acc, 0, 0, 0);
        }
      }
      red_i[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      __syncthreads();
      stamp(a.stamps, t, 7);
      if (wave < 4) {   // D[m][n]: lane (r16 = feature, kq) of every wave's partial holds rows 4 kq + i; wave w takes i = w
        const float* rp = (const float*)red_i + lane * 4 + wave;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += rp[w * 256];      // fixed order: deterministic
        if (a.mask_in && c < D) v = (v / a.keep_in) * mk;
        // four lanes r16 = 4j .. 4j+3 make one 16-byte store
        const float x1 = __shfl_down(v, 1, 64), x2 = __shfl_down(v, 2, 64), x3 = __shfl_down(v, 3, 64);
        if ((r16 & 3) == 0 && eb < B)
          store16_sc1(ds_r, (unsigned)((((size_t)t * B + eb) * 2 * D + c) * 4), make_float4(v, x1, x2, x3));
      }
      __syncthreads();
    }
  }

  // ---- after step 0: final states and the register accumulators ---------------------------------------------------------
  if (g_part < 8) {                                             // d h of the initial state = kept part + d h of step 0's operand
    const int row = min(g_row, B - 1);
    const unsigned so = (unsigned)((((size_t)row) * 2 * D + D + 8 * wi) * 4);
    const unsigned off2[2] = {0u, 16u};
    float4 hv[2] = {load16_sc1(ds_r, so), load16_sc1(ds_r, so + 16)};
    wait_written<2>(hv, ds_r, so, off2, 3u, wt);
    const float h8[8] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w};
    float vh = h8[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) vh = g_part == u ? h8[u] : vh;
    if (g_elem) {
      a.dc[(size_t)g_row * D + g_d] = g_dc;
      a.dh[(size_t)g_row * D + g_d] = g_dhk + vh;
    }
  }
  if (a_live) {
    if (has_own && !BIG) {
      float* dk = a.dkeys + ((size_t)ab * M + m_own) * D + k0;
      *(float4*)dk = make_float4(dk_acc[0], dk_acc[1], dk_acc[2], dk_acc[3]);
      *(float4*)(dk + 4) = make_float4(dk_acc[4], dk_acc[5], dk_acc[6], dk_acc[7]);
    }
    if (has_own2 && !BIG) {
      float* dk = a.dkeys + ((size_t)ab * M + m_own2) * D + k0;
      *(float4*)dk = make_float4(dk_acc2[0], dk_acc2[1], dk_acc2[2], dk_acc2[3]);
      *(float4*)(dk + 4) = make_float4(dk_acc2[4], dk_acc2[5], dk_acc2[6], dk_acc2[7]);
    }
    // parameter-gradient row of this workgroup: [d v | d ln_g | d ln_b | d tau], summed over its waves in fixed order
    float* pg = a.pgrad + ((size_t)ab * 4 + aq) * (3 * D + 1);
    __syncthreads();
    for (int k = 0; k < 3; ++k) {
      float sum = 0.f;
      if constexpr (BIG) {
        sum = k == 0 ? pg_v : k == 1 ? pg_g : pg_b;
      } else {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sum += pacc[(k * kWaves + w) * D + tid];   // fixed order: deterministic
      }
      pg[k * D + tid] = sum;
    }
    __syncthreads();
    const float dt = wave_sum(dtau);
    if (lane == 0) red_q[wave] = dt;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int w = 0; w < kWaves; ++w) s += red_q[w];
      pg[3 * D] = a.method == 0 ? s / a.tau[0] : 0.f;
    }
  }
}

inline int bwd_mode(int M) { return M <= 28 ? 0 : M <= 64 ? 1 : 2; }   // the whole key matrix fits up to M = 28
int64_t bwd_lds_bytes(int M, int mode = -1) {
  if (mode < 0) mode = bwd_mode(M);
  const int64_t key_rows = mode == 0 ? M : mode == 1 ? 16 : (M + 3) / 4;
  return key_rows * kD * 4 + 8 * kD * 4 + kWaves * 64 * 16 + kWaves * kD * 4 + 3 * 16 * (mode == 2 ? 64 : 32) * 4 + 3 * kD * 4 +
         (mode == 2 ? 0 : 3 * kWaves * kD * 4) + 32 * 4;
}

}  // namespace

bool comic_persist_bwd_supported(int B, int D, int E, int A, int M, int H, int Cv, int method, int prob,
                                 int context_layer, int tied) {
  if (!comic_persist_fwd_supported(B, D, E, A, M, H, Cv, method, context_layer, tied)) return false;
  if (!tied || prob != 0) return false;                        // d values folded into d keys; softmax probability
  if (M > 4 * 8 * kWaves || E % 16 != 0) return false;         // up to eight owned memory rows per wave (4 workgroups a batch row)
  return bwd_lds_bytes(M) <= 160 * 1024;
}

int comic_dropout_rows(float* x, const float* mask, float keep, long rows, int cols, int ld, hipStream_t st) {
  hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)cdiv64(rows * cols, 256)), dim3(256), 0, st, x, mask, keep, rows,
                     cols, ld);
  COMIC_LAUNCH_CHECK("dropout_rows");
  return 0;
}

int comic_persist_bwd_launch(const ComicPersistBwdArgs& a_in, hipStream_t st) {
  ComicPersistBwdArgs a = a_in;
  a.stamps = a.grp0 == 0 ? comic_persist_stamps(1, a.Tp, st) : nullptr;
  const int mode = a.own_rows ? 2 : bwd_mode(a.M);     // own_rows: the own-rows form (MODE 2) at any M (its d keys go to memory with atomics: zero-filled by the caller)
  if (mode != 0 && !a.dotp) {
    comic_set_error("persistent decoder backward: M = %d needs the dot-product hand-off buffer", a.M);
    return 2;
  }
  int64_t lds = bwd_lds_bytes(a.M, mode);
  if (lds < 96 * 1024) lds = 96 * 1024;                        // more than half of the LDS: one workgroup per CU
  static PerDeviceOnce attr_once__[3];
  bool& attr_set = attr_once__[mode].slot();   // hipFuncSetAttribute holds per device
  const void* kern = mode == 0 ? (const void*)decoder_bwd_persistent_kernel<0>
                   : mode == 1 ? (const void*)decoder_bwd_persistent_kernel<1> : (const void*)decoder_bwd_persistent_kernel<2>;
  if (!attr_set) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("persistent decoder backward: cannot reserve LDS");
      return 1;
    }
    attr_set = true;
  }
  const int groups = a.n_groups;
  if (groups < 1 || groups > kMaxGroups || (a.grp0 + groups - 1) * kGroupRows >= a.B) {
    comic_set_error("persistent decoder backward: bad group range %d + %d at batch %d", a.grp0, groups, a.B);
    return 2;
  }
  if (mode == 0) hipLaunchKernelGGL(decoder_bwd_persistent_kernel<0>, dim3(groups * kGroupWgs), dim3(kThreads), (size_t)lds, st, a);
  else if (mode == 1) hipLaunchKernelGGL(decoder_bwd_persistent_kernel<1>, dim3(groups * kGroupWgs), dim3(kThreads), (size_t)lds, st, a);
  else hipLaunchKernelGGL(decoder_bwd_persistent_kernel<2>, dim3(groups * kGroupWgs), dim3(kThreads), (size_t)lds, st, a);
  COMIC_LAUNCH_CHECK("persistent decoder backward");
  return 0;
}
