// The reference's other two recurrent cells (src/model_base.py:606-632, --rnn_name LN_LSTM | GRU) for gfx950: the
// element-wise / row-wise halves of a cell step, forward and backward.  The products around them ([x;att;h] K and its
// transposes) are the executor's GEMMs (decoder_exec.hip); these cells run on the per-step launch chain (the persistent
// loops and the fused / streaming step kernels are BasicLSTMCell's).
//
//   LN_LSTM  tf.contrib.rnn.LayerNormBasicLSTMCell(num_units) [TF-1.9 contrib/rnn/python/ops/rnn_cell.py]: layer_norm=True,
//            forget_bias 1, no bias on the product; i, j, f, o each through layers.layer_norm (scopes input / transform /
//            forget / output: moments over the units, nn.batch_normalization with epsilon 1e-12), new_c =
//            c*sigmoid(f+1) + sigmoid(i)*tanh(j) through layer_norm (scope state) -- the normalised value IS the new
//            cell state -- new_h = tanh(new_c)*sigmoid(o).
//   GRU      tf.contrib.rnn.GRUCell [TF-1.9 rnn_cell_impl.GRUCell.call]: [r,u] = sigmoid([x,h] W_g + b_g), cand =
//            tanh([x, r*h] W_c + b_c), new_h = u*h + (1-u)*cand; the state is h alone.
// Oracle: oracle/decoder_ref.py ln_lstm_cell / gru_cell / _cell_backward.
#include "decoder_math.h"

namespace {

constexpr int kCellThreads = 256;
constexpr int kCellMaxPer = 8;       // units per thread: D <= 2048

// sum over the workgroup (256 threads), result to every thread; `red` holds 4 floats per call site generation
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                                   // the previous use of `red` has been read
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// layers.layer_norm of one row held as z[k] = element threadIdx.x + 256 k: -> y (in place), xhat, rstd
template <int PER>
__device__ __forceinline__ float row_layer_norm(float (&z)[PER], float (&xhat)[PER], const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int D, float* red) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) s += (threadIdx.x + 256 * k < D) ? z[k] : 0.f;
  const float mean = block_sum(s, red) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const float c = (threadIdx.x + 256 * k < D) ? z[k] - mean : 0.f;
    q += c * c;
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)D + kLnEps);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    if (d < D) {
      xhat[k] = (z[k] - mean) * rstd;
      const float inv = rstd * gamma[d];
      z[k] = z[k] * inv + (beta[d] - mean * inv);      // nn.batch_normalization's form
    }
  }
  return rstd;
}

// backward of y = xhat*gamma + beta over one row: dy[k] -> dz[k] (in place); d gamma / d beta rows written
template <int PER>
__device__ __forceinline__ void row_layer_norm_bwd(float (&dy)[PER], const float (&xhat)[PER], float rstd,
                                                   const float* __restrict__ gamma, float* __restrict__ dgamma_row,
                                                   float* __restrict__ dbeta_row, int D, float* red) {
  float s1 = 0.f, s2 = 0.f;
  float dxh[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    dxh[k] = 0.f;
    if (d < D) {
      if (dgamma_row) {
        dgamma_row[d] = dy[k] * xhat[k];
        dbeta_row[d] = dy[k];
      }
      dxh[k] = dy[k] * gamma[d];
      s1 += dxh[k];
      s2 += dxh[k] * xhat[k];
    }
  }
  const float m1 = block_sum(s1, red) / (float)D;
  const float m2 = block_sum(s2, red) / (float)D;
#pragma unroll
  for (int k = 0; k < PER; ++k) dy[k] = rstd * (dxh[k] - m1 - xhat[k] * m2);
}

// ---------------------------------------------------------------------------------------------- LN_LSTM --------------
// One workgroup per row.  g: [S][B][4D] split-K partials of [x;att;h] K (no bias).  ln: 10 vectors gamma_i, beta_i,
// gamma_j, beta_j, gamma_f, beta_f, gamma_o, beta_o, gamma_c, beta_c at `ln_stride` floats.  Saved for the backward:
// gates_act [B][4D] (sigmoid i, tanh j, sigmoid f, sigmoid o), xhat [B][5D] (i, j, f, o, state), rstd [B][8].
// The state plumbing (finished rows keep their state, y with output dropout, h into the next operand row) is
// lstm_gates_fwd_kernel's.
template <int PER>
__global__ __launch_bounds__(kCellThreads) void ln_lstm_fwd_kernel(
    const float* __restrict__ g, int S, const float* __restrict__ ln, int ln_stride, const float* __restrict__ c_prev,
    const float* __restrict__ h_prev, float* __restrict__ gates_act, float* __restrict__ xhat_out,
    float* __restrict__ rstd_out, float* __restrict__ c_new, float* __restrict__ y, const float* __restrict__ mask_out,
    float keep_out, const int32_t* __restrict__ lens, int t, float* __restrict__ c_state, float* __restrict__ h_state, int B,
    int D, float* __restrict__ xh_next, int xh_ld) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float act[4][PER], xh[PER];
  float rs[5];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int d = threadIdx.x + 256 * k;
      float v = 0.f;
      if (d < D)
        for (int s = 0; s < S; ++s) v += g[((size_t)s * B + b) * 4 * D + q * D + d];
      act[q][k] = v;
    }
    rs[q] = row_layer_norm<PER>(act[q], xh, ln + (size_t)(2 * q) * ln_stride, ln + (size_t)(2 * q + 1) * ln_stride, D, red);
    if (xhat_out) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int d = threadIdx.x + 256 * k;
        if (d < D) xhat_out[(size_t)b * 5 * D + q * D + d] = xh[k];
      }
    }
  }
  float c2[PER], cp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    cp[k] = (c_prev && d < D) ? c_prev[(size_t)b * D + d] : 0.f;
    act[0][k] = sigmoidf_(act[0][k]);
    act[1][k] = tanhf(act[1][k]);
    act[2][k] = sigmoidf_(act[2][k] + 1.0f);
    act[3][k] = sigmoidf_(act[3][k]);
    c2[k] = cp[k] * act[2][k] + act[0][k] * act[1][k];
  }
  rs[4] = row_layer_norm<PER>(c2, xh, ln + (size_t)8 * ln_stride, ln + (size_t)9 * ln_stride, D, red);
  const bool fin = lens && (t >= lens[b]);
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    if (d >= D) continue;
    const size_t i = (size_t)b * D + d;
    if (xhat_out) xhat_out[(size_t)b * 5 * D + 4 * D + d] = xh[k];
    if (gates_act) {
      float* ga = gates_act + (size_t)b * 4 * D;
      ga[d] = act[0][k]; ga[D + d] = act[1][k]; ga[2 * D + d] = act[2][k]; ga[3 * D + d] = act[3][k];
    }
    const float h2 = tanhf(c2[k]) * act[3][k];
    if (c_new) c_new[i] = c2[k];
    if (y) y[i] = mask_out ? (h2 / keep_out) * mask_out[i] : h2;
    if (c_state) c_state[i] = fin ? cp[k] : c2[k];
    const float hs = fin ? (h_prev ? h_prev[i] : 0.f) : h2;
    if (h_state) h_state[i] = hs;
    if (xh_next) xh_next[(size_t)b * xh_ld + d] = hs;
  }
  if (rstd_out && threadIdx.x < 5) rstd_out[(size_t)b * 8 + threadIdx.x] = rs[threadIdx.x];
}

// Backward of the above for one row: dg [B][4D] = d(raw product), the row's LayerNorm parameter gradients
// pgrad [B][10][D] (order of `ln`), dc / dh state as lstm_gates_bwd_kernel.
template <int PER>
__global__ __launch_bounds__(kCellThreads) void ln_lstm_bwd_kernel(
    const float* __restrict__ gates_act, const float* __restrict__ xhat_in, const float* __restrict__ rstd_in,
    const float* __restrict__ ln, int ln_stride, const float* __restrict__ c_prev, const float* __restrict__ c_new,
    const float* __restrict__ dy, const float* __restrict__ dy_part, int S, const float* __restrict__ mask_out,
    float keep_out, const int32_t* __restrict__ lens, int t, float* __restrict__ dc_state, float* __restrict__ dh_state,
    float* __restrict__ dg, float* __restrict__ pgrad, int B, int D) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float live = (lens && t >= lens[b]) ? 0.f : 1.f;
  float rs[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) rs[q] = rstd_in[(size_t)b * 8 + q];
  float si[PER], tj[PER], sf[PER], so[PER], cp[PER], dso[PER], dcn[PER], xh[PER];
  float* pg = pgrad + (size_t)b * 10 * D;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    si[k] = tj[k] = sf[k] = so[k] = cp[k] = dso[k] = dcn[k] = xh[k] = 0.f;
    if (d >= D) continue;
    const size_t i = (size_t)b * D + d;
    const float* ga = gates_act + (size_t)b * 4 * D;
    si[k] = ga[d]; tj[k] = ga[D + d]; sf[k] = ga[2 * D + d]; so[k] = ga[3 * D + d];
    cp[k] = c_prev ? c_prev[i] : 0.f;
    const float tc = tanhf(c_new[i]);
    const float dcs = dc_state[i], dhs = dh_state[i];
    float dyv = dy ? dy[i] : 0.f;
    for (int s = 0; s < S; ++s) dyv += dy_part[(size_t)s * B * D + i];
    if (mask_out) dyv = (dyv / keep_out) * mask_out[i];
    const float dh2 = dhs * live + dyv;
    dso[k] = dh2 * tc;
    dcn[k] = dcs * live + dh2 * so[k] * (1.f - tc * tc);
    xh[k] = xhat_in[(size_t)b * 5 * D + 4 * D + d];
    dc_state[i] = dcs * (1.f - live);                 // + d c_prev below
    dh_state[i] = dhs * (1.f - live);
  }
  row_layer_norm_bwd<PER>(dcn, xh, rs[4], ln + (size_t)8 * ln_stride, pg + 8 * D, pg + 9 * D, D, red);   // dcn := d c_raw
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int d = threadIdx.x + 256 * k;
    if (d < D) dc_state[(size_t)b * D + d] += dcn[k] * sf[k];
  }
  float dp[PER];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int d = threadIdx.x + 256 * k;
      xh[k] = d < D ? xhat_in[(size_t)b * 5 * D + q * D + d] : 0.f;
      dp[k] = q == 0 ? dcn[k] * tj[k] * si[k] * (1.f - si[k])
            : q == 1 ? dcn[k] * si[k] * (1.f - tj[k] * tj[k])
            : q == 2 ? dcn[k] * cp[k] * sf[k] * (1.f - sf[k])
                     : dso[k] * so[k] * (1.f - so[k]);
    }
    row_layer_norm_bwd<PER>(dp, xh, rs[q], ln + (size_t)(2 * q) * ln_stride, pg + (2 * q) * D, pg + (2 * q + 1) * D, D, red);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int d = threadIdx.x + 256 * k;
      if (d < D) dg[(size_t)b * 4 * D + q * D + d] = dp[k];
    }
  }
}

// out[j*stride + d] (+)= in[j*D + d]
__global__ void scatter_vectors_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int D, int stride,
                                       float beta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  const int j = i / D, d = i % D;
  float* o = out + (size_t)j * stride + d;
  *o = (beta != 0.f ? beta * *o : 0.f) + in[i];
}

// -------------------------------------------------------------------------------------------------- GRU --------------
// gates: g1 [S][B][2D] partials of [x;att;h] W_g -> r, u (ru [B][ld_ru]: r at 0, u at D); xh2 = [x ; att ; r*h] (may be null)
__global__ void gru_gates_fwd_kernel(const float* __restrict__ g1, int S, const float* __restrict__ bias,
                                     const float* __restrict__ h_prev, const float* __restrict__ xh, int xh_ld,
                                     float* __restrict__ ru, int ld_ru, float* __restrict__ xh2, int xh2_ld, int B, int D,
                                     int EA) {
  const int W = EA + D;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, c = i % W;
  if (c < EA) {
    if (xh2) xh2[(size_t)b * xh2_ld + c] = xh[(size_t)b * xh_ld + c];
    return;
  }
  const int d = c - EA;
  float r = bias[d], u = bias[D + d];
  for (int s = 0; s < S; ++s) {
    const float* gs = g1 + ((size_t)s * B + b) * 2 * D;
    r += gs[d];
    u += gs[D + d];
  }
  r = sigmoidf_(r);
  u = sigmoidf_(u);
  ru[(size_t)b * ld_ru + d] = r;
  ru[(size_t)b * ld_ru + D + d] = u;
  if (xh2) xh2[(size_t)b * xh2_ld + c] = r * (h_prev ? h_prev[(size_t)b * D + d] : 0.f);   // (null: zero state, r*h = 0)
}

// candidate + new state: g2 [S][B][D] partials of [x;att;r*h] W_c
__global__ void gru_out_fwd_kernel(const float* __restrict__ g2, int S, const float* __restrict__ bias,
                                   const float* __restrict__ ru, int ld_ru, const float* __restrict__ h_prev,
                                   float* __restrict__ cand_out, int ld_cand, float* __restrict__ y,
                                   const float* __restrict__ mask_out, float keep_out, const int32_t* __restrict__ lens, int t,
                                   float* __restrict__ h_state, float* __restrict__ xh_next, int xh_ld, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i % D;
  float v = bias[d];
  for (int s = 0; s < S; ++s) v += g2[((size_t)s * B + b) * D + d];
  const float cand = tanhf(v);
  const float u = ru[(size_t)b * ld_ru + D + d];
  const float hp = h_prev ? h_prev[i] : 0.f;
  const float h2 = u * hp + (1.f - u) * cand;
  if (cand_out) cand_out[(size_t)b * ld_cand + d] = cand;
  if (y) y[i] = mask_out ? (h2 / keep_out) * mask_out[i] : h2;
  const bool fin = lens && (t >= lens[b]);
  const float hs = fin ? hp : h2;
  if (h_state) h_state[i] = hs;
  if (xh_next) xh_next[(size_t)b * xh_ld + d] = hs;
}

// backward, first half: d new_h -> d (candidate pre-activation) at dpre[b][2D + d], d (u pre-activation) at
// dpre[b][D + d]; dh_state := carried share + d new_h * u
__global__ void gru_bwd1_kernel(const float* __restrict__ dy, const float* __restrict__ dy_part, int S,
                                const float* __restrict__ mask_out, float keep_out, const int32_t* __restrict__ lens, int t,
                                float* __restrict__ dh_state, const float* __restrict__ ru, int ld_ru,
                                const float* __restrict__ cand, int ld_cand, const float* __restrict__ h_prev,
                                float* __restrict__ dpre, int ld_dpre, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i % D;
  const float live = (lens && t >= lens[b]) ? 0.f : 1.f;
  const float dhs = dh_state[i];
  float dyv = dy ? dy[i] : 0.f;
  for (int s = 0; s < S; ++s) dyv += dy_part[(size_t)s * B * D + i];
  if (mask_out) dyv = (dyv / keep_out) * mask_out[i];
  const float dh2 = dhs * live + dyv;
  const float u = ru[(size_t)b * ld_ru + D + d], c = cand[(size_t)b * ld_cand + d];
  const float hp = h_prev ? h_prev[i] : 0.f;
  dpre[(size_t)b * ld_dpre + 2 * D + d] = dh2 * (1.f - u) * (1.f - c * c);
  dpre[(size_t)b * ld_dpre + D + d] = dh2 * (hp - c) * u * (1.f - u);
  dh_state[i] = dhs * (1.f - live) + dh2 * u;
}

// backward, second half: dxh2 [B][Wd] = d cand_pre * W_c^T; its h third is d (r*h): d (r pre-activation) to
// dpre[b][d], and the third becomes d h's share d(r*h) * r (summed with the gates product by input_bwd_kernel)
__global__ void gru_bwd2_kernel(float* __restrict__ dxh2, int ld, const float* __restrict__ ru, int ld_ru,
                                const float* __restrict__ h_prev, float* __restrict__ dpre, int ld_dpre, int B, int D,
                                int EA) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i % D;
  float* p = dxh2 + (size_t)b * ld + EA + d;
  const float drh = *p;
  const float r = ru[(size_t)b * ld_ru + d];
  const float hp = h_prev ? h_prev[i] : 0.f;
  dpre[(size_t)b * ld_dpre + d] = drh * hp * r * (1.f - r);
  *p = drh * r;
}

template <int PER>
int launch_ln_fwd(const float* g, int S, const float* ln, int ln_stride, const float* c_prev, const float* h_prev,
                  float* gates_act, float* xhat, float* rstd, float* c_new, float* y, const float* mask_out, float keep_out,
                  const int32_t* lens, int t, float* c_state, float* h_state, int B, int D, float* xh_next, int xh_ld,
                  hipStream_t st) {
  hipLaunchKernelGGL((ln_lstm_fwd_kernel<PER>), dim3(B), dim3(kCellThreads), 0, st, g, S, ln, ln_stride, c_prev, h_prev,
                     gates_act, xhat, rstd, c_new, y, mask_out, keep_out, lens, t, c_state, h_state, B, D, xh_next, xh_ld);
  COMIC_LAUNCH_CHECK("ln_lstm_fwd");
  return 0;
}
template <int PER>
int launch_ln_bwd(const float* gates_act, const float* xhat, const float* rstd, const float* ln, int ln_stride,
                  const float* c_prev, const float* c_new, const float* dy, const float* dy_part, int S,
                  const float* mask_out, float keep_out, const int32_t* lens, int t, float* dc, float* dh, float* dg,
                  float* pgrad, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL((ln_lstm_bwd_kernel<PER>), dim3(B), dim3(kCellThreads), 0, st, gates_act, xhat, rstd, ln, ln_stride,
                     c_prev, c_new, dy, dy_part, S, mask_out, keep_out, lens, t, dc, dh, dg, pgrad, B, D);
  COMIC_LAUNCH_CHECK("ln_lstm_bwd");
  return 0;
}

}  // namespace

// stride between the ten LayerNorm vectors of comic_decoder_params::cell_ln
int comic_cell_ln_stride(int D) { return (D + 63) / 64 * 64; }

int comic_ln_lstm_fwd(const float* g, int S, const float* ln, const float* c_prev, const float* h_prev, float* gates_act,
                      float* xhat, float* rstd, float* c_new, float* y, const float* mask_out, float keep_out,
                      const int32_t* lens, int t, float* c_state, float* h_state, int B, int D, float* xh_next, int xh_ld,
                      hipStream_t st) {
  COMIC_REQUIRE(D > 0 && D <= 256 * kCellMaxPer, "LN_LSTM: rnn_size %d not supported (<= %d)", D, 256 * kCellMaxPer);
  const int ls = comic_cell_ln_stride(D);
#define COMIC_LN_FWD(P)                                                                                                   \
  return launch_ln_fwd<P>(g, S, ln, ls, c_prev, h_prev, gates_act, xhat, rstd, c_new, y, mask_out, keep_out, lens, t, c_state, \
                          h_state, B, D, xh_next, xh_ld, st)
  if (D <= 256) COMIC_LN_FWD(1);
  if (D <= 512) COMIC_LN_FWD(2);
  if (D <= 1024) COMIC_LN_FWD(4);
  COMIC_LN_FWD(8);
#undef COMIC_LN_FWD
}

int comic_ln_lstm_bwd(const float* gates_act, const float* xhat, const float* rstd, const float* ln, const float* c_prev,
                      const float* c_new, const float* dy, const float* dy_part, int S, const float* mask_out,
                      float keep_out, const int32_t* lens, int t, float* dc, float* dh, float* dg, float* pgrad, int B, int D,
                      hipStream_t st) {
  COMIC_REQUIRE(D > 0 && D <= 256 * kCellMaxPer, "LN_LSTM: rnn_size %d not supported (<= %d)", D, 256 * kCellMaxPer);
  const int ls = comic_cell_ln_stride(D);
#define COMIC_LN_BWD(P)                                                                                                  \
  return launch_ln_bwd<P>(gates_act, xhat, rstd, ln, ls, c_prev, c_new, dy, dy_part, S, mask_out, keep_out, lens, t, dc, dh, \
                          dg, pgrad, B, D, st)
  if (D <= 256) COMIC_LN_BWD(1);
  if (D <= 512) COMIC_LN_BWD(2);
  if (D <= 1024) COMIC_LN_BWD(4);
  COMIC_LN_BWD(8);
#undef COMIC_LN_BWD
}

// d cell_ln (+)= the column sums `sums` [10][D] of the per-row gradient rows
int comic_ln_lstm_scatter(const float* sums, float* cell_ln_grad, int D, float beta, hipStream_t st) {
  hipLaunchKernelGGL(scatter_vectors_kernel, dim3(cdiv(10 * D, 256)), dim3(256), 0, st, sums, cell_ln_grad, 10, D,
                     comic_cell_ln_stride(D), beta);
  COMIC_LAUNCH_CHECK("ln_lstm_scatter");
  return 0;
}

int comic_gru_gates_fwd(const float* g1, int S, const float* bias, const float* h_prev, const float* xh, int xh_ld, float* ru,
                        int ld_ru, float* xh2, int xh2_ld, int B, int D, int EA, hipStream_t st) {
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(cdiv(B * (EA + D), 256)), dim3(256), 0, st, g1, S, bias, h_prev, xh, xh_ld, ru,
                     ld_ru, xh2, xh2_ld, B, D, EA);
  COMIC_LAUNCH_CHECK("gru_gates_fwd");
  return 0;
}

int comic_gru_out_fwd(const float* g2, int S, const float* bias, const float* ru, int ld_ru, const float* h_prev,
                      float* cand, int ld_cand, float* y, const float* mask_out, float keep_out, const int32_t* lens, int t,
                      float* h_state, float* xh_next, int xh_ld, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(gru_out_fwd_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, g2, S, bias, ru, ld_ru, h_prev, cand,
                     ld_cand, y, mask_out, keep_out, lens, t, h_state, xh_next, xh_ld, B, D);
  COMIC_LAUNCH_CHECK("gru_out_fwd");
  return 0;
}

int comic_gru_bwd1(const float* dy, const float* dy_part, int S, const float* mask_out, float keep_out, const int32_t* lens,
                   int t, float* dh_state, const float* ru, int ld_ru, const float* cand, int ld_cand, const float* h_prev,
                   float* dpre, int ld_dpre, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(gru_bwd1_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, dy, dy_part, S, mask_out, keep_out, lens, t,
                     dh_state, ru, ld_ru, cand, ld_cand, h_prev, dpre, ld_dpre, B, D);
  COMIC_LAUNCH_CHECK("gru_bwd1");
  return 0;
}

int comic_gru_bwd2(float* dxh2, int ld, const float* ru, int ld_ru, const float* h_prev, float* dpre, int ld_dpre, int B,
                   int D, int EA, hipStream_t st) {
  hipLaunchKernelGGL(gru_bwd2_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, dxh2, ld, ru, ld_ru, h_prev, dpre, ld_dpre, B,
                     D, EA);
  COMIC_LAUNCH_CHECK("gru_bwd2");
  return 0;
}
