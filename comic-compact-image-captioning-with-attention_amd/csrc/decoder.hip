// Decoder step kernels for gfx950 (fp32): embeddings, dropout, fused LSTM gates,
// fused per-step multi-head attention (score + softmax + dropout + context) forward and
// backward, sequence cross-entropy, Adam, small reductions.
//
// Reference call-sites (see include/comic_hip.h for the per-function citations):
//   common/ops_rnn.py:531-565, :611-632, :660-755   attention mechanisms + wrapper step
//   common/ops.py:241-275                           layer_norm_activate (eps 1e-12)
//   src/model_base.py:557-594, :606-648, :325-417   embeddings, LSTM cell + dropout, losses
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "decoder_math.h"

namespace {

// ------------------------------------------------------------------ embeddings --------
__global__ void embed_fwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                 float* __restrict__ out, long total, int E, int V) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r = (int)(i / E), e = (int)(i % E);
  const int id = ids[r];
  out[i] = (id >= 0 && id < V) ? table[(size_t)id * E + e] : 0.f;
}

// d table[v, :] (+)= sum over the rows r with ids[r] == v of dout[r, :], deterministic and without atomics.
// One workgroup per (vocabulary row v, slice of EW table columns).  The id scan is parallel (ballot + ordered compaction
// into an LDS list); the SUM is parallel too: the list is cut into NS contiguous shares, row stream s (LPR lanes, VEC columns
// each, NC column steps) adds the rows of its share in ascending order with four loads in flight, and the NS partial sums are
// combined in stream order.  The shares depend only on (ids, rows), so two runs give the same bits.
// Why: radix vocabularies are frequency-sorted (datasets/preprocessing/prepro_base.py:186) -- the high digit 0 sits on most
// tokens -- and the one-sum-per-thread form this replaces walked such a row's 4 800 hits of a 224-hypothesis SCST step as ONE
// dependent chain per column: 905 us per call (profiles/r04_scst_kernel_stats.csv).
// A list that runs full is flushed into the streams' running sums and refilled: any number of hits per id, no special case.
template <bool SET, int NT, int LPR, int VEC, int NC>
__global__ __launch_bounds__(NT) void embed_bwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                                       float* __restrict__ dtable, int rows, int E) {
  constexpr int NS = NT / LPR;            // row streams of the workgroup
  constexpr int EW = LPR * VEC * NC;      // table columns of the workgroup
  constexpr int CAP = 4096;               // >= NT: the hits of one scan pass always fit an empty list
  static_assert(CAP >= NT && NT % 64 == 0 && NT % LPR == 0, "embed_bwd geometry");
  __shared__ int list[CAP];
  __shared__ int wcnt[NT / 64];
  __shared__ float part[NS][EW];
  const int v = blockIdx.x, e0 = blockIdx.y * EW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int stream = tid / LPR, sl = tid % LPR;
  float acc[NC][VEC];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[c][k] = 0.f;

  auto add_row = [&](int r) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = e0 + (c * LPR + sl) * VEC;
      if (col < E) {
        const float* src = dout + (size_t)r * E + col;
        if constexpr (VEC == 4) {
          const float4 x = *(const float4*)src;
          acc[c][0] += x.x; acc[c][1] += x.y; acc[c][2] += x.z; acc[c][3] += x.w;
        } else {
          acc[c][0] += src[0];
        }
      }
    }
  };
  auto flush = [&](int n) {               // stream s: entries [s*n/NS, (s+1)*n/NS) of the list, in order
    const int lo = (int)((long)stream * n / NS), hi = (int)((long)(stream + 1) * n / NS);
    int i = lo;
    for (; i + 4 <= hi; i += 4) {
      const int r0 = list[i], r1 = list[i + 1], r2 = list[i + 2], r3 = list[i + 3];
      if constexpr (VEC == 4 && NC == 1) {          // four rows in flight, added in list order
        const int col = e0 + sl * 4;
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0, x2 = x0, x3 = x0;
        if (col < E) {
          x0 = *(const float4*)(dout + (size_t)r0 * E + col);
          x1 = *(const float4*)(dout + (size_t)r1 * E + col);
          x2 = *(const float4*)(dout + (size_t)r2 * E + col);
          x3 = *(const float4*)(dout + (size_t)r3 * E + col);
        }
        acc[0][0] = (((acc[0][0] + x0.x) + x1.x) + x2.x) + x3.x;
        acc[0][1] = (((acc[0][1] + x0.y) + x1.y) + x2.y) + x3.y;
        acc[0][2] = (((acc[0][2] + x0.z) + x1.z) + x2.z) + x3.z;
        acc[0][3] = (((acc[0][3] + x0.w) + x1.w) + x2.w) + x3.w;
      } else {
        add_row(r0); add_row(r1); add_row(r2); add_row(r3);
      }
    }
    for (; i < hi; ++i) add_row(list[i]);
  };

  int fill = 0, n_total = 0;              // uniform: every thread derives them from the same LDS counts
  for (int base = 0; base < rows; base += NT) {
    const int r = base + tid;
    const bool hit = r < rows && ids[r] == v;
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, pass = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      const int c = wcnt[w];
      before += w < wave ? c : 0;
      pass += c;
    }
    if (fill + pass > CAP) {              // (uniform) the list cannot take this pass: fold it into the running sums first
      flush(fill);
      fill = 0;
      __syncthreads();
    }
    if (hit) list[fill + before + __popcll(bal & ((1ull << lane) - 1ull))] = r;
    fill += pass;
    n_total += pass;
    __syncthreads();
  }
  if (n_total == 0) {                     // SET: the table row is written, not accumulated into (no zero fill before the launch)
    if (SET)
      for (int e = tid; e < EW && e0 + e < E; e += NT) dtable[(size_t)v * E + e0 + e] = 0.f;
    return;
  }
  flush(fill);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k = 0; k < VEC; ++k) part[stream][(c * LPR + sl) * VEC + k] = acc[c][k];
  __syncthreads();
  for (int e = tid; e < EW && e0 + e < E; e += NT) {
    float sum = 0.f;
#pragma unroll 8
    for (int st = 0; st < NS; ++st) sum += part[st][e];
    if (SET) dtable[(size_t)v * E + e0 + e] = sum;
    else dtable[(size_t)v * E + e0 + e] += sum;
  }
}

// small vocabularies (radix / char tokens): 64 row streams of 16 lanes x float4 per 64-column slice, 1024 threads;
// otherwise (word vocabularies: tens of thousands of mostly empty rows) 4 streams of a wave each over 256 columns
template <bool SET>
int embed_bwd_launch(const int32_t* ids, const float* dout, float* dtable, int rows, int E, int V, hipStream_t st) {
  const bool vec = E % 4 == 0 && ((uintptr_t)dout & 15) == 0;
  const int slices = (E + 63) / 64;
  if (vec && (long)V * slices <= 4096)
    hipLaunchKernelGGL((embed_bwd_kernel<SET, 1024, 16, 4, 1>), dim3(V, slices), dim3(1024), 0, st, ids, dout, dtable, rows, E);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<SET, 256, 64, 1, 4>), dim3(V, (E + 255) / 256), dim3(256), 0, st, ids, dout, dtable,
                       rows, E);
  return 0;
}

// ------------------------------------------------------------------ dropout -----------
__global__ void dropout_apply_kernel(const float* __restrict__ x, const float* __restrict__ mask, float keep,
                                     float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = mask ? (x[i] / keep) * mask[i] : x[i];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void dropout_mask_kernel(float* __restrict__ mask, long n, float keep, uint64_t seed,
                                    const uint64_t* __restrict__ seed_dev, uint64_t offset) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (seed_dev) seed = seed_dev[0];  // graph replay: the seed lives in device memory
  const uint64_t r = splitmix64(splitmix64(seed) ^ (offset + (uint64_t)i));
  const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // [0,1)
  mask[i] = u < keep ? 1.f : 0.f;
}

// four consecutive masks (keep probability per segment) of one buffer in ONE launch; element i draws counter i,
// i.e. the same bits as four comic_dropout_mask_dev calls with cumulative offsets
struct MaskSegs {
  long end[4];
  float keep[4];
};
__global__ void dropout_masks4_kernel(float* __restrict__ mask, MaskSegs sg, const uint64_t* __restrict__ seed_dev) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sg.end[3]) return;
  const float keep = i < sg.end[0] ? sg.keep[0] : i < sg.end[1] ? sg.keep[1] : i < sg.end[2] ? sg.keep[2] : sg.keep[3];
  const uint64_t r = splitmix64(splitmix64(seed_dev[0]) ^ (uint64_t)i);
  const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
  mask[i] = u < keep ? 1.f : 0.f;
}

// out[0] = sum_{t,b} rows[t*B + b] * w[b*T + t]   (sequence_loss reduction: time-major rows, [B,T] weights)
__global__ __launch_bounds__(256) void weighted_sum_tb_kernel(const float* __restrict__ rows, const float* __restrict__ w,
                                                              int T, int B, float* __restrict__ out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < T * B; i += 256) acc += rows[i] * w[(i % B) * T + i / B];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------ LSTM gates --------
__global__ void lstm_gates_fwd_kernel(const float* __restrict__ g, const float* __restrict__ c_prev,
                                      const float* __restrict__ h_prev, float* __restrict__ gates_act,
                                      float* __restrict__ c_new, float* __restrict__ h_new, float* __restrict__ y,
                                      const float* __restrict__ mask_out, float keep_out,
                                      const int32_t* __restrict__ lens, int t, float* __restrict__ c_state,
                                      float* __restrict__ h_state, int B, int D, float* __restrict__ xh_next,
                                      int xh_ld, int S, const float* __restrict__ bias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i % D;
  const float* gr = g + (size_t)b * 4 * D;
  // S > 0: g is [S][B][4D] split-K partials of the pre-activations (bias not yet added)
  float gi = gr[d], gj = gr[D + d], gf = gr[2 * D + d], go = gr[3 * D + d];
  for (int s = 1; s < S; ++s) {
    const float* gs = gr + (size_t)s * B * 4 * D;
    gi += gs[d]; gj += gs[D + d]; gf += gs[2 * D + d]; go += gs[3 * D + d];
  }
  if (bias) { gi += bias[d]; gj += bias[D + d]; gf += bias[2 * D + d]; go += bias[3 * D + d]; }
  const float si = sigmoidf_(gi), tj = tanhf(gj);
  const float sf = sigmoidf_(gf + 1.0f), so = sigmoidf_(go);
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float c2 = cp * sf + si * tj;
  const float tc = tanhf(c2);
  const float h2 = tc * so;
  if (gates_act) {
    float* ga = gates_act + (size_t)b * 4 * D;
    ga[d] = si; ga[D + d] = tj; ga[2 * D + d] = sf; ga[3 * D + d] = so;
  }
  if (c_new) c_new[i] = c2;
  if (h_new) h_new[i] = h2;
  if (y) y[i] = mask_out ? (h2 / keep_out) * mask_out[i] : h2;
  const bool fin = lens && (t >= lens[b]);
  if (c_state) c_state[i] = fin ? cp : c2;
  const float hs = fin ? (h_prev ? h_prev[i] : 0.f) : h2;
  if (h_state) h_state[i] = hs;
  if (xh_next) xh_next[(size_t)b * xh_ld + d] = hs;  // recurrent part of the next step's GEMM operand
}

__global__ void lstm_gates_bwd_kernel(const float* __restrict__ gates_act, const float* __restrict__ c_prev,
                                      const float* __restrict__ c_new, const float* __restrict__ dy,
                                      const float* __restrict__ mask_out, float keep_out,
                                      const int32_t* __restrict__ lens, int t, float* __restrict__ dc_state,
                                      float* __restrict__ dh_state, float* __restrict__ dg, int B, int D,
                                      const float* __restrict__ dy_part, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i % D;
  const float live = (lens && t >= lens[b]) ? 0.f : 1.f;
  const float* ga = gates_act + (size_t)b * 4 * D;
  const float si = ga[d], tj = ga[D + d], sf = ga[2 * D + d], so = ga[3 * D + d];
  const float tc = tanhf(c_new[i]);
  const float dcs = dc_state[i], dhs = dh_state[i];
  float dyv = dy ? dy[i] : 0.f;
  for (int s = 0; s < S; ++s) dyv += dy_part[(size_t)s * B * D + i];  // + split-K partials of dq * W_q^T
  if (mask_out) dyv = (dyv / keep_out) * mask_out[i];
  const float dh2 = dhs * live + dyv;
  float dc2 = dcs * live;
  const float dso = dh2 * tc;
  dc2 += dh2 * so * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[i] : 0.f;
  const float dsf = dc2 * cp, dsi = dc2 * tj, dtj = dc2 * si;
  float* dgr = dg + (size_t)b * 4 * D;
  dgr[d] = dsi * si * (1.f - si);
  dgr[D + d] = dtj * (1.f - tj * tj);
  dgr[2 * D + d] = dsf * sf * (1.f - sf);
  dgr[3 * D + d] = dso * so * (1.f - so);
  dc_state[i] = dcs * (1.f - live) + dc2 * sf;
  dh_state[i] = dhs * (1.f - live);
}

// ------------------------------------------------------------------ attention ---------
// One workgroup (4 waves) per batch row.  A wave owns whole memory rows m (so the
// LayerNorm statistics and the per-head partial sums are wave shuffles); lane l holds the
// EPL = D/64 contiguous channels [l*EPL, (l+1)*EPL), all inside one head.
struct AttnArgs {
  comic_attn_desc d;
  const float *keys, *values, *q, *ln_g, *ln_b, *v, *tau, *alpha_in, *mask_alpha, *dctx, *dmap;
  float keep_alpha;
  float *alpha, *alpha_d, *ctx, *dq, *dkeys, *dvalues, *pgrad;
  // optional fusion of the wrapper's state plumbing (executor only; all NULL in the public op)
  const int32_t* lens;      // finished rule t >= lens[b]
  int t;
  const float* att_prev;    // [B,Cv] previous attention state (forward select)
  float* att_next;          // [B,Cv] att_next = fin ? att_prev : ctx
  float* xh_next;           // next step's LSTM input row: xh_next[b*xh_ld + c] = drop(att_next[c])
  int xh_ld;
  const float* mask_next;   // [B, mask_ld] input-dropout mask of the next step (offset applied by caller)
  int mask_ld;
  float keep_in;
  int q_parts;              // > 1: q is [q_parts][B][D] split-K partials; the reduced row goes to q_out
  int mem_div;              // > 1 (forward only): batch row b attends to memory row b / mem_div (beam search: the beams
                            // of an entry share its keys / values, which are then held once per entry)
  float* q_out;
  int pgrad_overwrite;      // backward: 1 = store this step's parameter-gradient row instead of adding to it
  const int32_t* stop;      // decode loops: see comic_stopped (common.h)
  int stop_t;
  // large memories (M >= kAttnSplitMinM, executor only): a batch row is served by gridDim.y workgroups; the scaled scores
  // (forward and backward) and d alpha_d (backward) of all memory rows pass through these [B][H][M] scratch arrays
  float* ws_s;
  float* ws_d;
  float* ws_parts;          // backward, gridDim.y > 2: [gridDim.y][B][4D + 4] partial d q | d v | d ln_g | d ln_b | d tau rows,
                            // summed in a fixed order by attn_bwd_reduce_kernel (no float atomics: bit-reproducible)
};

template <int EPL>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float* v) {
  if constexpr (EPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < EPL; i += 4) {
      const float4 t = *(const float4*)(p + i);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < EPL; ++i) v[i] = p[i];
  }
}

template <int EPL>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float* v) {
  if constexpr (EPL % 4 == 0) {
#pragma unroll
    for (int i = 0; i < EPL; i += 4) *(float4*)(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < EPL; ++i) p[i] = v[i];
  }
}

// sum over the `lph` consecutive lanes that share a head (lph is a power of two <= 64)
__device__ __forceinline__ float head_sum(float v, int lph) {
  if (lph == 1) return v;
  if (lph <= 16) return group_sum_dpp(v, lph);
  for (int o = 1; o < lph; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// raw (unscaled) per-head scores of one memory row; returns this lane's head partial
// already reduced over the head.  For add_LN also returns th = tanh(LN(z)), xh, rstd.
template <int EPL>
__device__ __forceinline__ float score_row(const AttnArgs& a, const float* kr, const float* qv, const float* gv,
                                           const float* bv, const float* vv, int lph, float* th, float* xh,
                                           float& rstd) {
  const int D = a.d.D;
  float part = 0.f;
  if (a.d.method == 0) {
    float z[EPL], s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      z[i] = kr[i] + qv[i];
      s += z[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const float c = z[i] - mean;
      s2 += c * c;
    }
    const float var = wave_sum(s2) / (float)D;
    rstd = 1.0f / sqrtf(var + kLnEps);
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      // tf.nn.batch_normalization form: x*inv + (beta - mean*inv), inv = rstd*gamma
      const float inv = rstd * gv[i];
      const float zh = z[i] * inv + (bv[i] - mean * inv);
      const float t = fast_tanh(zh);
      if (th) th[i] = t;
      if (xh) xh[i] = (z[i] - mean) * rstd;
      part += t * vv[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < EPL; ++i) part += kr[i] * qv[i];
  }
  return head_sum(part, lph);
}

constexpr int kAttnWaves = 16;   // waves per workgroup: each wave owns <= 2 memory rows at M=25 (8 waves measured: isolated step +0.15 ms, overlapped equal)
constexpr int kAttnThreads = kAttnWaves * 64;

// q row from its split-K partials [parts][B][D], added in slice order; the loads of four slices travel together
template <int EPL>
__device__ __forceinline__ void sum_q_parts(const float* q, int parts, size_t stride, float (&qv)[EPL]) {
  load_row<EPL>(q, qv);
  int s = 1;
  for (; s + 4 <= parts; s += 4) {
    float t[4][EPL];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_row<EPL>(q + (size_t)(s + u) * stride, t[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < EPL; ++i) qv[i] += t[u][i];
  }
  for (; s < parts; ++s) {
    float qs[EPL];
    load_row<EPL>(q + (size_t)s * stride, qs);
#pragma unroll
    for (int i = 0; i < EPL; ++i) qv[i] += qs[i];
  }
}

template <int EPL>
__global__ __launch_bounds__(kAttnThreads) void attn_fwd_kernel(AttnArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int B = a.d.B, M = a.d.M, D = a.d.D, H = a.d.H, Cv = a.d.Cv;
  float* sc = sm;  // [H][M]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  if (a.mem_div > 1) {
    // shared memories: workgroups x, x + 8, x + 16, ... are the beams of ONE entry -- consecutive workgroup ids go
    // round the 8 XCDs, so the beams of an entry run on one XCD at the same time and its keys / values leave the HBM once
    const int W = a.mem_div, wg = blockIdx.x;
    const int e = (wg / (8 * W)) * 8 + (wg & 7), w = (wg >> 3) % W;
    if (e * W >= B) return;
    b = e * W + w;
  }
  const int bm = a.mem_div > 1 ? b / a.mem_div : b;
  const int dh = D / H, lph = dh / EPL;
  const int k0 = lane * EPL, head = k0 / dh;
  float qv[EPL], gv[EPL], bv[EPL], vv[EPL];
  sum_q_parts<EPL>(a.q + (size_t)b * D + k0, a.q_parts, (size_t)a.d.B * D, qv);
  // small memories (M <= 28 rows of 2 x 1024 value channels: the 5 x 5 x 2048 map): the thread's two value channels of
  // every row are requested HERE, before the scores -- 205 KB per workgroup arrive at the CU's fill rate (4.4 us, phase
  // clocks) under the scoring and the probability function instead of behind them.  The two barriers of this path are raw
  // s_barriers behind an lgkmcnt wait: the fence of the library barrier would drain the loads (it waits vmcnt(0)).
  constexpr int kPre = 28;      // (32 rows spill at 8 elements per lane: 1024 threads leave 128 VGPRs)
  // (the tied 512-wide values of COMIC-256 measured no gain from the same treatment: 51 KB per row is not fill-bound)
  const bool fast = M <= kPre && Cv == 2 * kAttnThreads && ((Cv / H) & 1) == 0;
  float2 vpre[kPre];
  if (fast) {
    const float2* vp2 = (const float2*)(a.values + (size_t)bm * M * Cv) + tid;
#pragma unroll
    for (int m = 0; m < kPre; ++m)
      if (m < M) vpre[m] = vp2[(size_t)m * (Cv / 2)];
  }
  if (a.q_out && wave == 0) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) a.q_out[(size_t)b * D + k0 + i] = qv[i];
  }
  if (a.d.method == 0) {
    load_row<EPL>(a.ln_g + k0, gv);
    load_row<EPL>(a.ln_b + k0, bv);
    load_row<EPL>(a.v + k0, vv);
  }
  const float scale = a.d.method == 0 ? a.tau[0] : sqrtf((float)D / (float)H);
  for (int m = wave; m < M; m += kAttnWaves) {
    float kr[EPL], rstd;
    load_row<EPL>(a.keys + ((size_t)bm * M + m) * D + k0, kr);
    const float raw = score_row<EPL>(a, kr, qv, gv, bv, vv, lph, nullptr, nullptr, rstd);
    if ((lane % lph) == 0) sc[head * M + m] = raw / scale;
  }
  if (fast) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }
  // probability fn per head (wave per head), then dropout; sc <- alpha_d
  for (int h = wave; h < H; h += kAttnWaves) {
    float* row = sc + h * M;
    const size_t go = ((size_t)b * H + h) * M;
    if (a.d.prob == 0) {
      float mx = -INFINITY;
      for (int m = lane; m < M; m += 64) mx = fmaxf(mx, row[m]);
      mx = wave_max(mx);
      float s = 0.f;
      for (int m = lane; m < M; m += 64) s += expf(row[m] - mx);
      s = wave_sum(s);
      for (int m = lane; m < M; m += 64) {
        const float al = expf(row[m] - mx) / s;
        a.alpha[go + m] = al;
        const float ad = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go + m] : al;
        a.alpha_d[go + m] = ad;
        row[m] = ad;
      }
    } else {
      float s = 0.f;
      for (int m = lane; m < M; m += 64) s += sigmoidf_(row[m]);
      s = wave_sum(s);
      for (int m = lane; m < M; m += 64) {
        const float al = sigmoidf_(row[m]) / s;
        a.alpha[go + m] = al;
        const float ad = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go + m] : al;
        a.alpha_d[go + m] = ad;
        row[m] = ad;
      }
    }
  }
  if (fast) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }
  // context: ctx[c] = sum_m alpha_d[head(c)][m] * values[m][c]   (coalesced over c)
  const int dv = Cv / H;
  auto finish = [&](int c, float acc) {
    a.ctx[(size_t)b * Cv + c] = acc;
    if (a.att_next) {  // impute_finished select + next step's (dropped) LSTM input
      const bool fin = a.lens && a.t >= a.lens[b];
      const float av = fin ? a.att_prev[(size_t)b * Cv + c] : acc;
      a.att_next[(size_t)b * Cv + c] = av;
      if (a.xh_next) {
        float xv = av;
        if (a.mask_next) xv = (xv / a.keep_in) * a.mask_next[(size_t)b * a.mask_ld + c];
        a.xh_next[(size_t)b * a.xh_ld + c] = xv;
      }
    }
  };
  if (fast) {
    const float* al = sc + ((2 * tid) / dv) * M;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int m = 0; m < kPre; ++m)
      if (m < M) {
        a0 = fmaf(al[m], vpre[m].x, a0);
        a1 = fmaf(al[m], vpre[m].y, a1);
      }
    finish(2 * tid, a0);
    finish(2 * tid + 1, a1);
    return;
  }
  for (int c = tid; c < Cv; c += kAttnThreads) {
    const float* al = sc + (c / dv) * M;
    const float* vp = a.values + (size_t)bm * M * Cv + c;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc = fmaf(al[m], vp[(size_t)m * Cv], acc);
    finish(c, acc);
  }
}

// ---- large memories: gridDim.y workgroups per batch row ------------------------------------------------------------------
// At M = 196 (Inception-V1 Mixed_4f, the reference CLI's default feature map) a batch row's keys are 401 KB and the
// one-workgroup-per-row kernels above leave 192 of the 256 CUs idle at batch 64 (47 us forward, 59 us backward per time
// step).  Split form: kernel 1 -- each of the S workgroups of a row scores its share of the memory rows (LayerNorm +
// tanh: the VALU-heavy part) into a [B][H][M] scratch; kernel 2 -- every workgroup redoes the cheap probability fn of
// the whole row from the scratch (workgroup 0 of the row writes alpha / alpha_d) and forms the context of ITS share of
// the channels.  Same formulas; the context is summed over m in kCtxGroups interleaved chains instead of one.
constexpr int kAttnSplitMinM = 49;

template <int EPL>
__global__ __launch_bounds__(kAttnThreads) void attn_scores_kernel(AttnArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  const int M = a.d.M, D = a.d.D, H = a.d.H;
  const int b = blockIdx.x, S = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (M + S - 1) / S, m0 = (int)blockIdx.y * per, m1 = min(M, m0 + per);
  const int dh = D / H, lph = dh / EPL;
  const int k0 = lane * EPL, head = k0 / dh;
  float qv[EPL], gv[EPL], bv[EPL], vv[EPL];
  sum_q_parts<EPL>(a.q + (size_t)b * D + k0, a.q_parts, (size_t)a.d.B * D, qv);
  if (a.q_out && wave == 0 && blockIdx.y == 0) {
#pragma unroll
    for (int i = 0; i < EPL; ++i) a.q_out[(size_t)b * D + k0 + i] = qv[i];
  }
  if (a.d.method == 0) {
    load_row<EPL>(a.ln_g + k0, gv);
    load_row<EPL>(a.ln_b + k0, bv);
    load_row<EPL>(a.v + k0, vv);
  }
  const float scale = a.d.method == 0 ? a.tau[0] : sqrtf((float)D / (float)H);
  for (int m = m0 + wave; m < m1; m += kAttnWaves) {
    float kr[EPL], rstd;
    load_row<EPL>(a.keys + ((size_t)(a.mem_div > 1 ? b / a.mem_div : b) * M + m) * D + k0, kr);
    const float raw = score_row<EPL>(a, kr, qv, gv, bv, vv, lph, nullptr, nullptr, rstd);
    if ((lane % lph) == 0) a.ws_s[((size_t)b * H + head) * M + m] = raw / scale;
  }
}

__global__ __launch_bounds__(kAttnThreads) void attn_ctx_kernel(AttnArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = a.d.M, H = a.d.H, Cv = a.d.Cv;
  float* sc = sm;                 // [H][M]
  float* red = sm + H * M;        // [kAttnThreads] partial context sums
  const int b = blockIdx.x, S = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool writer = blockIdx.y == 0;
  for (int i = tid; i < H * M; i += kAttnThreads) sc[i] = a.ws_s[(size_t)b * H * M + i];
  __syncthreads();
  for (int h = wave; h < H; h += kAttnWaves) {
    float* row = sc + h * M;
    const size_t go = ((size_t)b * H + h) * M;
    if (a.d.prob == 0) {
      float mx = -INFINITY;
      for (int m = lane; m < M; m += 64) mx = fmaxf(mx, row[m]);
      mx = wave_max(mx);
      float s = 0.f;
      for (int m = lane; m < M; m += 64) s += expf(row[m] - mx);
      s = wave_sum(s);
      for (int m = lane; m < M; m += 64) {
        const float al = expf(row[m] - mx) / s;
        const float ad = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go + m] : al;
        if (writer) {
          a.alpha[go + m] = al;
          a.alpha_d[go + m] = ad;
        }
        row[m] = ad;
      }
    } else {
      float s = 0.f;
      for (int m = lane; m < M; m += 64) s += sigmoidf_(row[m]);
      s = wave_sum(s);
      for (int m = lane; m < M; m += 64) {
        const float al = sigmoidf_(row[m]) / s;
        const float ad = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go + m] : al;
        if (writer) {
          a.alpha[go + m] = al;
          a.alpha_d[go + m] = ad;
        }
        row[m] = ad;
      }
    }
  }
  __syncthreads();
  // context of the channels [c_lo, c_hi) of this workgroup: G chains over m per channel, then a fixed-order sum
  const int dv = Cv / H;
  const int CW = (Cv + S - 1) / S, c_lo = (int)blockIdx.y * CW, c_hi = min(Cv, c_lo + CW);
  const int G = max(1, kAttnThreads / CW);
  const int g = tid / CW, cl = tid - g * CW, c = c_lo + cl;
  float acc = 0.f;
  if (g < G && c < c_hi) {
    const float* al = sc + (c / dv) * M;
    const float* vp = a.values + (size_t)(a.mem_div > 1 ? b / a.mem_div : b) * M * Cv + c;
    for (int m = g; m < M; m += G) acc = fmaf(al[m], vp[(size_t)m * Cv], acc);
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < CW && c_lo + tid < c_hi) {
    float sum = 0.f;
    for (int k = 0; k < G; ++k) sum += red[k * CW + tid];
    const int cc = c_lo + tid;
    a.ctx[(size_t)b * Cv + cc] = sum;
    if (a.att_next) {  // impute_finished select + next step's (dropped) LSTM input
      const bool fin = a.lens && a.t >= a.lens[b];
      const float av = fin ? a.att_prev[(size_t)b * Cv + cc] : sum;
      a.att_next[(size_t)b * Cv + cc] = av;
      if (a.xh_next) {
        float xv = av;
        if (a.mask_next) xv = (xv / a.keep_in) * a.mask_next[(size_t)b * a.mask_ld + cc];
        a.xh_next[(size_t)b * a.xh_ld + cc] = xv;
      }
    }
  }
}

// Backward, kernel 1 of the split form: the recomputed scaled scores and d alpha_d of this workgroup's memory rows
// (one pass over its key rows; d values of its rows when values are not tied).
template <int EPL>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_scores_kernel(AttnArgs a) {
  const int M = a.d.M, D = a.d.D, H = a.d.H, Cv = a.d.Cv;
  const int b = blockIdx.x, S = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (M + S - 1) / S, m0 = (int)blockIdx.y * per, m1 = min(M, m0 + per);
  const int dh = D / H, lph = dh / EPL;
  const int k0 = lane * EPL, head = k0 / dh;
  float qv[EPL], gv[EPL], bv[EPL], vv[EPL];
  load_row<EPL>(a.q + (size_t)b * D + k0, qv);
  if (a.d.method == 0) {
    load_row<EPL>(a.ln_g + k0, gv);
    load_row<EPL>(a.ln_b + k0, bv);
    load_row<EPL>(a.v + k0, vv);
  }
  const float scale = a.d.method == 0 ? a.tau[0] : sqrtf((float)D / (float)H);
  const int dv = Cv / H, eplv = Cv / 64, lphv = dv / eplv;
  const int c0 = lane * eplv, headv = c0 / dv;
  const float* dctx = a.dctx + (size_t)b * Cv;
  const float live = (a.lens && a.t >= a.lens[b]) ? 0.f : 1.f;
  const bool tied = a.d.tied != 0;
  float dcl[EPL];
  if (tied) {
    load_row<EPL>(dctx + k0, dcl);
#pragma unroll
    for (int i = 0; i < EPL; ++i) dcl[i] *= live;
  }
  for (int m = m0 + wave; m < m1; m += kAttnWaves) {
    float kr[EPL], rstd;
    load_row<EPL>(a.keys + ((size_t)b * M + m) * D + k0, kr);
    const float raw = score_row<EPL>(a, kr, qv, gv, bv, vv, lph, nullptr, nullptr, rstd);
    if ((lane % lph) == 0) a.ws_s[((size_t)b * H + head) * M + m] = raw / scale;
    float part = 0.f;
    if (tied) {
#pragma unroll
      for (int i = 0; i < EPL; ++i) part = fmaf(dcl[i], kr[i], part);
    } else {
      const size_t go = ((size_t)b * H + headv) * M + m;
      const float al = a.alpha_in[go];
      const float ad = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go] : al;
      const float* vr = a.values + ((size_t)b * M + m) * Cv + c0;
      float* dvr = a.dvalues + ((size_t)b * M + m) * Cv + c0;
      for (int i = 0; i < eplv; ++i) {
        const float dc = dctx[c0 + i] * live;
        part = fmaf(dc, vr[i], part);
        dvr[i] += ad * dc;
      }
    }
    part = head_sum(part, lphv);
    if ((lane % lphv) == 0) a.ws_d[((size_t)b * H + headv) * M + m] = part + (a.dmap ? a.dmap[(size_t)b * M + m] : 0.f);
  }
}

template <int EPL>
__global__ __launch_bounds__(kAttnThreads) void attn_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int M = a.d.M, D = a.d.D, H = a.d.H, Cv = a.d.Cv;
  float* ss = sm;               // [H][M] scaled scores s
  float* sd = ss + H * M;       // [H][M] d alpha_d, then d raw
  float* sa = sd + H * M;                // [H][M] alpha_d (tied values)
  float* red = sa + H * M;               // [kAttnWaves][512] cross-wave reduction
  float* misc = red + kAttnWaves * 512;  // [kAttnWaves] d tau partials
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dh = D / H, lph = dh / EPL;
  const int k0 = lane * EPL, head = k0 / dh;
  float qv[EPL], gv[EPL], bv[EPL], vv[EPL];
  load_row<EPL>(a.q + (size_t)b * D + k0, qv);
  if (a.d.method == 0) {
    load_row<EPL>(a.ln_g + k0, gv);
    load_row<EPL>(a.ln_b + k0, bv);
    load_row<EPL>(a.v + k0, vv);
  }
  const float scale = a.d.method == 0 ? a.tau[0] : sqrtf((float)D / (float)H);
  // values-side lane mapping: EPLV contiguous channels per lane, all in one head
  const int dv = Cv / H, eplv = Cv / 64, lphv = dv / eplv;
  const int c0 = lane * eplv, headv = c0 / dv;
  const float* dctx = a.dctx + (size_t)b * Cv;
  // state-gradient form: only live rows pass d(att state) into the context (finished rows
  // kept their previous state)
  const float live = (a.lens && a.t >= a.lens[b]) ? 0.f : 1.f;
  // split mode (grid (B, 2), executor only): the two workgroups of a batch row each own half of the memory rows for
  // the expensive parts (score recomputation, LayerNorm / tanh backward, d keys); both compute the cheap d alpha of
  // ALL rows (the probability backward needs the full sum).  d q and the parameter-gradient row are then sums of two
  // contributions, added atomically into zero-filled buffers: with exactly two addends the result does not depend
  // on their order.
  // gridDim.y > 2 (large memories): phase A ran as attn_bwd_scores_kernel (a.ws_s / a.ws_d hold s and d alpha_d of ALL
  // rows); the S contributions to d q and to the parameter-gradient row are added atomically (for S > 2 the sum of
  // the partials depends on their arrival order in the last bits: the step is no longer bit-reproducible there)
  const bool split = gridDim.y >= 2;
  const int S_ = (int)gridDim.y, per_ = (M + S_ - 1) / S_;
  const int m0 = split ? (int)blockIdx.y * per_ : 0;
  const int m1 = split ? min(M, m0 + per_) : M;
  const bool pre = a.ws_s != nullptr;

  // ---- phase A: scores (recomputed) and d alpha_d; d values accumulation ----------------
  // tied values: the value row IS the key row (one load), and the alpha_d * dctx term of
  // d keys is folded into phase C's single read-modify-write (alpha_d parked in LDS)
  const bool tied = a.d.tied != 0;
  float dcl[EPL];   // this lane's slice of d ctx (tied layout: c0 == k0, eplv == EPL)
  if (tied) {
    load_row<EPL>(dctx + k0, dcl);
#pragma unroll
    for (int i = 0; i < EPL; ++i) dcl[i] *= live;
  }
  if (pre) {
    for (int i = tid; i < H * M; i += kAttnThreads) {
      const size_t go = (size_t)b * H * M + i;
      ss[i] = a.ws_s[go];
      sd[i] = a.ws_d[go];
      const float al = a.alpha_in[go];
      sa[i] = a.mask_alpha ? (al / a.keep_alpha) * a.mask_alpha[go] : al;
    }
  }
  for (int m = wave; m < (pre ? 0 : M); m += kAttnWaves) {
    float kr[EPL], rstd;
    load_row<EPL>(a.keys + ((size_t)b * M + m) * D + k0, kr);
    const size_t go = ((size_t)b * H + headv) * M + m;
    const float al = a.alpha_in[go];
    const float mk = a.mask_alpha ? a.mask_alpha[go] : 1.f;
    const bool own = m >= m0 && m < m1;          // wave-uniform
    if (own) {
      const float raw = score_row<EPL>(a, kr, qv, gv, bv, vv, lph, nullptr, nullptr, rstd);
      if ((lane % lph) == 0) ss[head * M + m] = raw / scale;
    }
    // alpha_d of this lane's value head
    const float ad = a.mask_alpha ? (al / a.keep_alpha) * mk : al;
    float part = 0.f;
    if (tied) {
#pragma unroll
      for (int i = 0; i < EPL; ++i) part = fmaf(dcl[i], kr[i], part);
      if ((lane % lphv) == 0) sa[headv * M + m] = ad;
    } else {
      const float* vr = a.values + ((size_t)b * M + m) * Cv + c0;
      float* dvr = a.dvalues + ((size_t)b * M + m) * Cv + c0;
      for (int i = 0; i < eplv; ++i) {
        const float dc = dctx[c0 + i] * live;
        part = fmaf(dc, vr[i], part);
        if (own) dvr[i] += ad * dc;
      }
    }
    part = head_sum(part, lphv);
    if ((lane % lphv) == 0) sd[headv * M + m] = part + (a.dmap ? a.dmap[(size_t)b * M + m] : 0.f);
  }
  __syncthreads();
  // ---- phase B: through dropout and the probability fn; sd <- d raw -----------------------
  float dtau = 0.f;
  for (int h = wave; h < H; h += kAttnWaves) {
    const size_t go = ((size_t)b * H + h) * M;
    float* srow = ss + h * M;
    float* drow = sd + h * M;
    if (a.d.prob == 0) {
      float dot = 0.f;
      for (int m = lane; m < M; m += 64) {
        float da = drow[m];
        if (a.mask_alpha) da = (da / a.keep_alpha) * a.mask_alpha[go + m];
        drow[m] = da;
        dot += a.alpha_in[go + m] * da;
      }
      dot = wave_sum(dot);
      for (int m = lane; m < M; m += 64) {
        const float ds = a.alpha_in[go + m] * (drow[m] - dot);
        if (m >= m0 && m < m1) dtau -= ds * srow[m];
        drow[m] = ds / scale;
      }
    } else {
      float S = 0.f, dsum = 0.f;
      for (int m = lane; m < M; m += 64) {
        float da = drow[m];
        if (a.mask_alpha) da = (da / a.keep_alpha) * a.mask_alpha[go + m];
        drow[m] = da;
        const float sg = sigmoidf_(srow[m]);
        S += sg;
        dsum += da * sg;
      }
      S = wave_sum(S);
      dsum = wave_sum(dsum);
      for (int m = lane; m < M; m += 64) {
        const float sg = sigmoidf_(srow[m]);
        const float dsg = drow[m] / S - dsum / (S * S);
        const float ds = dsg * sg * (1.f - sg);
        if (m >= m0 && m < m1) dtau -= ds * srow[m];
        drow[m] = ds / scale;
      }
    }
  }
  dtau = wave_sum(dtau);
  if (lane == 0) misc[wave] = dtau;
  __syncthreads();
  // ---- phase C: through tanh / LayerNorm (or the dot product) ------------------------------
  float dq_acc[EPL], dv_acc[EPL], dg_acc[EPL], db_acc[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) dq_acc[i] = dv_acc[i] = dg_acc[i] = db_acc[i] = 0.f;
  for (int m = m0 + wave; m < m1; m += kAttnWaves) {
    float kr[EPL], th[EPL], xh[EPL], rstd = 0.f;
    const float* kp = a.keys + ((size_t)b * M + m) * D + k0;
    float* dkp = a.dkeys + ((size_t)b * M + m) * D + k0;
    load_row<EPL>(kp, kr);
    const float draw = sd[head * M + m];
    if (a.d.method == 0) {
      score_row<EPL>(a, kr, qv, gv, bv, vv, lph, th, xh, rstd);
      float dxh[EPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        dv_acc[i] += draw * th[i];
        const float dzh = draw * vv[i] * (1.f - th[i] * th[i]);
        dg_acc[i] += dzh * xh[i];
        db_acc[i] += dzh;
        dxh[i] = dzh * gv[i];
        s1 += dxh[i];
        s2 += dxh[i] * xh[i];
      }
      const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
      const float adm = tied ? sa[head * M + m] : 0.f;
      float dko[EPL];
      load_row<EPL>(dkp, dko);
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const float dz = rstd * (dxh[i] - m1 - xh[i] * m2);
        dko[i] += dz + (tied ? adm * dcl[i] : 0.f);
        dq_acc[i] += dz;
      }
      store_row<EPL>(dkp, dko);
    } else {
      const float adm = tied ? sa[head * M + m] : 0.f;
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        dkp[i] += draw * qv[i] + (tied ? adm * dcl[i] : 0.f);
        dq_acc[i] += draw * kr[i];
      }
    }
  }
  // cross-wave reductions (4 waves), one array at a time through `red`
  // cross-wave reductions, one array at a time through `red` ([kAttnWaves][<=512] floats,
  // processed in 512-channel chunks so the LDS footprint is independent of D)
  auto reduce_store = [&](const float* acc, float* dst, int mode) {       // 0 store, 1 +=, 2 atomic add
    for (int cb = 0; cb < D; cb += 512) {
      const int cw = min(512, D - cb);
      __syncthreads();
      if (k0 >= cb && k0 < cb + cw) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) red[wave * 512 + (k0 - cb) + i] = acc[i];
      }
      __syncthreads();
      for (int k = tid; k < cw; k += kAttnThreads) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kAttnWaves; ++w) s += red[w * 512 + k];
        if (mode == 2)
          unsafeAtomicAdd(dst + cb + k, s);
        else if (mode == 1)
          dst[cb + k] += s;
        else
          dst[cb + k] = s;
      }
    }
  };
  if (a.ws_parts) {     // more than two workgroups per row: partial rows, reduced in a fixed order by the next launch
    float* pr = a.ws_parts + ((size_t)blockIdx.y * a.d.B + b) * (4 * D + 4);
    reduce_store(dq_acc, pr, 0);
    if (a.d.method == 0 && a.pgrad) {
      reduce_store(dv_acc, pr + D, 0);
      reduce_store(dg_acc, pr + 2 * D, 0);
      reduce_store(db_acc, pr + 3 * D, 0);
      if (tid == 0) {
        float dt = 0.f;
        for (int w = 0; w < kAttnWaves; ++w) dt += misc[w];
        pr[4 * D] = dt / a.tau[0];
      }
    }
    return;
  }
  reduce_store(dq_acc, a.dq + (size_t)b * D, split ? 2 : 0);
  if (a.d.method == 0 && a.pgrad) {
    float* pg = a.pgrad + (size_t)b * (3 * D + 1);
    // executor: one row per (step, batch row), summed once at the end (overwrite; split: two atomic contributions)
    const int mode = split ? 2 : (a.pgrad_overwrite == 0 ? 1 : 0);
    reduce_store(dv_acc, pg, mode);
    reduce_store(dg_acc, pg + D, mode);
    reduce_store(db_acc, pg + 2 * D, mode);
    if (tid == 0) {
      float dt = 0.f;
      for (int w = 0; w < kAttnWaves; ++w) dt += misc[w];
      if (mode == 2)
        unsafeAtomicAdd(pg + 3 * D, dt / a.tau[0]);
      else
        pg[3 * D] = (mode == 1 ? pg[3 * D] : 0.f) + dt / a.tau[0];
    }
  }
}

// d q [B][D] and the parameter-gradient row [B][3D + 1] of a step from the S partial rows of the split backward kernel
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const float* __restrict__ parts, float* __restrict__ dq,
                                                              float* __restrict__ pgrad, int B, int D, int S, int n) {
  const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  float v[8];
#pragma unroll
  for (int y = 0; y < 8; ++y) v[y] = y < S ? parts[((size_t)y * B + b) * (4 * D + 4) + k] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int y = 0; y < 8; ++y) s += v[y];                 // fixed order; the absent partials add exact zeros
  if (k < D)
    dq[(size_t)b * D + k] = s;
  else
    pgrad[(size_t)b * (3 * D + 1) + (k - D)] = s;
}

// ------------------------------------------------------------------ cross-entropy -----
// one row of the sequence loss: softmax cross-entropy of logits[row], d logits (row stride ld_dl >= V, the padding
// columns are written as zeros), arg-max id; `smax` / `sidx`: 256-entry LDS scratch of the calling workgroup
__device__ __forceinline__ void xent_row(int row, float* __restrict__ logits, const int32_t* __restrict__ targets_bt,
                                         const float* __restrict__ coef_bt, const float* __restrict__ wmask_bt,
                                         const int32_t* __restrict__ lens, float* __restrict__ loss_rows,
                                         float* __restrict__ dlogits, int32_t* __restrict__ ids_tb, int T, int B, int V,
                                         int ld_dl, float* smax, int* sidx) {
  const int t = row / B, b = row % B, tid = threadIdx.x;
  float* lg = logits + (size_t)row * V;
  float* dl = dlogits ? dlogits + (size_t)row * ld_dl : nullptr;
  if (dl)
    for (int v = V + tid; v < ld_dl; v += 256) dl[v] = 0.f;
  // T = stride of the [B,T] tables
  if (lens && t >= lens[b]) {  // impute_finished: zero outputs
    for (int v = tid; v < V; v += 256) {
      lg[v] = 0.f;
      if (dl) dl[v] = 0.f;
    }
    if (tid == 0) {
      if (loss_rows) loss_rows[row] = 0.f;
      if (ids_tb) ids_tb[row] = 0;
    }
    return;
  }
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float x = lg[v];
    if (x > mx) {
      mx = x;
      mi = v;
    }
  }
  smax[tid] = mx;
  sidx[tid] = mi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float o = smax[tid + s];
      const int oi = sidx[tid + s];
      if (o > smax[tid] || (o == smax[tid] && oi < sidx[tid])) {
        smax[tid] = o;
        sidx[tid] = oi;
      }
    }
    __syncthreads();
  }
  mx = smax[0];
  const int amax = sidx[0];
  __syncthreads();
  float s = 0.f;
  for (int v = tid; v < V; v += 256) s += expf(lg[v] - mx);
  smax[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) smax[tid] += smax[tid + st];
    __syncthreads();
  }
  const float sum = smax[0];
  const int tgt = targets_bt[(size_t)b * T + t];
  const float coef = coef_bt ? coef_bt[(size_t)b * T + t] : 0.f;
  if (dl)
    for (int v = tid; v < V; v += 256) dl[v] = (expf(lg[v] - mx) / sum - (v == tgt ? 1.f : 0.f)) * coef;
  if (tid == 0) {
    if (loss_rows) loss_rows[row] = (logf(sum) - (lg[tgt] - mx)) * (wmask_bt ? wmask_bt[(size_t)b * T + t] : 1.f);
    if (ids_tb) ids_tb[row] = amax;
  }
}

__global__ __launch_bounds__(256) void xent_kernel(float* __restrict__ logits, const int32_t* __restrict__ targets_bt,
                                                   const float* __restrict__ coef_bt,
                                                   const float* __restrict__ wmask_bt,
                                                   const int32_t* __restrict__ lens, float* __restrict__ loss_rows,
                                                   float* __restrict__ dlogits, int32_t* __restrict__ ids_tb, int T,
                                                   int B, int V, int ld_dl) {  // T = stride of the [B,T] tables
  __shared__ float smax[256];
  __shared__ int sidx[256];
  xent_row(blockIdx.x, logits, targets_bt, coef_bt, wmask_bt, lens, loss_rows, dlogits, ids_tb, T, B, V, ld_dl, smax, sidx);
}

// The sequence loss AND the attention-map loss (model_base.py:349-360: mean((1 - sum_h alpha)^2) * scale, with its
// gradient) in one launch: workgroups [0, rows) take one logits row each, the next `nparts` one 256-element piece of the
// map each; the piece sums are combined in piece order by whichever map workgroup arrives last (sc1 stores / loads and an
// agent-scope ticket: the hand-off form of gemm_group.hip), so map_loss has the bits of the two-launch form.
struct MapLossArgs {
  const float* hist;   // [Tp][B][H][M] attention probabilities
  float* dmap;         // [Tp][B][M] or null
  float* partial;      // [nparts] scratch
  float* map_loss;     // [1]
  unsigned* ticket;    // zero before the launch, zero after it
  int Tp, B, H, M, nparts;
  float scale;
};
__global__ __launch_bounds__(256) void xent_maploss_kernel(float* __restrict__ logits, const int32_t* __restrict__ targets_bt,
                                                           const float* __restrict__ coef_bt, const float* __restrict__ wmask_bt,
                                                           const int32_t* __restrict__ lens, float* __restrict__ loss_rows,
                                                           float* __restrict__ dlogits, int32_t* __restrict__ ids_tb, int T,
                                                           int B, int V, int ld_dl, int rows, MapLossArgs ma) {
  __shared__ float smax[256];
  __shared__ int sidx[256];
  if ((int)blockIdx.x < rows) {
    xent_row(blockIdx.x, logits, targets_bt, coef_bt, wmask_bt, lens, loss_rows, dlogits, ids_tb, T, B, V, ld_dl, smax, sidx);
    return;
  }
  const int piece = blockIdx.x - rows, tid = threadIdx.x;
  const long n = (long)ma.Tp * ma.B * ma.M;
  const long i = (long)piece * 256 + tid;
  float acc = 0.f;
  if (i < n) {
    const int m = (int)(i % ma.M);
    const long tb = i / ma.M;
    const float* p = ma.hist + (size_t)tb * ma.H * ma.M + m;
    float f = 0.f;
    for (int h = 0; h < ma.H; ++h) f += p[(size_t)h * ma.M];
    const float d = 1.0f - f;
    acc = d * d;
    if (ma.dmap) ma.dmap[i] = 2.0f * (f - 1.0f) / (float)n * ma.scale;
  }
  smax[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) smax[tid] += smax[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(ma.partial + piece, smax[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ma.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sidx[0] = (t == (unsigned)(ma.nparts - 1)) ? 1 : 0;
    if (sidx[0]) __hip_atomic_store(ma.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!sidx[0]) return;
  __syncthreads();
  float tot = 0.f;
  for (int k = tid; k < ma.nparts; k += 256) tot += __hip_atomic_load(ma.partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  smax[tid] = tot;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) smax[tid] += smax[tid + s];
    __syncthreads();
  }
  if (tid == 0) ma.map_loss[0] = smax[0] / (float)n * ma.scale;
}

// ------------------------------------------------------------------ optimiser etc. ----
__global__ void adam_tf_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps, float l2,
                               float gscale, const float* __restrict__ skip) {
  if (skip && skip[0] != 0.f) return;                     // a voided step (comic_decoder_params::status): no update at all
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float wi = w[i];
  const float ge = g[i] * gscale + l2 * wi;
  const float mi = m[i] + (ge - m[i]) * (1.f - b1);
  const float vi = v[i] + (ge * ge - v[i]) * (1.f - b2);
  m[i] = mi;
  v[i] = vi;
  w[i] = wi - (mi * lr_t) / (sqrtf(vi) + eps);
}

// ---- legacy encoder head (model_base.py:80-91): y = tanh(LayerNorm(x) * gamma + beta), eps 1e-12, over the last axis ----
// One workgroup per row; xhat is kept for the backward.  C <= 8 * 256.
__global__ __launch_bounds__(256) void ln_tanh_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ xhat, int C, float eps) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)b * C;
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float var = (red[4] + red[5] + red[6] + red[7]) / (float)C;     // biased variance (tf.nn.moments)
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = tid + 256 * i;
    if (c < C) {
      const float xh = (v[i] - mean) * rstd;
      xhat[(size_t)b * C + c] = xh;
      y[(size_t)b * C + c] = tanhf(xh * gamma[c] + beta[c]);
    }
  }
}
// rows of the parameter gradients: pg[b][c] = dy * (1 - y^2) * xhat, pb[b][c] = dy * (1 - y^2) (column sums follow)
__global__ void ln_tanh_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                        const float* __restrict__ xhat, float* __restrict__ pg, float* __restrict__ pb,
                                        long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = dy[i] * (1.f - y[i] * y[i]);
  pg[i] = t * xhat[i];
  pb[i] = t;
}

// tf.train.MomentumOptimizer (use_nesterov=False), ApplyMomentum [TF-1.9]: accum = momentum*accum + g; w -= lr*accum
__global__ void momentum_tf_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ accum, long n,
                                   float lr, float momentum, float l2, float gscale, const float* __restrict__ skip) {
  if (skip && skip[0] != 0.f) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float wi = w[i];
  const float ge = g[i] * gscale + l2 * wi;
  const float ai = accum[i] * momentum + ge;
  accum[i] = ai;
  w[i] = wi - lr * ai;
}

// ---- per-variable gradient clipping: slim.learning.clip_gradient_norms -> tf.clip_by_norm per tensor (model_base.py:394-401) ----
// chunk table: (segment, first element, elements, first chunk of the segment, chunks of the segment) per chunk of a variable.
// Both kernels sum in a fixed order (strided thread partials, wave tree, waves in order): the clipped step is reproducible.
struct ClipChunk { long seg, start, len, first, count; };
__device__ __forceinline__ float clip_block_sum(float s, float* red) {
  s = wave_sum(s);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void clip_partial_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                           const ClipChunk* __restrict__ chunks, float l2, float gscale,
                                                           float* __restrict__ partial) {
  __shared__ float red[4];
  const ClipChunk c = chunks[blockIdx.x];
  float s = 0.f;
  for (long i = threadIdx.x; i < c.len; i += 256) {
    const float ge = g[c.start + i] * gscale + l2 * w[c.start + i];
    s = fmaf(ge, ge, s);
  }
  s = clip_block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_apply_kernel(float* __restrict__ g, const float* __restrict__ w,
                                                         const ClipChunk* __restrict__ chunks, float l2, float gscale,
                                                         float clip, const float* __restrict__ partial,
                                                         const float* __restrict__ skip) {
  __shared__ float red[4];
  if (skip && skip[0] != 0.f) return;
  const ClipChunk c = chunks[blockIdx.x];
  float s = 0.f;
  for (long i = threadIdx.x; i < c.count; i += 256) s += partial[c.first + i];
  const float norm = sqrtf(clip_block_sum(s, red));
  if (!(norm > clip)) return;                       // t * clip / max(norm, clip) == t
  const float f = clip / norm, inv = 1.0f / gscale;
  for (long i = threadIdx.x; i < c.len; i += 256) {
    const float lw = l2 * w[c.start + i];
    const float ge = g[c.start + i] * gscale + lw;
    g[c.start + i] = (ge * f - lw) * inv;           // the optimiser's g*gscale + l2*w is then the clipped g_eff
  }
}

// out[j] = beta*out[j] + sum_i in[i*cols+j]; one thread per column (coalesced over j)
__global__ void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, float beta) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  // fixed summation tree (4 interleaved partial sums) -> deterministic, 4 loads in flight
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = 0;
  for (; i + 3 < rows; i += 4) {
    a0 += in[(size_t)i * cols + j];
    a1 += in[(size_t)(i + 1) * cols + j];
    a2 += in[(size_t)(i + 2) * cols + j];
    a3 += in[(size_t)(i + 3) * cols + j];
  }
  for (; i < rows; ++i) a0 += in[(size_t)i * cols + j];
  out[j] = (beta != 0.f ? beta * out[j] : 0.f) + ((a0 + a1) + (a2 + a3));
}

// two-stage column sum for tall inputs: stage 1 sums row chunks into partial[R][cols]
__global__ void colsum_part_kernel(const float* __restrict__ in, float* __restrict__ partial, int rows, int cols,
                                   int rows_per_chunk) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float a0 = 0.f, a1 = 0.f;
  int i = r0;
  for (; i + 1 < r1; i += 2) {
    a0 += in[(size_t)i * cols + j];
    a1 += in[(size_t)(i + 1) * cols + j];
  }
  if (i < r1) a0 += in[(size_t)i * cols + j];
  partial[(size_t)blockIdx.y * cols + j] = a0 + a1;
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

int attn_check(const comic_attn_desc* d) {
  COMIC_REQUIRE(d, "attn: null descriptor");
  COMIC_REQUIRE(d->B > 0 && d->M > 0 && d->H > 0, "attn: bad shape");
  COMIC_REQUIRE(d->D % 64 == 0 && d->D <= 1024, "attn: D must be a multiple of 64 and <= 1024 (got %d)", d->D);
  COMIC_REQUIRE(d->D % d->H == 0 && (d->D / d->H) % (d->D / 64) == 0 && 64 % d->H == 0,
                "attn: heads must divide 64 and D (D=%d H=%d)", d->D, d->H);
  COMIC_REQUIRE(d->Cv % 64 == 0 && d->Cv % d->H == 0 && (d->Cv / d->H) % (d->Cv / 64) == 0,
                "attn: value channels must be a multiple of 64 (Cv=%d H=%d)", d->Cv, d->H);
  COMIC_REQUIRE(!d->tied || d->Cv == d->D, "attn: tied values need Cv == D");
  COMIC_REQUIRE((size_t)d->H * d->M * 3 * 4 + (size_t)kAttnWaves * 512 * 4 + 256 <= 64 * 1024, "attn: H*M too large for LDS");
  return 0;
}

template <typename F>
int attn_dispatch(int D, F&& f) {
  switch (D / 64) {
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    default:
      comic_set_error("attn: unsupported D=%d (D/64 must be 1,2,4,8,16)", D);
      return 2;
  }
  return 0;
}

}  // namespace

extern "C" int comic_embed_fwd(const float* table, const int32_t* ids, float* out, int rows, int E, int V,
                               void* stream) {
  const long total = (long)rows * E;
  if (total == 0) return 0;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, table,
                     ids, out, total, E, V);
  COMIC_LAUNCH_CHECK("embed_fwd");
  return 0;
}

extern "C" int comic_embed_bwd(const int32_t* ids, const float* dout, float* dtable, int rows, int E, int V,
                               void* stream) {
  embed_bwd_launch<false>(ids, dout, dtable, rows, E, V, (hipStream_t)stream);
  COMIC_LAUNCH_CHECK("embed_bwd");
  return 0;
}
// dtable = (not +=) the scattered sums: every table row is written, so no zero fill precedes it
int comic_embed_bwd_set(const int32_t* ids, const float* dout, float* dtable, int rows, int E, int V, hipStream_t st) {
  embed_bwd_launch<true>(ids, dout, dtable, rows, E, V, st);
  COMIC_LAUNCH_CHECK("embed_bwd_set");
  return 0;
}

extern "C" int comic_dropout_apply(const float* x, const float* mask, float keep, float* y, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, x, mask,
                     keep, y, (long)n);
  COMIC_LAUNCH_CHECK("dropout_apply");
  return 0;
}

extern "C" int comic_dropout_mask(float* mask, int64_t n, float keep, uint64_t seed, uint64_t offset, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, mask,
                     (long)n, keep, seed, (const uint64_t*)nullptr, offset);
  COMIC_LAUNCH_CHECK("dropout_mask");
  return 0;
}

extern "C" int comic_dropout_mask_dev(float* mask, int64_t n, float keep, const uint64_t* seed_dev, uint64_t offset,
                                      void* stream) {
  if (n == 0) return 0;
  COMIC_REQUIRE(seed_dev, "dropout_mask_dev: null seed");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, mask,
                     (long)n, keep, (uint64_t)0, seed_dev, offset);
  COMIC_LAUNCH_CHECK("dropout_mask_dev");
  return 0;
}

extern "C" int comic_dropout_masks4_dev(float* mask, const int64_t* n4, const float* keep4, const uint64_t* seed_dev,
                                        void* stream) {
  COMIC_REQUIRE(mask && n4 && keep4 && seed_dev, "dropout_masks4_dev: null argument");
  MaskSegs sg;
  long end = 0;
  for (int i = 0; i < 4; ++i) {
    COMIC_REQUIRE(n4[i] >= 0, "dropout_masks4_dev: negative length");
    end += n4[i];
    sg.end[i] = end;
    sg.keep[i] = keep4[i];
  }
  if (end == 0) return 0;
  hipLaunchKernelGGL(dropout_masks4_kernel, dim3((unsigned)cdiv64(end, 256)), dim3(256), 0, (hipStream_t)stream, mask, sg,
                     seed_dev);
  COMIC_LAUNCH_CHECK("dropout_masks4_dev");
  return 0;
}

extern "C" int comic_weighted_sum_tb(const float* rows_tb, const float* w_bt, int T, int B, float* out, void* stream) {
  COMIC_REQUIRE(rows_tb && w_bt && out && T > 0 && B > 0, "weighted_sum_tb: bad argument");
  hipLaunchKernelGGL(weighted_sum_tb_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows_tb, w_bt, T, B, out);
  COMIC_LAUNCH_CHECK("weighted_sum_tb");
  return 0;
}

extern "C" int comic_lstm_gates_fwd(const float* g, const float* c_prev, const float* h_prev, float* gates_act,
                                    float* c_new, float* h_new, float* y, const float* mask_out, float keep_out,
                                    const int32_t* lens, int t, float* c_state, float* h_state, int B, int D,
                                    void* stream) {
  COMIC_REQUIRE(g, "lstm_gates_fwd: null input");
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, g, c_prev,
                     h_prev, gates_act, c_new, h_new, y, mask_out, keep_out, lens, t, c_state, h_state, B, D,
                     (float*)nullptr, 0, 1, (const float*)nullptr);
  COMIC_LAUNCH_CHECK("lstm_gates_fwd");
  return 0;
}

// executor-internal: also scatters the carried h into the next step's [x;att;h] operand row
int comic_lstm_gates_fwd_ex(const float* g, const float* c_prev, const float* h_prev, float* gates_act, float* c_new,
                            float* y, const float* mask_out, float keep_out, const int32_t* lens, int t,
                            float* c_state, float* h_state, int B, int D, float* xh_next, int xh_ld, int S,
                            const float* bias, hipStream_t st) {
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, g, c_prev, h_prev, gates_act,
                     c_new, (float*)nullptr, y, mask_out, keep_out, lens, t, c_state, h_state, B, D, xh_next, xh_ld,
                     S, bias);
  COMIC_LAUNCH_CHECK("lstm_gates_fwd");
  return 0;
}

extern "C" int comic_lstm_gates_bwd(const float* gates_act, const float* c_prev, const float* c_new, const float* dy,
                                    const float* mask_out, float keep_out, const int32_t* lens, int t,
                                    float* dc_state, float* dh_state, float* dg, int B, int D, void* stream) {
  COMIC_REQUIRE(gates_act && c_new && dc_state && dh_state && dg, "lstm_gates_bwd: null pointer");
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, gates_act,
                     c_prev, c_new, dy, mask_out, keep_out, lens, t, dc_state, dh_state, dg, B, D,
                     (const float*)nullptr, 0);
  COMIC_LAUNCH_CHECK("lstm_gates_bwd");
  return 0;
}

int comic_lstm_gates_bwd_ex(const float* gates_act, const float* c_prev, const float* c_new, const float* dy,
                            const float* dy_part, int S, const float* mask_out, float keep_out, const int32_t* lens,
                            int t, float* dc_state, float* dh_state, float* dg, int B, int D, hipStream_t st) {
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, st, gates_act, c_prev, c_new, dy,
                     mask_out, keep_out, lens, t, dc_state, dh_state, dg, B, D, dy_part, S);
  COMIC_LAUNCH_CHECK("lstm_gates_bwd");
  return 0;
}

int comic_attn_splits(int B, int M);
// executor-internal forms (fused state plumbing); the public ops pass no extras
int comic_attn_fwd_ex(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                      const float* ln_g, const float* ln_b, const float* v, const float* tau, const float* mask_alpha,
                      float keep_alpha, float* alpha, float* alpha_d, float* ctx, const int32_t* lens, int t,
                      const float* att_prev, float* att_next, float* xh_next, int xh_ld, const float* mask_next,
                      int mask_ld, float keep_in, int q_parts, float* q_out, float* scores_ws, hipStream_t st, int mem_div) {
  if (int rc = attn_check(d)) return rc;
  COMIC_REQUIRE(mem_div >= 1 && d->B % mem_div == 0, "attn_fwd: the batch must be a multiple of the rows per shared memory");
  COMIC_REQUIRE(keys && values && q && alpha && alpha_d && ctx, "attn_fwd: null pointer");
  COMIC_REQUIRE(d->method != 0 || (ln_g && ln_b && v && tau), "attn_fwd: add_LN needs ln_g/ln_b/v/tau");
  AttnArgs a{};
  a.d = *d;
  a.keys = keys; a.values = values; a.q = q; a.ln_g = ln_g; a.ln_b = ln_b; a.v = v; a.tau = tau;
  a.mask_alpha = mask_alpha; a.keep_alpha = keep_alpha; a.alpha = alpha; a.alpha_d = alpha_d; a.ctx = ctx;
  a.lens = lens; a.t = t; a.att_prev = att_prev; a.att_next = att_next; a.xh_next = xh_next; a.xh_ld = xh_ld;
  a.mask_next = mask_next; a.mask_ld = mask_ld; a.keep_in = keep_in; a.q_parts = q_parts; a.q_out = q_out;
  a.mem_div = mem_div;
  a.stop = g_comic_stop.p; a.stop_t = g_comic_stop.t;
  const size_t lds = (size_t)d->H * d->M * sizeof(float);
  const int S = scores_ws ? comic_attn_splits(d->B, d->M) : 1;
  if (S > 1) {        // large memory: S workgroups per batch row, scores and probability / context as two launches
    a.ws_s = scores_ws;
    int rc = attn_dispatch(d->D, [&](auto epl) {
      hipLaunchKernelGGL((attn_scores_kernel<decltype(epl)::value>), dim3(d->B, S), dim3(kAttnThreads), 0, st, a);
    });
    if (rc) return rc;
    hipLaunchKernelGGL(attn_ctx_kernel, dim3(d->B, S), dim3(kAttnThreads), lds + kAttnThreads * sizeof(float), st, a);
    COMIC_LAUNCH_CHECK("attn_fwd (split)");
    return 0;
  }
  int rc = attn_dispatch(d->D, [&](auto epl) {
    const int grid = mem_div > 1 ? ((d->B / mem_div + 7) / 8) * 8 * mem_div : d->B;
    hipLaunchKernelGGL((attn_fwd_kernel<decltype(epl)::value>), dim3(grid), dim3(kAttnThreads), lds, st, a);
  });
  if (rc) return rc;
  COMIC_LAUNCH_CHECK("attn_fwd");
  return 0;
}

// workgroups per batch row of the attention kernels when a scratch for the split form is available: enough to put a
// workgroup on every CU (batch 64 -> 4), 1 for the memories the one-workgroup kernels are built for
int comic_attn_splits(int B, int M) {
  if (M < kAttnSplitMinM || B < 1) return 1;
  const int S = std::min(8, (256 + B - 1) / B);      // 8 per row at batch 64 measured slower (4.13 vs 3.68 ms per step)
  return S >= 2 ? S : 1;
}

int comic_attn_bwd_ex(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                      const float* ln_g, const float* ln_b, const float* v, const float* tau, const float* alpha,
                      const float* mask_alpha, float keep_alpha, const float* dctx, const float* dmap, float* dq,
                      float* dkeys, float* dvalues, float* pgrad, const int32_t* lens, int t, hipStream_t st,
                      int pgrad_overwrite, float* ws_s, float* ws_d) {
  if (int rc = attn_check(d)) return rc;
  COMIC_REQUIRE(keys && values && q && alpha && dctx && dq && dkeys && dvalues, "attn_bwd: null pointer");
  COMIC_REQUIRE(d->method != 0 || (ln_g && ln_b && v && tau), "attn_bwd: add_LN needs ln_g/ln_b/v/tau");
  AttnArgs a{};
  a.d = *d;
  a.keys = keys; a.values = values; a.q = q; a.ln_g = ln_g; a.ln_b = ln_b; a.v = v; a.tau = tau;
  a.alpha_in = alpha; a.mask_alpha = mask_alpha; a.keep_alpha = keep_alpha; a.dctx = dctx; a.dmap = dmap;
  a.dq = dq; a.dkeys = dkeys; a.dvalues = dvalues; a.pgrad = pgrad; a.lens = lens; a.t = t;
  a.pgrad_overwrite = pgrad_overwrite;
  COMIC_REQUIRE(pgrad_overwrite != 2 || d->prob == 0, "attn_bwd: the split form needs the softmax probability fn");
  const size_t lds = ((size_t)d->H * d->M * 3 + kAttnWaves * 512 + kAttnWaves + 16) * sizeof(float);
  // pgrad_overwrite == 2: split mode -- the caller zero-filled dq and the pgrad rows; two workgroups per batch row, or,
  // with scratch for the scores and a large memory, comic_attn_splits workgroups behind attn_bwd_scores_kernel
  const bool split = pgrad_overwrite == 2;
  const int S = (split && ws_s && ws_d) ? comic_attn_splits(d->B, d->M) : 1;
  if (S > 2) {
    a.ws_s = ws_s;
    a.ws_d = ws_d;
    a.ws_parts = ws_d + (size_t)d->B * d->H * d->M;        // the caller's scratch holds all three (comic_attn_bwd_scratch)
    int rc0 = attn_dispatch(d->D, [&](auto epl) {
      hipLaunchKernelGGL((attn_bwd_scores_kernel<decltype(epl)::value>), dim3(d->B, S), dim3(kAttnThreads), 0, st, a);
    });
    if (rc0) return rc0;
  }
  int rc = attn_dispatch(d->D, [&](auto epl) {
    hipLaunchKernelGGL((attn_bwd_kernel<decltype(epl)::value>), dim3(d->B, S > 2 ? S : (split ? 2 : 1)), dim3(kAttnThreads), lds, st, a);
  });
  if (rc) return rc;
  if (S > 2) {
    const int n = (d->method == 0 && pgrad) ? 4 * d->D + 1 : d->D;
    hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3((n + 255) / 256, d->B), dim3(256), 0, st, a.ws_parts, dq, pgrad, d->B,
                       d->D, S, n);
  }
  COMIC_LAUNCH_CHECK("attn_bwd");
  return 0;
}

// floats of scratch the split backward needs behind ws_s: d alpha_d [B][H][M] + the partial rows
long comic_attn_bwd_scratch(int B, int H, int M, int D) {
  return (long)B * H * M + (long)comic_attn_splits(B, M) * B * (4 * D + 4);
}

extern "C" int comic_attn_step_fwd(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                                   const float* ln_g, const float* ln_b, const float* v, const float* tau,
                                   const float* mask_alpha, float keep_alpha, float* alpha, float* alpha_d, float* ctx,
                                   void* stream) {
  return comic_attn_fwd_ex(d, keys, values, q, ln_g, ln_b, v, tau, mask_alpha, keep_alpha, alpha, alpha_d, ctx, nullptr,
                           0, nullptr, nullptr, nullptr, 0, nullptr, 0, 1.f, 1, nullptr, nullptr, (hipStream_t)stream, 1);
}

extern "C" int comic_attn_step_bwd(const comic_attn_desc* d, const float* keys, const float* values, const float* q,
                                   const float* ln_g, const float* ln_b, const float* v, const float* tau,
                                   const float* alpha, const float* mask_alpha, float keep_alpha, const float* dctx,
                                   const float* dmap, float* dq, float* dkeys, float* dvalues, float* pgrad,
                                   void* stream) {
  return comic_attn_bwd_ex(d, keys, values, q, ln_g, ln_b, v, tau, alpha, mask_alpha, keep_alpha, dctx, dmap, dq,
                           dkeys, dvalues, pgrad, nullptr, 0, (hipStream_t)stream, 0, nullptr, nullptr);
}

// t_rows time steps of logits are processed; the [B, t_stride] tables are indexed b*t_stride + t
int comic_xent_ex(float* logits, const int32_t* targets_bt, const float* coef_bt, const float* wmask_bt,
                  const int32_t* lens, float* loss_rows, float* dlogits, int32_t* ids_tb, int t_rows, int t_stride,
                  int B, int V, hipStream_t st) {
  COMIC_REQUIRE(logits && targets_bt, "xent: null pointer");
  COMIC_REQUIRE(t_rows > 0 && t_rows <= t_stride, "xent: bad time extent");
  hipLaunchKernelGGL(xent_kernel, dim3(t_rows * B), dim3(256), 0, st, logits, targets_bt, coef_bt, wmask_bt, lens,
                     loss_rows, dlogits, ids_tb, t_stride, B, V, V);
  COMIC_LAUNCH_CHECK("xent");
  return 0;
}
// sequence loss (d logits with row stride ld_dl) + map loss in one launch; `partial`: >= ceil(Tp*B*M / 256) floats
int comic_xent_maploss(float* logits, const int32_t* targets_bt, const float* coef_bt, const float* wmask_bt,
                       const int32_t* lens, float* loss_rows, float* dlogits, int ld_dl, int32_t* ids_tb, int t_rows,
                       int t_stride, int B, int V, const float* hist, float* dmap, float* partial, float* map_loss,
                       unsigned* ticket, int H, int M, float scale, hipStream_t st) {
  COMIC_REQUIRE(logits && targets_bt && hist && partial && map_loss && ticket, "xent_maploss: null pointer");
  COMIC_REQUIRE(t_rows > 0 && t_rows <= t_stride && ld_dl >= V, "xent_maploss: bad extents");
  MapLossArgs ma{hist, dmap, partial, map_loss, ticket, t_rows, B, H, M, (int)cdiv64((long)t_rows * B * M, 256), scale};
  hipLaunchKernelGGL(xent_maploss_kernel, dim3(t_rows * B + ma.nparts), dim3(256), 0, st, logits, targets_bt, coef_bt,
                     wmask_bt, lens, loss_rows, dlogits, ids_tb, t_stride, B, V, ld_dl, t_rows * B, ma);
  COMIC_LAUNCH_CHECK("xent_maploss");
  return 0;
}

extern "C" int comic_xent_fwd_bwd(float* logits, const int32_t* targets_bt, const float* coef_bt,
                                  const float* wmask_bt, const int32_t* lens, float* loss_rows, float* dlogits,
                                  int32_t* ids_tb, int T, int B, int V, void* stream) {
  return comic_xent_ex(logits, targets_bt, coef_bt, wmask_bt, lens, loss_rows, dlogits, ids_tb, T, T, B, V,
                       (hipStream_t)stream);
}

extern "C" int comic_adam_tf(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                             float beta2, float eps, float l2, float gscale, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, v,
                     (long)n, lr_t, beta1, beta2, eps, l2, gscale, (const float*)nullptr);
  COMIC_LAUNCH_CHECK("adam_tf");
  return 0;
}
extern "C" int comic_adam_tf_gated(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                                   float beta2, float eps, float l2, float gscale, const float* skip_flag, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, v,
                     (long)n, lr_t, beta1, beta2, eps, l2, gscale, skip_flag);
  COMIC_LAUNCH_CHECK("adam_tf_gated");
  return 0;
}

extern "C" int comic_clip_by_norm(float* g, const float* w, const int64_t* chunks, int n_chunks, float l2, float gscale,
                                  float clip_norm, float* partial, const float* skip_flag, void* stream) {
  if (n_chunks == 0 || !(clip_norm > 0.f)) return 0;
  COMIC_REQUIRE(g && w && chunks && partial && gscale != 0.f, "clip_by_norm: bad arguments");
  static_assert(sizeof(ClipChunk) == 5 * sizeof(int64_t), "chunk records are five int64");
  hipLaunchKernelGGL(clip_partial_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, g, w,
                     (const ClipChunk*)chunks, l2, gscale, partial);
  hipLaunchKernelGGL(clip_apply_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, g, w, (const ClipChunk*)chunks,
                     l2, gscale, clip_norm, (const float*)partial, skip_flag);
  COMIC_LAUNCH_CHECK("clip_by_norm");
  return 0;
}

extern "C" int comic_ln_tanh_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat, int B,
                                 int C, float eps, void* stream) {
  COMIC_REQUIRE(x && gamma && beta && y && xhat && B > 0 && C > 0 && C <= 2048, "ln_tanh_fwd: bad arguments (C %d)", C);
  hipLaunchKernelGGL(ln_tanh_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, xhat, C, eps);
  COMIC_LAUNCH_CHECK("ln_tanh_fwd");
  return 0;
}

extern "C" int comic_ln_tanh_bwd_rows(const float* dy, const float* y, const float* xhat, float* pgamma, float* pbeta,
                                      int B, int C, void* stream) {
  COMIC_REQUIRE(dy && y && xhat && pgamma && pbeta && B > 0 && C > 0, "ln_tanh_bwd_rows: bad arguments");
  const long n = (long)B * C;
  hipLaunchKernelGGL(ln_tanh_bwd_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y,
                     xhat, pgamma, pbeta, n);
  COMIC_LAUNCH_CHECK("ln_tanh_bwd_rows");
  return 0;
}

extern "C" int comic_momentum_tf(float* w, const float* g, float* accum, int64_t n, float lr, float momentum, float l2,
                                 float gscale, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(momentum_tf_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, accum,
                     (long)n, lr, momentum, l2, gscale, (const float*)nullptr);
  COMIC_LAUNCH_CHECK("momentum_tf");
  return 0;
}
extern "C" int comic_momentum_tf_gated(float* w, const float* g, float* accum, int64_t n, float lr, float momentum, float l2,
                                       float gscale, const float* skip_flag, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(momentum_tf_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, accum,
                     (long)n, lr, momentum, l2, gscale, skip_flag);
  COMIC_LAUNCH_CHECK("momentum_tf_gated");
  return 0;
}

// test aid: workgroups that stay resident for a while (bounded)
__global__ void occupy_kernel(long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int comic_debug_occupy_cus(int n_workgroups, int microseconds, void* stream) {
  COMIC_REQUIRE(n_workgroups > 0 && n_workgroups <= 4096 && microseconds >= 0 && microseconds <= 2000000, "occupy: bad arguments");
  hipLaunchKernelGGL(occupy_kernel, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream, (long)microseconds * 100);
  COMIC_LAUNCH_CHECK("occupy");
  return 0;
}

extern "C" int comic_colsum(const float* in, float* out, int rows, int cols, float beta, void* stream) {
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(64), 0, (hipStream_t)stream, in, out, rows, cols,
                     beta);
  COMIC_LAUNCH_CHECK("colsum");
  return 0;
}

// executor-internal: `ws` holds at least 64*cols floats
int comic_colsum_ws(const float* in, float* out, int rows, int cols, float beta, float* ws, hipStream_t st) {
  if (!ws || rows < 256) return comic_colsum(in, out, rows, cols, beta, (void*)st);
  const int R = std::min(64, (rows + 31) / 32);
  const int rpc = (rows + R - 1) / R;
  hipLaunchKernelGGL(colsum_part_kernel, dim3(cdiv(cols, 64), R), dim3(64), 0, st, in, ws, rows, cols, rpc);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(64), 0, st, (const float*)ws, out, R, cols, beta);
  COMIC_LAUNCH_CHECK("colsum_ws");
  return 0;
}

extern "C" int comic_axpy(float* y, const float* x, float a, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, a,
                     (long)n);
  COMIC_LAUNCH_CHECK("axpy");
  return 0;
}
