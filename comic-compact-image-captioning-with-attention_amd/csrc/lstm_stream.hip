// LSTM step of the decode loops at many rows (beam search: rows = batch x beam = 150 ... 224): the gate product as ONE
// streaming pass over the [Wd][4D] kernel, then the cell as an element-wise reduction of its K-slices.
//
// Replaces, inside infer_step_fused (decoder_exec.hip; BasicLSTMCell of rnn_decoder_*search, common/ops_rnn.py:49-180):
// comic_lstm_step_fused, whose grid (unit tiles x 16-row tiles) re-reads the kernel once per row tile -- at 150 rows
// and Wd = 2816 (word baseline: E 256 + A 2048 + D 512) ten passes over 23 MB out of the L2, 42.7 us per step.
//
// Here the kernel is packed once per decode call (lstm_pack_k_kernel) as bf16 hi / lo halves in MFMA-fragment order,
// per CHUNK of 16 hidden units (4 gates x 16 units = 64 columns) and 32-deep k-step: 8 KB.  A workgroup (chunk, K-slice)
// pulls its slice (<= 16 k-steps, <= 128 KB) into the LDS by LDS-DMA in one go, multiplies ALL rows by it
// (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, the arithmetic of comic_gemm_f32_split3, product error ~2^-16;
// the waves split the 16-row tiles, the operand rows arrive as pre-split fragments written by the step's gather kernel)
// and stores its partial gate sums; D[col][row] orientation puts the four gates of a unit into one lane.  The kernel is
// read from HBM exactly once per step.  lstm_cell_kernel adds the slices in slice order (deterministic), the bias and
// the forget bias, and applies the cell -- c2, h2, y as comic_lstm_step_fused writes them, plus y as hi / lo fragments
// for the products that consume it.  The same streaming kernel serves plain skinny products out = x W over the step's
// fragments (comic_stream_gemm: query projection; vocabulary projection at a small V; two of them in one launch).
#include <algorithm>

#include "conv_common.h"
#include "lstm_prep.h"
#include "lstm_stream_dev.h"

#define RC(x)      \
  do {             \
    int rc_ = (x); \
    if (rc_) return rc_; \
  } while (0)

namespace {

constexpr int kStepBytes = kStreamStepBytes;
constexpr int kMaxSliceSteps = 16;       // 128 KB of LDS

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// K [Wd][N] fp32 -> [chunk][k-step][tile g][hi, lo][lane]: 8 bf16, lane (fr, fg) <-> K[32 s + 8 fg + j][g gstride + chunk cstride + fr]
// (LSTM kernel, N = 4D: tile g = gate g of 16 units, gstride D, cstride 16; plain matrix: four adjacent 16-column tiles,
// gstride 16, cstride 64)
__global__ __launch_bounds__(256) void lstm_pack_k_kernel(const float* __restrict__ K, uint4* __restrict__ out, int N, int Wd,
                                                          int KS, int gstride, int cstride, long units) {   // N: row stride AND column count of K (columns >= N pack as zeros)
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int lane = (int)(u & 63), hl = (int)((u >> 6) & 1), g = (int)((u >> 7) & 3);
  const long t = u >> 9;
  const int s = (int)(t % KS);
  const int c = (int)(t / KS);
  const int fr = lane & 15, fg = lane >> 4;
  const int col = g * gstride + c * cstride + fr, k0 = s * 32 + fg * 8;
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + 2 * j + e;
      x[e] = (k < Wd && col < N) ? K[(size_t)k * N + col] : 0.f;
    }
    const uint32_t h = pack_bf16x2(x[0], x[1]);
    w[j] = hl == 0 ? h : pack_bf16x2(x[0] - __uint_as_float(h << 16), x[1] - __uint_as_float(h & 0xFFFF0000u));
  }
  out[u] = make_uint4(w[0], w[1], w[2], w[3]);
}

// The step's operand gather (infer_prep_kernel: embedding of the last word | attention state and hidden state through
// the parent beams) written straight as hi / lo fragments, plus the gathered cell state (lstm_prep.h).  One thread per
// (row, segment); rows R .. Rp - 1 of the last tile are zero filled.
__global__ __launch_bounds__(256) void lstm_prep_frag_kernel(LstmPrepArgs p, const int32_t* __restrict__ ids,
                                                             const int32_t* __restrict__ parent, int W, int R, int Rp,
                                                             const int32_t* __restrict__ stop, int stop_t) {
  if (comic_stopped(stop, stop_t)) return;
  const int segs = p.KS * 4 + p.D / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Rp * segs) return;
  const int r = (int)(i / segs), sg = (int)(i % segs);
  const bool live = r < R;
  const int src = !live ? 0 : parent ? (r / W) * W + min(max(parent[r], 0), W - 1) : r;
  lstm_prep_segment(p, r, src, live ? ids[r] : -1, live, sg);
}

// One launch serves one product or two that read the same operand rows (query projection + vocabulary projection of a
// decode step): workgroups [0, n_a) belong to a, the rest to b; within a product, workgroup id = slice * chunks + chunk.
__global__ __launch_bounds__(512) void lstm_stream_kernel(LstmStreamArgs a, LstmStreamArgs b, int n_a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool second = (int)blockIdx.x >= n_a;
  lstm_stream_block(second ? b : a, smem, wave, lane, tid, second ? blockIdx.x - n_a : blockIdx.x);
}

// gates = sum of the K-slices (slice order) + bias; i, j, f, o -> c2 = c sigma(f + 1) + sigma(i) tanh(j), h2 = tanh(c2) sigma(o)
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias,
                                                        const float* __restrict__ c_prev, float* __restrict__ c_state,
                                                        float* __restrict__ h_state, float* __restrict__ y,
                                                        uint4* __restrict__ y_frag, int R, int D,
                                                        const int32_t* __restrict__ stop, int stop_t) {
  __shared__ float sy[256];
  if (comic_stopped(stop, stop_t)) return;
  const long t0 = (long)blockIdx.x * blockDim.x, t = t0 + threadIdx.x;
  float h2 = 0.f;
  if (t < (long)R * D) {
    const int r = (int)(t / D), d = (int)(t % D);
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= S; s += 4) {          // four slices' loads travel together; added in slice order
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = part + ((size_t)(s + u) * R + r) * 4 * D + d;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = p[(size_t)q * D];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] += v[u][q];
    }
    for (; s < S; ++s) {
      const float* p = part + ((size_t)s * R + r) * 4 * D + d;
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] += p[(size_t)q * D];
    }
    if (bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] += bias[q * D + d];
    }
    const float si = sigmoid_(g[0]), tj = tanhf(g[1]);
    const float sf = sigmoid_(g[2] + 1.0f), so = sigmoid_(g[3]);   // forget_bias = 1
    const float c2 = c_prev[t] * sf + si * tj;
    h2 = tanhf(c2) * so;
    c_state[t] = c2;
    h_state[t] = h2;
    y[t] = h2;
  }
  if (!y_frag) return;
  // y also as hi / lo fragments [16-row tile][k-step][hi, lo][lane] for the products that consume it (query
  // projection, vocabulary projection): a segment of 8 units is one lane's 16 bytes (D % 8 == 0)
  sy[threadIdx.x] = h2;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int seg = threadIdx.x >> 1, hl = threadIdx.x & 1;
    const long e = t0 + seg * 8;
    if (e < (long)R * D) {
      const int r = (int)(e / D), d = (int)(e % D), KS = (D + 31) / 32;
      const float* x = sy + seg * 8;
      uint32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t hh = pack_bf16x2(x[2 * j], x[2 * j + 1]);
        w[j] = hl == 0 ? hh : pack_bf16x2(x[2 * j] - __uint_as_float(hh << 16), x[2 * j + 1] - __uint_as_float(hh & 0xFFFF0000u));
      }
      y_frag[((size_t)(r >> 4) * KS + (d >> 5)) * 128 + hl * 64 + ((d >> 3) & 3) * 16 + (r & 15)] = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

}  // namespace

// Shapes the streaming step serves: hidden size a multiple of 16, operand thirds multiples of 8, 33 ... 256 rows.
bool comic_lstm_stream_supported(int D, int E, int A, int R) {
  return D % 16 == 0 && D >= 16 && E % 8 == 0 && A % 8 == 0 && D % 8 == 0 && R > 32 && R <= 256;
}
static inline int lstm_ks(int Wd) { return (Wd + 31) / 32; }
// K-slices: as many as fill the device (chunks x S ~ 256 workgroups), at most 16 k-steps each
static void lstm_slices(int N, int Wd, int* ksteps, int* S) {
  const int KS = lstm_ks(Wd), chunks = N / 64;
  int s = std::min(8, std::max(1, 256 / chunks));     // (more slices: more partial rows for the consumer to add)
  int n = (KS + s - 1) / s;
  if (n > kMaxSliceSteps) n = kMaxSliceSteps;
  *ksteps = n;
  *S = (KS + n - 1) / n;
}
int64_t comic_lstm_stream_kfrag_floats(int D, int Wd) { return (int64_t)4 * D * lstm_ks(Wd) * 32; }
int64_t comic_lstm_stream_xfrag_floats(int R, int Wd) { return (int64_t)((R + 15) / 16 * 16) * lstm_ks(Wd) * 32; }
int64_t comic_lstm_stream_part_bytes(int D, int Wd, int R) {
  int n, S;
  lstm_slices(4 * D, Wd, &n, &S);
  return (int64_t)S * R * 4 * D * 4;
}

static LstmStreamArgs stream_args(const void* k_frag, const void* x_frag, float* part, int R, int N, int Kin, int gstride,
                                  int cstride, int* S_out) {
  int ksteps, S;
  lstm_slices(N, Kin, &ksteps, &S);
  LstmStreamArgs a;
  a.k_frag = (const uint4*)k_frag; a.x_frag = (const uint4*)x_frag; a.part = part;
  a.R = R; a.N = N; a.KS = lstm_ks(Kin); a.ksteps = ksteps; a.gstride = gstride; a.cstride = cstride;
  a.stop = g_comic_stop.p; a.stop_t = g_comic_stop.t;
  *S_out = S;
  return a;
}
static int stream_launch2(const LstmStreamArgs& a, int Sa, const LstmStreamArgs* b, int Sb, hipStream_t st) {
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lstm_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
      comic_set_error("lstm_stream: cannot reserve the LDS");
      return 1;
    }
    attr_set = true;
  }
  const int n_a = (a.N / 64) * Sa, n_b = b ? (b->N / 64) * Sb : 0;
  const int ksteps = std::max(a.ksteps, b ? b->ksteps : 0);
  hipLaunchKernelGGL(lstm_stream_kernel, dim3(n_a + n_b), dim3(512), (size_t)ksteps * kStepBytes, st, a, b ? *b : a, n_a);
  return 0;
}
static int stream_launch(const void* k_frag, const void* x_frag, float* part, int R, int N, int Kin, int gstride, int cstride,
                         int* S_out, hipStream_t st) {
  const LstmStreamArgs a = stream_args(k_frag, x_frag, part, R, N, Kin, gstride, cstride, S_out);
  return stream_launch2(a, *S_out, nullptr, 0, st);
}

int comic_lstm_stream_pack(const float* K, void* k_frag, int D, int Wd, hipStream_t st) {
  const int KS = lstm_ks(Wd);
  const long units = (long)(D / 16) * KS * 512;
  hipLaunchKernelGGL(lstm_pack_k_kernel, dim3((unsigned)cdiv64(units, 256)), dim3(256), 0, st, K, (uint4*)k_frag, 4 * D, Wd, KS,
                     D, 16, units);
  COMIC_LAUNCH_CHECK("lstm_stream_pack");
  return 0;
}

// ---- plain skinny product out[R][N] = x[R][Kin] W[Kin][N] through the same streaming kernel (query projection) -----------
// x arrives as fragments (lstm_cell_kernel's y_frag), the result leaves as S K-slice partials [S][R][N] that the consumer
// sums in slice order (comic_attn_fwd_ex takes them as they are).
// (N need not be a multiple of 64: the packed matrix and the partial rows are padded to Np = ceil64(N) columns of zeros)
bool comic_stream_gemm_supported(int Kin, int N, int R) { return N >= 1 && Kin % 8 == 0 && R > 32 && R <= 256; }
static inline int np64(int N) { return (N + 63) / 64 * 64; }
int64_t comic_stream_gemm_wfrag_floats(int Kin, int N) { return (int64_t)np64(N) * lstm_ks(Kin) * 32; }
int64_t comic_stream_gemm_part_bytes(int Kin, int N, int R) {
  int n, S;
  lstm_slices(np64(N), Kin, &n, &S);
  return (int64_t)S * R * np64(N) * 4;
}
int comic_stream_gemm_pack(const float* Wm, void* w_frag, int Kin, int N, hipStream_t st) {
  const int KS = lstm_ks(Kin);
  const long units = (long)(np64(N) / 64) * KS * 512;
  hipLaunchKernelGGL(lstm_pack_k_kernel, dim3((unsigned)cdiv64(units, 256)), dim3(256), 0, st, Wm, (uint4*)w_frag, N, Kin, KS,
                     16, 64, units);
  COMIC_LAUNCH_CHECK("stream_gemm_pack");
  return 0;
}
// The product's launch record for a caller that runs its workgroups inside another launch (beam_logits.hip: the query
// projection beside the vocabulary projection); *n_wg workgroups of 512 threads, LDS *lds_bytes
LstmStreamArgs comic_stream_gemm_args(const void* x_frag, const void* w_frag, float* part, int R, int Kin, int N, int* S,
                                      int* n_wg, int* lds_bytes, int max_wg) {
  LstmStreamArgs a = stream_args(w_frag, x_frag, part, R, np64(N), Kin, 16, 64, S);
  if (max_wg > 0 && (a.N / 64) * *S > max_wg) {        // fewer, longer K-slices: at most max_wg workgroups (the host CUs are few)
    const int chunks = a.N / 64, s = std::max(1, max_wg / chunks);
    int n = std::min(kMaxSliceSteps, (a.KS + s - 1) / s);
    a.ksteps = n;
    *S = (a.KS + n - 1) / n;
  }
  *n_wg = (a.N / 64) * *S;
  *lds_bytes = a.ksteps * kStepBytes;
  return a;
}
// Two products over the same operand rows in one launch (outputs apart: part_a, part_b)
int comic_stream_gemm2(const void* x_frag, const void* w_a, float* part_a, int N_a, int* S_a, const void* w_b, float* part_b,
                       int N_b, int* S_b, int64_t part_bytes_each, int R, int Kin, hipStream_t st) {
  COMIC_REQUIRE(comic_stream_gemm_supported(Kin, N_a, R) && comic_stream_gemm_supported(Kin, N_b, R), "stream_gemm2: unsupported shape");
  COMIC_REQUIRE(part_bytes_each >= comic_stream_gemm_part_bytes(Kin, N_a, R) && part_bytes_each >= comic_stream_gemm_part_bytes(Kin, N_b, R),
                "stream_gemm2: partial buffer too small");
  const LstmStreamArgs a = stream_args(w_a, x_frag, part_a, R, np64(N_a), Kin, 16, 64, S_a);
  const LstmStreamArgs b = stream_args(w_b, x_frag, part_b, R, np64(N_b), Kin, 16, 64, S_b);
  RC(stream_launch2(a, *S_a, &b, *S_b, st));
  COMIC_LAUNCH_CHECK("stream_gemm2");
  return 0;
}
int comic_stream_gemm(const void* x_frag, const void* w_frag, float* part, int64_t part_bytes, int R, int Kin, int N, int* S,
                      hipStream_t st) {
  COMIC_REQUIRE(comic_stream_gemm_supported(Kin, N, R), "stream_gemm: unsupported shape (K %d, N %d, rows %d)", Kin, N, R);
  COMIC_REQUIRE(part_bytes >= comic_stream_gemm_part_bytes(Kin, N, R), "stream_gemm: partial buffer too small");
  RC(stream_launch(w_frag, x_frag, part, R, np64(N), Kin, 16, 64, S, st));     // partial rows of np64(N) floats
  COMIC_LAUNCH_CHECK("stream_gemm");
  return 0;
}

// One decode step: gather + split the operand rows (skip_prep: the previous step's beam merge already did), stream the
// kernel, apply the cell.
int comic_lstm_stream_step(const float* table, const int32_t* ids, const int32_t* parent, int W, const float* att_src,
                           const float* h_src, const float* c_src, const void* k_frag, const float* bias, void* x_frag,
                           float* c_in, float* part, int64_t part_bytes, float* c_state, float* h_state, float* y,
                           void* y_frag, int R, int E, int A, int D, int V, int skip_prep, hipStream_t st) {
  const int Wd = E + A + D, KS = lstm_ks(Wd);
  COMIC_REQUIRE(comic_lstm_stream_supported(D, E, A, R), "lstm_stream: unsupported shape (D %d, E %d, A %d, rows %d)", D, E, A, R);
  COMIC_REQUIRE(part_bytes >= comic_lstm_stream_part_bytes(D, Wd, R), "lstm_stream: partial buffer too small");
  const int Rp = (R + 15) / 16 * 16;
  if (!skip_prep) {
    const long n = (long)Rp * (KS * 4 + D / 8);
    LstmPrepArgs pa{table, att_src, h_src, c_src, (uint4*)x_frag, c_in, E, A, D, V, KS};
    hipLaunchKernelGGL(lstm_prep_frag_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, pa, ids, parent, W, R, Rp,
                       g_comic_stop.p, g_comic_stop.t);
  }
  int S = 1;
  RC(stream_launch(k_frag, x_frag, part, R, 4 * D, Wd, D, 16, &S, st));
  {
    const long n = (long)R * D;
    hipLaunchKernelGGL(lstm_cell_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (const float*)part, S, bias,
                       (const float*)c_in, c_state, h_state, y, (uint4*)y_frag, R, D, g_comic_stop.p, g_comic_stop.t);
  }
  COMIC_LAUNCH_CHECK("lstm_stream_step");
  return 0;
}

// ---- the streaming product as an operator of its own (C-ABI): out[R][N] = x[R][Kin] W[Kin][N] + bias ------------------------------
// What the decode executors do per step with resident packed weights, end to end in one call: pack W and the rows,
// stream, add the K-slices in slice order.  33 ... 256 rows, Kin a multiple of 8, any N.
namespace {
__global__ __launch_bounds__(256) void stream_pack_x_kernel(const float* __restrict__ x, uint4* __restrict__ out, int R, int Kin,
                                                            int KS, long units) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int lane = (int)(u & 63), hl = (int)((u >> 6) & 1);
  const long t = u >> 7;
  const int s = (int)(t % KS), tile = (int)(t / KS);
  const int row = tile * 16 + (lane & 15), k0 = s * 32 + (lane >> 4) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (row < R && k0 + j < Kin) ? x[(size_t)row * Kin + k0 + j] : 0.f;
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t h = pack_bf16x2(v[2 * j], v[2 * j + 1]);
    w[j] = hl == 0 ? h : pack_bf16x2(v[2 * j] - __uint_as_float(h << 16), v[2 * j + 1] - __uint_as_float(h & 0xFFFF0000u));
  }
  out[u] = make_uint4(w[0], w[1], w[2], w[3]);
}
__global__ __launch_bounds__(256) void stream_sum_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         float* __restrict__ out, int S, int R, int N, int Np) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)R * N) return;
  const int r = (int)(t / N), n = (int)(t % N);
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += part[((size_t)s * R + r) * Np + n];
  out[t] = acc + (bias ? bias[n] : 0.f);
}
}  // namespace

extern "C" int64_t comic_gemm_f32_stream_workspace(int R, int Kin, int N) {
  return (comic_stream_gemm_wfrag_floats(Kin, N) + comic_lstm_stream_xfrag_floats(R, Kin)) * 4 +
         comic_stream_gemm_part_bytes(Kin, N, R) + 1024;
}
extern "C" int comic_gemm_f32_stream(const float* x, const float* W, const float* bias, float* out, int R, int Kin, int N,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  COMIC_REQUIRE(x && W && out && workspace, "gemm_stream: null pointer");
  COMIC_REQUIRE(comic_stream_gemm_supported(Kin, N, R), "gemm_stream: 33 ... 256 rows, K a multiple of 8 (got rows %d, K %d)", R, Kin);
  COMIC_REQUIRE(workspace_bytes >= comic_gemm_f32_stream_workspace(R, Kin, N), "gemm_stream: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  void* w_frag = w;
  w += (comic_stream_gemm_wfrag_floats(Kin, N) * 4 + 255) / 256 * 256;
  void* x_frag = w;
  w += (comic_lstm_stream_xfrag_floats(R, Kin) * 4 + 255) / 256 * 256;
  float* part = (float*)w;
  RC(comic_stream_gemm_pack(W, w_frag, Kin, N, st));
  {
    const int KS = lstm_ks(Kin);
    const long units = (long)((R + 15) / 16) * KS * 2 * 64;
    hipLaunchKernelGGL(stream_pack_x_kernel, dim3((unsigned)cdiv64(units, 256)), dim3(256), 0, st, x, (uint4*)x_frag, R, Kin, KS,
                       units);
  }
  int S = 1;
  RC(comic_stream_gemm(x_frag, w_frag, part, comic_stream_gemm_part_bytes(Kin, N, R), R, Kin, N, &S, st));
  const long n = (long)R * N;
  hipLaunchKernelGGL(stream_sum_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, (const float*)part, bias, out, S, R, N,
                     np64(N));
  COMIC_LAUNCH_CHECK("gemm_stream");
  return 0;
}
