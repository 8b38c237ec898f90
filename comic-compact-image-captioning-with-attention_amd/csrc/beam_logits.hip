// Beam-search step at a large vocabulary (word tokens: V = 25 599, rows = batch x beam = 150): the vocabulary projection
// and the first half of the top-k as ONE streaming launch, the second half as a merge launch.
//
// Replaces, inside comic_decoder_beam (rnn_decoder_beam_search, common/ops_rnn.py:49-112; tf.contrib.seq2seq
// _beam_search_step [TF-1.9]): logits = y W_o + b_o, log_softmax, _mask_probs, top_k over the flattened beam * V axis.
// Round 2 ran it as four launches -- a hi/lo-split GEMM that wrote the [rows][V] logits (43.7 us: W_o streamed at
// 1.2 TB/s), a statistics pass and a per-chunk top-k pass that read them back (8.0 + 14.8 us) and a merge (6.6 us).
//
// Here a workgroup owns a CHUNK of 128 vocabulary columns for ALL rows:
//   * W_o is pre-packed once per decode call (beam_pack_wo_kernel): per chunk and 32-deep k-step the eight 16-column
//     tiles as bf16 hi and lo halves in MFMA-fragment order -- a chunk's K-quarter is 64 contiguous KB that go
//     global -> LDS by LDS-DMA (two quarters in flight), the 52 MB stream is read exactly once per step;
//   * the eight waves split the ROWS (16-row tiles); the rows of y arrive pre-split as hi / lo fragments (from the LSTM
//     cell kernel, or beam_pack_y_kernel) and go to registers with a rolling four-step prefetch, the weight fragments
//     come from the LDS: hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (the arithmetic of comic_gemm_f32_split3,
//     product error about 2^-16);
//   * D[v][row] orientation: a lane holds 4 consecutive columns of ONE row per tile, so the per-row work of the chunk is
//     lane-local plus two cross-lane steps (lanes r, r+16, r+32, r+48): the chunk's maximum and sum of exponentials
//     (the log-softmax partials) and its top-W columns BY LOGIT -- inside one beam the order by logit is the order by
//     total score, so the global top-W of an entry lies in the union of its (beam, chunk) top-W lists;
//   * the logits never reach memory: per row and chunk 2 + 2W words instead of 128.
// The merge launch (one workgroup per entry) combines the partials in chunk order into the log-softmax constants,
// scores the candidates, applies _mask_probs to finished beams (their candidates are synthesised: EOS and the lowest
// columns), selects the top W under the total order (score descending, flat index ascending) with the bookkeeping of
// beam_step_kernel, counts the step's finished entries (the last one to arrive writes steps_executed) and gathers the
// next step's LSTM operand rows through the parents it has chosen (lstm_prep.h).
// Small vocabularies (V <= 1024: radix-256) take beam_step_small_kernel below: a beam's logits in a wave's registers,
// the same tail.
#include <float.h>

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

#ifndef BL_VT
#define BL_VT 7
#endif
// 16-column tiles per workgroup: 7 -> 112-column chunks, 229 workgroups at V = 25 599 (8 -> 200 of the 256 CUs)
constexpr int kVT = BL_VT;
constexpr int kChunkCols = 16 * kVT;    // vocabulary columns per workgroup
constexpr int kQuarterBytes = 4 * kVT * 2 * 1024;     // four k-steps of kVT tiles x {hi, lo} x 1 KB
constexpr int kBiasBytes = 512;

struct BLVal {
  float v;
  int i;
};
__device__ __forceinline__ bool bl_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// ---- W_o [D][ld] fp32 -> per chunk / k32-step / tile / {hi, lo} / lane: 8 bf16 ---------------------------------------
// (+ b_o padded with zeros to whole chunks, behind the fragments)
__global__ __launch_bounds__(256) void beam_pack_wo_kernel(const float* __restrict__ W_o, const float* __restrict__ b_o, int ld,
                                                           uint4* __restrict__ out, int D, int V, long units) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) {
    float* bias = (float*)(out + units);
    const long v = u - units;
    if (v < (long)((V + kChunkCols - 1) / kChunkCols) * kChunkCols) bias[v] = v < V ? b_o[v] : 0.f;
    return;
  }
  const int KS = D / 32;
  const int lane = (int)(u & 63), hl = (int)((u >> 6) & 1);
  const long t1 = u >> 7;
  const int vt = (int)(t1 % kVT);
  const long t = t1 / kVT;
  const int s = (int)(t % KS);
  const long c = t / KS;
  const int fr = lane & 15, fg = lane >> 4;
  const long v = c * kChunkCols + vt * 16 + fr;
  const int k0 = s * 32 + fg * 8;
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) x[e] = v < V ? W_o[(size_t)(k0 + 2 * j + e) * ld + v] : 0.f;
    const uint32_t h = pack_bf16x2(x[0], x[1]);
    w[j] = hl == 0 ? h : pack_bf16x2(x[0] - __uint_as_float(h << 16), x[1] - __uint_as_float(h & 0xFFFF0000u));
  }
  out[u] = make_uint4(w[0], w[1], w[2], w[3]);
}

struct BeamLogitsArgs {
  const uint4* y_frag;     // beam_pack_y_kernel: the step's decoder outputs as bf16 hi / lo fragments
  const uint4* wo_frag;    // beam_pack_wo_kernel
  const float* bias_pad;   // [chunks * 128] b_o, zero padded (beam_pack_wo_kernel)
  float* pmax;             // [R][chunks]
  float* psum;             // [R][chunks]
  float* cand_v;           // [R][chunks][W]
  int32_t* cand_i;         // [R][chunks][W]   column index, -1: no such candidate
  int R, D, V, W, chunks;
  const int32_t* stop;
  int stop_t;
};

// y [R][D] fp32 -> per 16-row tile / k32-step / {hi, lo} / lane: 8 bf16 (the B operand of the products, split once per
// step instead of once per workgroup)
__global__ __launch_bounds__(256) void beam_pack_y_kernel(const float* __restrict__ y, uint4* __restrict__ out, int R, int D,
                                                          long units, const int32_t* __restrict__ stop, int stop_t) {
  if (comic_stopped(stop, stop_t)) return;
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int KS = D / 32;
  const int lane = (int)(u & 63), hl = (int)((u >> 6) & 1);
  const long t = u >> 7;
  const int s = (int)(t % KS), tile = (int)(t / KS);
  const int row = tile * 16 + (lane & 15), k0 = s * 32 + (lane >> 4) * 8;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < R) {
    const float4 a = *(const float4*)(y + (size_t)row * D + k0), b = *(const float4*)(y + (size_t)row * D + k0 + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  }
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t h = pack_bf16x2(x[2 * j], x[2 * j + 1]);
    w[j] = hl == 0 ? h : pack_bf16x2(x[2 * j] - __uint_as_float(h << 16), x[2 * j + 1] - __uint_as_float(h & 0xFFFF0000u));
  }
  out[u] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Loads the compiler must not count (cdna_hip_programming.md, section 5.7 item 1): beside a builtin LDS-DMA hipcc waits
// vmcnt(0) before every ds_read and at every use of an ordinary load, which serialises the quarter's stream behind the
// quarter's products (measured: 4.3 us per quarter = 2.3 us of load latency + 2.0 us of products).  Both load kinds of the
// main loop are therefore inline asm and the waits are counted by hand.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bl_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// 16 bytes per lane global -> LDS (lds_dst: the wave's base address, lane l lands at + 16 l); M0 saved and restored
__device__ __forceinline__ void bl_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ u32x4_t bl_load16(const void* p) {
  u32x4_t r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void bl_wait(u32x4_t& a, u32x4_t& b) {      // vmcnt(N), and a / b are not read above it
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}

// One wave's share: NT (0, 1 or 2) of the 16-row tiles wave, wave + 8.  Every wave takes part in the LDS-DMA of every
// quarter and in every barrier whatever its NT.
//   registers (NT = 2): 64 accumulators, 64 of y fragments (one K-quarter, refilled step by step for the next quarter as
//   soon as a step's products have issued: a rolling four-step prefetch), 2 x 32 of weight fragments (half a k-step is
//   read from the LDS while the half before it multiplies).
//   per quarter q: drain (DMA(q) and Y(q, 0..3) have landed), barrier, issue DMA(q + 1), multiply quarter q while
//   Y(q + 1, s) is requested behind the products of step s.
template <int NT>
__device__ __forceinline__ void beam_logits_wave(const BeamLogitsArgs& a, unsigned char* smem, int wave, int lane, int tid) {
  const int fr = lane & 15, fg = lane >> 4;
  const int c = blockIdx.x, D = a.D, KS = D / 32, NQ = D / 128;
  const unsigned char* wsrc = (const unsigned char*)a.wo_frag + (size_t)c * KS * kVT * 2 * 1024;
  constexpr int NR = NT > 0 ? NT : 1;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(bl_lds_addr(smem)) + wave * 1024;

  auto issue = [&](int q, int buf) {     // a K-quarter: kVT rounds of 512 lanes x 16 B
#pragma unroll
    for (int i = 0; i < kVT; ++i)
      bl_dma16(wsrc + (size_t)q * kQuarterBytes + i * 8192 + tid * 16, lds0 + buf * kQuarterBytes + i * 8192);
  };
  int row[NR];
  const uint4* ysrc[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int r = (wave + 8 * m) * 16 + fr;
    row[m] = r < a.R ? r : -1;
    ysrc[m] = a.y_frag + (size_t)(wave + 8 * m) * KS * 128 + lane;
  }
  u32x4_t yf[NR][4][2];                   // [tile][k-step of the quarter][hi, lo]
  auto load_y = [&](int q, int s) {
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      yf[m][s][0] = bl_load16(ysrc[m] + (size_t)(q * 4 + s) * 128);
      yf[m][s][1] = bl_load16(ysrc[m] + (size_t)(q * 4 + s) * 128 + 64);
    }
  };
  f32x4_t acc[NR][kVT];
#pragma unroll
  for (int m = 0; m < NR; ++m)
#pragma unroll
    for (int vt = 0; vt < kVT; ++vt) acc[m][vt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (wave == 0 && lane < kChunkCols / 4) bl_dma16(a.bias_pad + c * kChunkCols + lane * 4, lds0 + 2 * kQuarterBytes);   // the chunk's bias
  issue(0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) load_y(0, s);
  for (int q = 0; q < NQ; ++q) {
    const int buf = q & 1;
    const bool more = q + 1 < NQ;
    // this wave's pieces of quarter q and its row fragments have landed.  (A counted wait that left the younger row-fragment
    // loads in flight -- vmcnt(8 NT) -- read stale weights in about one workgroup per launch whenever the row loads were
    // L2 hits and the LDS-DMA was not: the two kinds do not retire in issue order against each other.  Nothing below
    // relies on their relative order: everything is drained here, the next quarter's DMA and row loads are issued behind it.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // ... everybody's; the other buffer is no longer read
    if (more) issue(q + 1, buf ^ 1);
    if constexpr (NT > 0) {
      const uint4* wl = (const uint4*)(smem + buf * kQuarterBytes) + lane;
      uint4 wa[2][8];                                      // [ring][4 tiles x {hi, lo}] of one half k-step
      auto load_half = [&](int h, uint4 (&dst)[8]) {       // half h of k-step h / 2: tiles 4 (h & 1) .. of its kVT
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if ((h & 1) * 4 + (j >> 1) < kVT) dst[j] = wl[(((h >> 1) * kVT + (h & 1) * 4) * 2 + j) * 64];
      };
      load_half(0, wa[0]);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (h + 1 < 8) load_half(h + 1, wa[(h + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);                 // the next half's LDS reads are in flight before these products
        const int s = h >> 1;
        if ((h & 1) == 0) {                                // first use of the step's row fragments: landed since the top
#pragma unroll                                              // of the quarter; the statement keeps their readers below it
          for (int m = 0; m < NT; ++m) bl_wait<63>(yf[m][s][0], yf[m][s][1]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int vt = (h & 1) * 4 + j;
          if (vt >= kVT) continue;
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, wa[h & 1][2 * j]);
          const bf16x8_t al = __builtin_bit_cast(bf16x8_t, wa[h & 1][2 * j + 1]);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8_t, yf[m][s][0]), acc[m][vt], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8_t, yf[m][s][1]), acc[m][vt], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8_t, yf[m][s][0]), acc[m][vt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if ((h & 1) && more) load_y(q + 1, s);             // this step's y registers refill for the next quarter
      }
    }
  }
  if constexpr (NT == 0) return;

  // ---- per row: the chunk's log-softmax partials and its top-W columns by logit ------------------------------------------
  // lane (fr, fg) holds columns v = kChunkCols c + 16 vt + 4 fg + i (i < 4) of row `row[m]`
  const int v_base = c * kChunkCols + 4 * fg;
  float4 bias[kVT];
#pragma unroll
  for (int vt = 0; vt < kVT; ++vt) bias[vt] = *(const float4*)(smem + 2 * kQuarterBytes + (16 * vt + 4 * fg) * 4);
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    // branch-free: dead columns are -inf (exp -> 0, never the strict maximum of a scan)
    float x[4 * kVT];
    float mx = -INFINITY;
#pragma unroll
    for (int vt = 0; vt < kVT; ++vt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = v_base + 16 * vt + i;
        const float bb = i == 0 ? bias[vt].x : i == 1 ? bias[vt].y : i == 2 ? bias[vt].z : bias[vt].w;
        const float val = v < a.V ? acc[m][vt][i] + bb : -INFINITY;
        x[vt * 4 + i] = val;
        mx = fmaxf(mx, val);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float se = 0.f;
#pragma unroll
    for (int e = 0; e < 4 * kVT; ++e) se += __expf(x[e] - mx);
    // the four lane groups' partial sums: (s_r + s_r^16) + (s_r^32 + s_r^48), the same bits in all four lanes
    se += __shfl_xor(se, 16, 64);
    se += __shfl_xor(se, 32, 64);
    const size_t ro = (size_t)(row[m] < 0 ? 0 : row[m]) * a.chunks + c;
    if (fg == 0 && row[m] >= 0) {
      a.pmax[ro] = mx;
      a.psum[ro] = se;
    }
    for (int k = 0; k < a.W; ++k) {
      // lane-local strict maximum, first element on ties (elements ascend with the column index)
      float bv = -INFINITY;
      int be = -1;
#pragma unroll
      for (int e = 0; e < 4 * kVT; ++e) {
        const bool g = x[e] > bv;
        bv = g ? x[e] : bv;
        be = g ? e : be;
      }
      const int mine = be < 0 ? 0x7fffffff : v_base + 16 * (be >> 2) + (be & 3);
      int bi = mine;
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool g = bl_better(ov, oi, bv, bi);
        bv = g ? ov : bv;
        bi = g ? oi : bi;
      }
      // the lane that holds the winner retires it
      const int gone = (bi == mine) ? be : -1;
#pragma unroll
      for (int e = 0; e < 4 * kVT; ++e) x[e] = (e == gone) ? -INFINITY : x[e];
      if (fg == 0 && row[m] >= 0) {
        a.cand_v[ro * a.W + k] = bv;
        a.cand_i[ro * a.W + k] = bi == 0x7fffffff ? -1 : bi;
      }
    }
  }
}

// The last n_q workgroups of the launch are not chunks of W_o: they run the query projection q = y W_q of the same step
// (lstm_stream_dev.h) on the CUs the 229 chunks leave idle -- it reads the same y fragments and its 7 us no longer stand
// in front of the attention step as a launch of their own.
__global__ __launch_bounds__(512) void beam_logits_kernel(BeamLogitsArgs a, LstmStreamArgs q, int n_q) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= (int)gridDim.x - n_q) {
    lstm_stream_block(q, smem, wave, lane, tid, (int)blockIdx.x - ((int)gridDim.x - n_q));
    return;
  }
  const int tiles = (a.R + 15) >> 4;
  const int nt = (wave < tiles ? 1 : 0) + (wave + 8 < tiles ? 1 : 0);        // wave-uniform (scalar)
  if (nt == 2) beam_logits_wave<2>(a, smem, wave, lane, tid);
  else if (nt == 1) beam_logits_wave<1>(a, smem, wave, lane, tid);
  else beam_logits_wave<0>(a, smem, wave, lane, tid);
}

// ---- merge: one workgroup per batch entry -----------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}
// wave-wide best of one (score, flat index) pair per lane under the total order (score descending, index ascending);
// 0x7fffffff: no candidate.  Two DPP reductions: the maximum score, then the lowest index that carries it.
__device__ __forceinline__ BLVal wave_best(float v, int i) {
  const float mx = wave_max(i == 0x7fffffff ? -INFINITY : v);
  const int gi = wave_min_i32(((i != 0x7fffffff) & (v == mx)) ? i : 0x7fffffff);
  return BLVal{mx, gi};
}

// Shared state of an entry's workgroup between the candidate phases and the tail of the step
struct BeamEntryShared {
  float max[8], logsum[8], lp[8], selv[8], fv[64];
  int fin[8], sel[8], fi[64], alldone, word[8], par[8];
  long long len[8];
};

// Tail of a beam step, common to the large- and the small-vocabulary kernels: the W best of the W x W finalists in
// sh.fv / sh.fi (wave 0, one finalist per lane), bookkeeping, the step's completion counter (done_cnt may be null: the
// caller keeps steps_executed by other means) and the NEXT step's LSTM operand rows (prep.x_frag may be null).  Enter
// behind a barrier that made the finalists and sh.alldone = 1 visible.
__device__ __forceinline__ void beam_entry_tail(BeamEntryShared& sh, int b, int W, int V, int end_id,
                                                float* __restrict__ log_probs, int32_t* __restrict__ finished,
                                                int64_t* __restrict__ lengths, int32_t* __restrict__ word_ids,
                                                int32_t* __restrict__ parent_ids, float* __restrict__ scores,
                                                unsigned long long* __restrict__ done_cnt,
                                                int32_t* __restrict__ steps_executed, int t, int max_steps,
                                                const LstmPrepArgs& prep) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the W best of the finalists (wave 0, one finalist per lane)
  if (wave == 0) {
    float fv = lane < W * W ? sh.fv[lane] : -INFINITY;
    int fi = lane < W * W ? sh.fi[lane] : 0x7fffffff;
    for (int r = 0; r < W; ++r) {
      const BLVal best = wave_best(fv, fi);
      if (best.i != 0x7fffffff && fi == best.i) fi = 0x7fffffff;
      if (lane == 0) {
        int sel = best.i;
        if (sel == 0x7fffffff) {   // all-NaN corner, as in beam_step_kernel: lowest untaken flat index
          sel = 0;
          bool again = true;
          while (again) {
            again = false;
            for (int q = 0; q < r; ++q)
              if (sh.sel[q] == sel) {
                ++sel;
                again = true;
              }
          }
        }
        sh.sel[r] = sel;
        sh.selv[r] = best.v;
      }
    }
  }
  __syncthreads();
  if (tid < W) {
    const int f = sh.sel[tid];
    const int parent = f / V, word = f - parent * V;
    const int prev_fin = sh.fin[parent];
    const int fin = (prev_fin || word == end_id) ? 1 : 0;
    word_ids[b * W + tid] = word;
    parent_ids[b * W + tid] = parent;
    scores[b * W + tid] = sh.selv[tid];
    log_probs[b * W + tid] = sh.selv[tid];
    finished[b * W + tid] = fin;
    lengths[b * W + tid] = sh.len[parent] + (prev_fin ? 0 : 1);
    if (!fin) sh.alldone = 0;
    sh.word[tid] = word;
    sh.par[tid] = parent;
  }
  __syncthreads();
  // steps_executed = t + 1 at the first step after which every beam of every entry is finished: the last entry to
  // arrive at the step's counter sees how many entries are done
  if (tid == 0 && done_cnt) {
    const unsigned long long add = ((unsigned long long)(sh.alldone ? 1 : 0) << 32) | 1ull;
    const unsigned long long old = atomicAdd(done_cnt, add);
    if ((unsigned)(old & 0xffffffffull) == gridDim.x - 1) {
      const unsigned done = (unsigned)(old >> 32) + (sh.alldone ? 1u : 0u);
      if (done == gridDim.x && steps_executed[0] == max_steps) steps_executed[0] = t + 1;
    }
  }
  // the NEXT step's LSTM operand rows of this entry (lstm_prep.h): embedding of the new words, attention / hidden / cell
  // state of the parents -- what lstm_prep_frag_kernel would gather through the ids this workgroup has just chosen
  if (prep.x_frag) {
    const int segs = prep.KS * 4 + prep.D / 8;
    const int nt = blockDim.x;
    for (int i0 = tid; i0 < W * segs; i0 += nt * 6) {
      float4 la[6], lb[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + nt * u;
        if (i < W * segs) {
          const int w = i / segs;
          lstm_prep_load(prep, b * W + sh.par[w], sh.word[w], true, i - w * segs, la[u], lb[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + nt * u;
        if (i < W * segs) {
          const int w = i / segs;
          lstm_prep_store(prep, b * W + w, true, i - w * segs, la[u], lb[u]);
        }
      }
    }
  }
}

constexpr int kMergeThreads = 512;      // eight waves: a beam each up to beam 8, twice the threads for the fill and the gather
__global__ __launch_bounds__(512) void beam_merge2_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                          const float* __restrict__ cand_v, const int32_t* __restrict__ cand_i,
                                                          float* __restrict__ log_probs, int32_t* __restrict__ finished,
                                                          int64_t* __restrict__ lengths, int32_t* __restrict__ word_ids,
                                                          int32_t* __restrict__ parent_ids, float* __restrict__ scores,
                                                          int W, int V, int chunks, int end_id,
                                                          unsigned long long* __restrict__ done_cnt,
                                                          int32_t* __restrict__ steps_executed, int t, int max_steps,
                                                          LstmPrepArgs prep, const int32_t* __restrict__ stop, int stop_t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ BeamEntryShared sh;
  if (comic_stopped(stop, stop_t)) return;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = chunks * W, n = W * per;
  float* c_tot = (float*)dyn;                  // candidate scores / flat indices of this entry (0x7fffffff: no candidate)
  int* c_f = (int*)(c_tot + n);
  // every global load the kernel can issue without a dependence goes out first: the beam state, the chunk partials
  // of this wave's first beam (<= 8 per lane) and the first 4 candidates per thread
  constexpr int CP = 8, CQ = 4;
  const bool pre = wave < W && chunks <= 64 * CP;
  float pm[CP], ps[CP];
  if (pre) {
    const size_t ro = (size_t)(b * W + wave) * chunks;
#pragma unroll
    for (int u = 0; u < CP; ++u) {
      const int k = lane + 64 * u;
      pm[u] = k < chunks ? pmax[ro + k] : -INFINITY;
      ps[u] = k < chunks ? psum[ro + k] : 0.f;
    }
  }
  int vv0[CQ];
  float xx0[CQ];
#pragma unroll
  for (int u = 0; u < CQ; ++u) {
    const int j = tid + u * kMergeThreads;
    vv0[u] = -1;
    xx0[u] = 0.f;
    if (j < n) {
      vv0[u] = cand_i[(size_t)b * n + j];
      xx0[u] = cand_v[(size_t)b * n + j];
    }
  }
  if (tid < W) {
    sh.lp[tid] = log_probs[b * W + tid];
    sh.fin[tid] = finished[b * W + tid];
    sh.len[tid] = lengths[b * W + tid];
  }
  if (tid == 0) sh.alldone = 1;
  // log-softmax constants of every beam from the per-chunk partials (a wave per beam; every lane walks its chunks in
  // ascending order, the wave reduction is a fixed tree: the same bits on every launch)
  for (int w = wave; w < W; w += kMergeThreads / 64) {
    const size_t ro = (size_t)(b * W + w) * chunks;
    float mx = -INFINITY, s = 0.f;
    if (pre && w == wave) {
#pragma unroll
      for (int u = 0; u < CP; ++u) mx = fmaxf(mx, pm[u]);
      mx = wave_max(mx);
#pragma unroll
      for (int u = 0; u < CP; ++u) s += ps[u] * __expf(pm[u] - mx);      // (absent chunks: 0 * exp(-inf) = 0)
    } else {
      for (int k = lane; k < chunks; k += 64) mx = fmaxf(mx, pmax[ro + k]);
      mx = wave_max(mx);
      for (int k = lane; k < chunks; k += 64) s += psum[ro + k] * __expf(pmax[ro + k] - mx);
    }
    s = wave_sum(s);
    if (lane == 0) {
      sh.max[w] = mx;
      sh.logsum[w] = logf(s);
    }
  }
  __syncthreads();
  // candidate slots: beam w, slot k < chunks * W.  A live beam's slot is the k-th (chunk, rank) entry of its lists; a
  // finished beam (_mask_probs: 0 at EOS, float32 min elsewhere) has W + 1 synthetic ones: EOS and the W lowest other
  // columns, which is all a top-W selection can ever take from it.
  const float inv_per = 1.0f / (float)per;
  auto place = [&](int j, int v_in, float x_in) {          // selects only: no branches
    const int w = (int)(((float)j + 0.5f) * inv_per), k = j - w * per;     // j / per (exact: j < 2^16, margin 0.5 / per)
    const bool fin = sh.fin[w] != 0;
    const int vs = (k == 0) ? end_id : (k - 1 + ((k - 1 >= end_id) ? 1 : 0));     // EOS, then the (k-1)-th column that is not EOS
    const bool ok = fin ? ((k <= W) & (vs < V)) : (v_in >= 0);
    const float step = fin ? ((vs == end_id) ? 0.f : -FLT_MAX) : ((x_in - sh.max[w]) - sh.logsum[w]);
    const int v = fin ? vs : v_in;
    c_tot[j] = ok ? sh.lp[w] + step : -INFINITY;
    c_f[j] = ok ? w * V + v : 0x7fffffff;
  };
#pragma unroll
  for (int u = 0; u < CQ; ++u) {
    const int j = tid + u * kMergeThreads;
    if (j < n) place(j, vv0[u], xx0[u]);
  }
  for (int j0 = tid + kMergeThreads * CQ; j0 < n; j0 += kMergeThreads * 4) {
    int vv[4];
    float xx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * kMergeThreads;
      vv[u] = -1;
      xx[u] = 0.f;
      if (j < n) {
        vv[u] = cand_i[(size_t)b * n + j];
        xx[u] = cand_v[(size_t)b * n + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * kMergeThreads;
      if (j < n) place(j, vv[u], xx[u]);
    }
  }
  __syncthreads();
  // the W best of every beam (a wave per beam): W x W finalists
  constexpr int CL = 16;
  for (int w = wave; w < W; w += kMergeThreads / 64) {
    if (per <= 64 * CL) {         // the beam's candidates in registers: one LDS pass, W branch-free rounds
      float rt[CL];
      int rf[CL];
#pragma unroll
      for (int u = 0; u < CL; ++u) {                       // (clamped index + select, bitwise conditions: no branches)
        const int j = lane + 64 * u, jc = min(j, per - 1);
        const int f = c_f[w * per + jc];
        const float tt = c_tot[w * per + jc];
        rf[u] = j < per ? f : 0x7fffffff;
        rt[u] = j < per ? tt : -INFINITY;
      }
      for (int r = 0; r < W; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < CL; ++u) {
          const bool g = (rf[u] != 0x7fffffff) & ((rt[u] > bv) | ((rt[u] == bv) & (rf[u] < bi)));
          bv = g ? rt[u] : bv;
          bi = g ? rf[u] : bi;
        }
        const BLVal best = wave_best(bv, bi);
#pragma unroll
        for (int u = 0; u < CL; ++u) rf[u] = rf[u] == best.i ? 0x7fffffff : rf[u];
        if (lane == 0) {
          sh.fv[w * W + r] = best.v;
          sh.fi[w * W + r] = best.i;
        }
      }
      continue;
    }
    for (int r = 0; r < W; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff, bj = -1;
      for (int j = w * per + lane; j < (w + 1) * per; j += 64) {
        const int f = c_f[j];
        if (f == 0x7fffffff) continue;
        const float tot = c_tot[j];
        if (bl_better(tot, f, bv, bi)) {
          bv = tot;
          bi = f;
          bj = j;
        }
      }
      const BLVal best = wave_best(bv, bi);
      if (best.i != 0x7fffffff && bi == best.i) c_f[bj] = 0x7fffffff;     // flat indices are unique: one lane retires it
      if (lane == 0) {
        sh.fv[w * W + r] = best.v;
        sh.fi[w * W + r] = best.i;
      }
    }
  }
  __syncthreads();
  beam_entry_tail(sh, b, W, V, end_id, log_probs, finished, lengths, word_ids, parent_ids, scores, done_cnt, steps_executed, t,
                  max_steps, prep);
}

// ---- small vocabularies (radix-256: V = 258): the whole step of an entry in one workgroup -------------------------------
// Replaces beam_step_kernel (decode.hip) inside comic_decoder_beam: that kernel scans the W V candidates of an entry
// W + 2 times from memory with a division per candidate and an 8-level block reduction per round (35.6 us at beam 7,
// V = 258: a third of the SCST rollout step).  Here a wave keeps a beam's logits in registers (V <= 1024): log-softmax
// constants, scores, and the beam's own top-W by W branch-free rounds + two DPP reductions each; then the common tail.
// Same arithmetic (expf / logf, sums in the same order) and the same total order as beam_step_kernel.
template <int CL>
__global__ __launch_bounds__(512) void beam_step_small_kernel(const float* __restrict__ logits, float* __restrict__ log_probs,
                                                              int32_t* __restrict__ finished, int64_t* __restrict__ lengths,
                                                              int32_t* __restrict__ word_ids, int32_t* __restrict__ parent_ids,
                                                              float* __restrict__ scores, int W, int V, int end_id,
                                                              const float* __restrict__ bias, int S, int ld, long slice_stride,
                                                              unsigned long long* __restrict__ done_cnt,
                                                              int32_t* __restrict__ steps_executed, int t, int max_steps,
                                                              LstmPrepArgs prep, const int32_t* __restrict__ stop, int stop_t) {
  __shared__ BeamEntryShared sh;
  if (comic_stopped(stop, stop_t)) return;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (eight waves: a beam each up to beam 8; CL = logits per lane: 5 serves V <= 320, 16 V <= 1024)
  // this wave's first beam: its logits are requested before the beam state is known
  // logits: [B W][ld] rows, or S K-slice partials of them slice_stride floats apart (comic_stream_gemm) + bias
  float x[CL];
  auto load_row = [&](int w) {
    const float* row = logits + (size_t)(b * W + w) * ld;
#pragma unroll
    for (int u = 0; u < CL; ++u) x[u] = row[min(lane + 64 * u, V - 1)];
    int sl = 1;
    for (; sl + 3 <= S; sl += 3) {        // three slices' loads travel together; added in slice order
      float v[3][CL];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int u = 0; u < CL; ++u) v[q][u] = row[(size_t)(sl + q) * slice_stride + min(lane + 64 * u, V - 1)];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int u = 0; u < CL; ++u) x[u] += v[q][u];
    }
    for (; sl < S; ++sl) {
#pragma unroll
      for (int u = 0; u < CL; ++u) x[u] += row[(size_t)sl * slice_stride + min(lane + 64 * u, V - 1)];
    }
    if (bias) {
#pragma unroll
      for (int u = 0; u < CL; ++u) x[u] += bias[min(lane + 64 * u, V - 1)];
    }
  };
  load_row(min(wave, W - 1));
  if (tid < W) {
    sh.lp[tid] = log_probs[b * W + tid];
    sh.fin[tid] = finished[b * W + tid];
    sh.len[tid] = lengths[b * W + tid];
  }
  if (tid == 0) sh.alldone = 1;
  __syncthreads();
  for (int w = wave; w < W; w += 8) {
    if (w != wave) load_row(w);
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < CL; ++u) {
      x[u] = lane + 64 * u < V ? x[u] : -INFINITY;
      mx = fmaxf(mx, x[u]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < CL; ++u) se += lane + 64 * u < V ? expf(x[u] - mx) : 0.f;
    const float logsum = logf(wave_sum(se));
    const bool fin = sh.fin[w] != 0;
    const float lp = sh.lp[w];
    float rt[CL];
    int rf[CL];
#pragma unroll
    for (int u = 0; u < CL; ++u) {
      const int v = lane + 64 * u;
      const float step = fin ? ((v == end_id) ? 0.f : -FLT_MAX) : ((x[u] - mx) - logsum);     // _mask_probs: dtype.min
      rt[u] = v < V ? lp + step : -INFINITY;
      rf[u] = v < V ? w * V + v : 0x7fffffff;
    }
    for (int r = 0; r < W; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int u = 0; u < CL; ++u) {
        const bool g = (rf[u] != 0x7fffffff) & ((rt[u] > bv) | ((rt[u] == bv) & (rf[u] < bi)));
        bv = g ? rt[u] : bv;
        bi = g ? rf[u] : bi;
      }
      const BLVal best = wave_best(bv, bi);
#pragma unroll
      for (int u = 0; u < CL; ++u) rf[u] = rf[u] == best.i ? 0x7fffffff : rf[u];
      if (lane == 0) {
        sh.fv[w * W + r] = best.v;
        sh.fi[w * W + r] = best.i;
      }
    }
  }
  __syncthreads();
  beam_entry_tail(sh, b, W, V, end_id, log_probs, finished, lengths, word_ids, parent_ids, scores, done_cnt, steps_executed, t,
                  max_steps, prep);
}

}  // namespace

// The fused form serves large vocabularies on decoders whose output size is a multiple of 128, up to 256 rows and 8 beams.
bool comic_beam_logits_supported(int D, int V, int R, int W) {
  return D % 128 == 0 && D >= 128 && D <= 1024 && V >= 4096 && R >= 1 && R <= 256 && W >= 1 && W <= 8 &&
         (long)W * V < (1L << 31) && (long)W * W * ((V + kChunkCols - 1) / kChunkCols) * 8 <= 144 * 1024;
}
int comic_beam_logits_chunks(int V) { return (V + kChunkCols - 1) / kChunkCols; }
// bytes of the packed W_o and of the per-step partial arrays
int64_t comic_beam_logits_pack_bytes(int D, int V) { return (int64_t)comic_beam_logits_chunks(V) * kChunkCols * (D + 1) * 4; }
// floats of the per-step partial arrays plus the per-step completion counters (8 bytes each)
// (and the step's y as hi / lo fragments: 4 bytes per element of the 16-row tiles)
int64_t comic_beam_logits_partial_floats(int D, int V, int R, int W, int max_steps) {
  return (int64_t)R * comic_beam_logits_chunks(V) * (2 + 2 * W) + 2 * (int64_t)max_steps + 4 + (int64_t)((R + 15) / 16 * 16) * D;
}

int comic_beam_pack_wo(const float* W_o, const float* b_o, int ld, void* wo_frag, int D, int V, hipStream_t st) {
  const long units = (long)comic_beam_logits_chunks(V) * (D / 32) * kVT * 2 * 64;
  const long total = units + (long)comic_beam_logits_chunks(V) * kChunkCols;
  hipLaunchKernelGGL(beam_pack_wo_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, W_o, b_o, ld, (uint4*)wo_frag,
                     D, V, units);
  COMIC_LAUNCH_CHECK("beam_pack_wo");
  return 0;
}

// zero the completion counters (once per decode call, before step 0; a kernel, so that a captured decode replays it)
__global__ void beam_zero_kernel(uint32_t* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
int comic_beam_logits_begin(float* partials, int B, int W, int V, int max_steps, hipStream_t st) {
  const int R = B * W;
  float* cnt = partials + (size_t)R * comic_beam_logits_chunks(V) * (2 + 2 * W);
  hipLaunchKernelGGL(beam_zero_kernel, dim3((unsigned)cdiv64(2L * max_steps, 256)), dim3(256), 0, st, (uint32_t*)cnt,
                     2L * max_steps);
  COMIC_LAUNCH_CHECK("beam_logits_begin");
  return 0;
}

// The step in two launches (a caller may put the attention step between them: the merge gathers the attention output
// for the next step's operand rows).
// y_frag_in: the step's decoder outputs already as fragments (lstm_cell_kernel), or null: split from y here.
// q / n_q / q_lds: an optional streaming product (comic_stream_gemm_args: the query projection) whose n_q workgroups ride
// at the end of the projection launch.
static void beam_logits_layout(BeamLogitsArgs& a, float* partials, int R, int W, int chunks, int max_steps,
                               unsigned long long** cnt, uint4** y_frag) {
  a.pmax = partials;
  a.psum = a.pmax + (size_t)R * chunks;
  a.cand_v = a.psum + (size_t)R * chunks;
  a.cand_i = (int32_t*)(a.cand_v + (size_t)R * chunks * W);
  *cnt = (unsigned long long*)(a.cand_i + (size_t)R * chunks * W);
  *y_frag = (uint4*)(((uintptr_t)(*cnt + max_steps) + 15) & ~(uintptr_t)15);
}
static int beam_logits_attrs() {
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)beam_logits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("beam_logits: cannot reserve the LDS");
      return 1;
    }
    if (hipFuncSetAttribute((const void*)beam_merge2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) {
      comic_set_error("beam_logits: cannot reserve the merge LDS");
      return 1;
    }
    attr_set = true;
  }
  return 0;
}
int comic_beam_logits_launch(const float* y, const void* y_frag_in, const void* wo_frag, float* partials, int max_steps, int B,
                             int W, int D, int V, const LstmStreamArgs* q, int n_q, int q_lds, hipStream_t st) {
  const int R = B * W, chunks = comic_beam_logits_chunks(V);
  COMIC_REQUIRE(comic_beam_logits_supported(D, V, R, W), "beam_logits: unsupported shape (D %d, V %d, rows %d, beam %d)", D, V, R, W);
  BeamLogitsArgs a;
  a.wo_frag = (const uint4*)wo_frag;
  a.bias_pad = (const float*)((const uint4*)wo_frag + (size_t)chunks * (D / 32) * kVT * 2 * 64);
  unsigned long long* cnt;
  uint4* y_frag;
  beam_logits_layout(a, partials, R, W, chunks, max_steps, &cnt, &y_frag);
  a.y_frag = y_frag_in ? (const uint4*)y_frag_in : y_frag;
  a.R = R; a.D = D; a.V = V; a.W = W; a.chunks = chunks;
  a.stop = g_comic_stop.p; a.stop_t = g_comic_stop.t;
  if (!y_frag_in) {
    const long units = (long)((R + 15) / 16) * (D / 32) * 2 * 64;
    hipLaunchKernelGGL(beam_pack_y_kernel, dim3((unsigned)cdiv64(units, 256)), dim3(256), 0, st, y, y_frag, R, D, units,
                       g_comic_stop.p, g_comic_stop.t);
  }
  RC(beam_logits_attrs());
  const int lds = std::max(2 * kQuarterBytes + kBiasBytes, q && n_q > 0 ? q_lds : 0);
  LstmStreamArgs qa{};
  if (q && n_q > 0) qa = *q;
  else n_q = 0;
  hipLaunchKernelGGL(beam_logits_kernel, dim3(chunks + n_q), dim3(512), lds, st, a, qa, n_q);   // row tiles w, w + 8 per wave
  COMIC_LAUNCH_CHECK("beam_logits_launch");
  return 0;
}
int comic_beam_merge_launch(float* partials, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                            int32_t* parent_ids, float* scores, int32_t* steps_executed, int t, int max_steps, int B, int W,
                            int V, int end_id, const LstmPrepArgs* prep, hipStream_t st) {
  const int R = B * W, chunks = comic_beam_logits_chunks(V);
  BeamLogitsArgs a;
  unsigned long long* cnt;
  uint4* y_frag;
  beam_logits_layout(a, partials, R, W, chunks, max_steps, &cnt, &y_frag);
  RC(beam_logits_attrs());
  const size_t merge_lds = (size_t)W * chunks * W * 8;
  hipLaunchKernelGGL(beam_merge2_kernel, dim3(B), dim3(kMergeThreads), merge_lds, st, (const float*)a.pmax, (const float*)a.psum,
                     (const float*)a.cand_v, (const int32_t*)a.cand_i, log_probs, finished, lengths, word_ids, parent_ids,
                     scores, W, V, chunks, end_id, cnt + t, steps_executed, t, max_steps, prep ? *prep : LstmPrepArgs{}, g_comic_stop.p, g_comic_stop.t);
  COMIC_LAUNCH_CHECK("beam_merge_launch");
  return 0;
}
int comic_beam_logits_step(const float* y, const void* y_frag_in, const void* wo_frag, float* partials, float* log_probs,
                           int32_t* finished, int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores,
                           int32_t* steps_executed, int t, int max_steps, int B, int W, int D, int V, int end_id,
                           const LstmPrepArgs* prep, hipStream_t st) {
  RC(comic_beam_logits_launch(y, y_frag_in, wo_frag, partials, max_steps, B, W, D, V, nullptr, 0, 0, st));
  return comic_beam_merge_launch(partials, log_probs, finished, lengths, word_ids, parent_ids, scores, steps_executed, t,
                                 max_steps, B, W, V, end_id, prep, st);
}

// ---- small vocabularies ------------------------------------------------------------------------------------------------------
bool comic_beam_step_small_supported(int V, int W) { return V >= 1 && V <= 1024 && W >= 1 && W <= 8 && (long)W * V < (1L << 31); }
// zero n 8-byte completion counters (once per decode call)
int comic_beam_counters_zero(void* cnt, int n, hipStream_t st) {
  hipLaunchKernelGGL(beam_zero_kernel, dim3((unsigned)cdiv64(2L * n, 256)), dim3(256), 0, st, (uint32_t*)cnt, 2L * n);
  COMIC_LAUNCH_CHECK("beam_counters_zero");
  return 0;
}
// One beam step from [B * W][V] logits.  cnt: this step's completion counter (zeroed before step 0) or null (the caller keeps
// steps_executed with its own launch); prep: gather the next step's LSTM operand rows, or null.
// logits: [B W][ld]; S > 1: K-slice partials slice_stride floats apart (comic_stream_gemm), summed in slice order, + bias
int comic_beam_step_small(const float* logits, const float* bias, int S, int ld, long slice_stride, float* log_probs,
                          int32_t* finished, int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores, int B,
                          int W, int V, int end_id, void* cnt, int32_t* steps_executed, int t, int max_steps,
                          const LstmPrepArgs* prep, hipStream_t st) {
  COMIC_REQUIRE(comic_beam_step_small_supported(V, W), "beam_step_small: unsupported shape (V %d, beam %d)", V, W);
  if (V <= 320)
    hipLaunchKernelGGL(beam_step_small_kernel<5>, dim3(B), dim3(512), 0, st, logits, log_probs, finished, lengths, word_ids,
                       parent_ids, scores, W, V, end_id, bias, S, ld, slice_stride, (unsigned long long*)cnt, steps_executed, t,
                       max_steps, prep ? *prep : LstmPrepArgs{}, g_comic_stop.p, g_comic_stop.t);
  else
    hipLaunchKernelGGL(beam_step_small_kernel<16>, dim3(B), dim3(512), 0, st, logits, log_probs, finished, lengths, word_ids,
                       parent_ids, scores, W, V, end_id, bias, S, ld, slice_stride, (unsigned long long*)cnt, steps_executed, t,
                       max_steps, prep ? *prep : LstmPrepArgs{}, g_comic_stop.p, g_comic_stop.t);
  COMIC_LAUNCH_CHECK("beam_step_small");
  return 0;
}
