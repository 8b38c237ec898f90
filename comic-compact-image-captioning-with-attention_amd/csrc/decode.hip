// Decoding kernels: row argmax (greedy), one fused beam-search step (log-softmax, finished
// masking, top-k over beam*V, length/finished bookkeeping), parent gather, gather_tree.
//
// Restates tf.contrib.seq2seq [TF-1.9] as used by common/ops_rnn.py:49-180:
//   GreedyEmbeddingHelper.sample = argmax (lowest index wins ties)
//   _beam_search_step: log_softmax -> _mask_probs(finished rows: float32.min, 0 at EOS)
//     -> total = log_probs[:, :, None] + step -> top_k(beam) over the flattened beam*V axis
//     (lower flat index first among equal values) -> word = idx % V, parent = idx / V
//   gather_tree: back-track parents from max_len-1, EOS-fill after the first EOS.
#include <float.h>

#include <algorithm>

#include "common.h"

namespace {

struct ValIdx {
  float v;
  int i;
};
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ ValIdx block_argmax(float v, int i, ValIdx* sh) {
  const int tid = threadIdx.x;
  sh[tid].v = v;
  sh[tid].i = i;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s && better(sh[tid + s].v, sh[tid + s].i, sh[tid].v, sh[tid].i)) sh[tid] = sh[tid + s];
    __syncthreads();
  }
  const ValIdx r = sh[0];
  __syncthreads();
  return r;
}

// 256-thread argmax with ONE barrier per call: wave-level butterfly, the four wave winners through LDS slots that
// alternate with `parity` (so a call needs no trailing barrier before the next one reuses the other slots).
__device__ __forceinline__ ValIdx block_argmax_1b(float v, int i, ValIdx* sh8, int parity) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (better(ov, oi, v, i)) {
      v = ov;
      i = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh8[parity * 4 + wave].v = v;
    sh8[parity * 4 + wave].i = i;
  }
  __syncthreads();
  ValIdx r = sh8[parity * 4];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const ValIdx o = sh8[parity * 4 + w];
    if (better(o.v, o.i, r.v, r.i)) r = o;
  }
  return r;
}

// noise (may be null): per-element Gumbel noise of SampleEmbeddingHelper's categorical draw -- argmax(logits + g) is a
// sample of softmax(logits)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                          int32_t* __restrict__ idx, int V) {
  __shared__ ValIdx sh[256];
  const float* row = x + (size_t)blockIdx.x * V;
  const float* nrow = noise ? noise + (size_t)blockIdx.x * V : nullptr;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float t = nrow ? row[v] + nrow[v] : row[v];
    if (better(t, v, bv, bi)) {
      bv = t;
      bi = v;
    }
  }
  const ValIdx r = block_argmax(bv, bi, sh);
  if (threadIdx.x == 0) idx[blockIdx.x] = r.i == 0x7fffffff ? 0 : r.i;
}

// One workgroup per batch entry.
__global__ __launch_bounds__(256) void beam_step_kernel(const float* __restrict__ logits, float* __restrict__ log_probs,
                                                        int32_t* __restrict__ finished, int64_t* __restrict__ lengths,
                                                        int32_t* __restrict__ word_ids, int32_t* __restrict__ parent_ids,
                                                        float* __restrict__ scores, int W, int V, int end_id,
                                                        float lpw, const int32_t* __restrict__ stop, int stop_t) {
  // lpw: length_penalty_weight of BeamSearchDecoder ([TF-1.9] _get_scores): candidates are ranked by
  // total / ((5 + length) / 6)^lpw with length = the beam's + 1 unless the beam is finished or the candidate is EOS;
  // the beam state keeps the unpenalised total, the step's `scores` output the penalised one.  0: no penalty.
  __shared__ ValIdx sh[256];
  __shared__ float s_max[64], s_logsum[64], s_lp[64];
  __shared__ int s_fin[64], s_sel[64];
  __shared__ float s_selv[64];
  __shared__ long long s_len[64];
  if (comic_stopped(stop, stop_t)) return;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lg = logits + (size_t)b * W * V;
  for (int w = tid; w < W; w += 256) {
    s_lp[w] = log_probs[b * W + w];
    s_fin[w] = finished[b * W + w];
    s_len[w] = lengths[b * W + w];
  }
  // log-softmax statistics per beam (one wave per beam)
  for (int w = wave; w < W; w += 4) {
    const float* row = lg + (size_t)w * V;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, row[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(row[v] - mx);
    s = wave_sum(s);
    if (lane == 0) {
      s_max[w] = mx;
      s_logsum[w] = logf(s);
    }
  }
  __syncthreads();
  const int total = W * V;
  for (int r = 0; r < W; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int f = tid; f < total; f += 256) {
      bool taken = false;
      for (int q = 0; q < r; ++q) taken |= (s_sel[q] == f);
      if (taken) continue;
      const int w = f / V, v = f - w * V;
      float step;
      if (s_fin[w])
        step = (v == end_id) ? 0.f : -FLT_MAX;  // dtype.min
      else
        step = (lg[f] - s_max[w]) - s_logsum[w];
      float tot = s_lp[w] + step;
      if (lpw != 0.f) {
        const long long len = s_len[w] + ((s_fin[w] || v == end_id) ? 0 : 1);
        tot = tot / powf((5.f + (float)len) / 6.f, lpw);
      }
      if (better(tot, f, bv, bi)) {
        bv = tot;
        bi = f;
      }
    }
    const ValIdx best = block_argmax(bv, bi, sh);
    if (tid == 0) {
      // all-(-inf) corner: fall back to the lowest untaken flat index (matches a stable sort)
      int sel = best.i;
      if (sel == 0x7fffffff) {
        sel = 0;
        bool again = true;
        while (again) {
          again = false;
          for (int q = 0; q < r; ++q)
            if (s_sel[q] == sel) {
              ++sel;
              again = true;
            }
        }
      }
      s_sel[r] = sel;
      s_selv[r] = best.v;
    }
    __syncthreads();
  }
  if (tid < W) {
    const int f = s_sel[tid];
    const int parent = f / V, word = f - parent * V;
    const int prev_fin = s_fin[parent];
    word_ids[b * W + tid] = word;
    parent_ids[b * W + tid] = parent;
    scores[b * W + tid] = s_selv[tid];
    float total = s_selv[tid];
    if (lpw != 0.f) {      // the state carries the unpenalised total log probability of the chosen candidate
      const float step = prev_fin ? ((word == end_id) ? 0.f : -FLT_MAX) : ((lg[f] - s_max[parent]) - s_logsum[parent]);
      total = s_lp[parent] + step;
    }
    log_probs[b * W + tid] = total;
    finished[b * W + tid] = (prev_fin || word == end_id) ? 1 : 0;
    lengths[b * W + tid] = s_len[parent] + (prev_fin ? 0 : 1);
  }
}

// ---- large vocabularies: the step split over `chunks` workgroups per batch entry ------------------
// (word tokens: V = 25 599, beam 3 -> 76 797 candidates per entry; one workgroup per entry leaves
// the GPU empty and scans them 2 + W times).  Same arithmetic and the same total order
// (value descending, flat index ascending) as beam_step_kernel:
//   beam_stats_kernel ..... per (entry, beam, chunk): max and sum exp(x - max) of the chunk
//   beam_chunk_topk_kernel  per (entry, chunk): log-softmax constants from the partials, then the
//                           chunk's own top-W candidates (W rounds over a cache-resident slice)
//   beam_merge_kernel ..... per entry: top-W of the chunks' candidates + bookkeeping
// The global top-W under a total order is the top-W of the union of the per-chunk top-W lists.
__global__ __launch_bounds__(256) void beam_stats_kernel(const float* __restrict__ logits, float* __restrict__ pmax,
                                                         float* __restrict__ psum, int W, int V, int chunks,
                                                         const int32_t* __restrict__ stop, int stop_t) {
  if (comic_stopped(stop, stop_t)) return;
  __shared__ float sh[4];
  const int c = blockIdx.x, w = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (V + chunks - 1) / chunks, v0 = c * per, v1 = min(V, v0 + per);
  const float* row = logits + ((size_t)b * W + w) * V;
  float mx = -INFINITY;
  for (int v = v0 + tid; v < v1; v += 256) mx = fmaxf(mx, row[v]);
  mx = wave_max(mx);
  if (lane == 0) sh[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  float s = 0.f;
  for (int v = v0 + tid; v < v1; v += 256) s += expf(row[v] - mx);
  s = wave_sum(s);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (tid == 0) {
    const size_t o = ((size_t)b * W + w) * chunks + c;
    pmax[o] = mx;
    psum[o] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}

template <int KLOCAL>
__global__ __launch_bounds__(256) void beam_chunk_topk_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ log_probs,
                                                              const int32_t* __restrict__ finished,
                                                              const float* __restrict__ pmax,
                                                              const float* __restrict__ psum, float* __restrict__ cand_v,
                                                              int32_t* __restrict__ cand_i, int W, int V, int chunks,
                                                              int end_id, const int32_t* __restrict__ stop, int stop_t) {
  __shared__ ValIdx sh[256];
  __shared__ float s_max[64], s_logsum[64], s_lp[64];
  __shared__ int s_fin[64], s_sel[64];
  if (comic_stopped(stop, stop_t)) return;
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* lg = logits + (size_t)b * W * V;
  // log-softmax constants of every beam from the per-chunk partials: one wave per beam, one lane per chunk
  // (chunks <= 32), partials combined in chunk order by lane 0 so that every workgroup of the entry gets the same bits
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int w = wave; w < W; w += 4) {
      const float pm = lane < chunks ? pmax[((size_t)b * W + w) * chunks + lane] : -INFINITY;
      const float ps = lane < chunks ? psum[((size_t)b * W + w) * chunks + lane] : 0.f;
      const float mx = wave_max(pm);
      const float term = lane < chunks ? ps * expf(pm - mx) : 0.f;
      float s = 0.f;
      for (int k = 0; k < chunks; ++k) s += __shfl(term, k, 64);      // fixed order: chunk 0, 1, ...
      if (lane == 0) {
        s_max[w] = mx;
        s_logsum[w] = logf(s);
        s_lp[w] = log_probs[b * W + w];
        s_fin[w] = finished[b * W + w];
      }
    }
  }
  __syncthreads();
  const int per = (V + chunks - 1) / chunks, v0 = c * per, v1 = min(V, v0 + per), nv = max(0, v1 - v0);
  const int total = W * nv;
  // Fast path: a thread's share of the W * nv candidates (column v0 + tid + 256*k of every beam) fits in registers.
  // ONE pass over the logits with all loads in flight together, then W selection rounds on the register copy
  // (the rescanning form below pays an L2 round trip per element and round: 31 -> 9 us at W = 3, V = 25 599).
  constexpr int kLocal = KLOCAL > 0 ? KLOCAL : 1;   // capacity chosen by the launcher (16 / 40; 0 = rescanning form)
  const int kper = (nv + 255) >> 8;                 // columns per thread and beam
  if (KLOCAL > 0 && W * kper <= kLocal) {
    __shared__ ValIdx sh8[8];
    float tv[kLocal];
    int ti[kLocal];
#pragma unroll
    for (int e = 0; e < kLocal; ++e) {
      tv[e] = -INFINITY;
      ti[e] = 0x7fffffff;
    }
    int w = 0, k = 0;                               // (beam, column slot) of register slot e: scalar counters
#pragma unroll
    for (int e = 0; e < kLocal; ++e) {
      const int v = v0 + tid + 256 * k;
      if (w < W && v < v1) {
        const int f = w * V + v;
        const float x = lg[f];
        const float step = s_fin[w] ? ((v == end_id) ? 0.f : -FLT_MAX) : (x - s_max[w]) - s_logsum[w];
        tv[e] = s_lp[w] + step;
        ti[e] = f;
      }
      if (++k == kper) {
        k = 0;
        ++w;
      }
    }
    for (int r = 0; r < W; ++r) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < kLocal; ++e)
        if (ti[e] != 0x7fffffff && better(tv[e], ti[e], bv, bi)) {
          bv = tv[e];
          bi = ti[e];
        }
      const ValIdx best = block_argmax_1b(bv, bi, sh8, r & 1);
#pragma unroll
      for (int e = 0; e < kLocal; ++e)
        if (ti[e] == best.i) ti[e] = 0x7fffffff;    // taken (flat indices are unique; 0x7fffffff marks "none")
      if (tid == 0) {
        const size_t o = ((size_t)b * chunks + c) * W + r;
        cand_v[o] = best.v;
        cand_i[o] = best.i;
      }
    }
    return;
  }
  for (int r = 0; r < W; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < total; j += 256) {
      const int w = j / nv, v = v0 + (j - w * nv);
      const int f = w * V + v;
      bool taken = false;
      for (int q = 0; q < r; ++q) taken |= (s_sel[q] == f);
      if (taken) continue;
      float step;
      if (s_fin[w])
        step = (v == end_id) ? 0.f : -FLT_MAX;  // dtype.min
      else
        step = (lg[f] - s_max[w]) - s_logsum[w];
      const float tot = s_lp[w] + step;
      if (better(tot, f, bv, bi)) {
        bv = tot;
        bi = f;
      }
    }
    const ValIdx best = block_argmax(bv, bi, sh);
    if (tid == 0) {
      s_sel[r] = best.i;                        // 0x7fffffff when the chunk has fewer than r+1 candidates
      const size_t o = ((size_t)b * chunks + c) * W + r;
      cand_v[o] = best.v;
      cand_i[o] = best.i;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void beam_merge_kernel(const float* __restrict__ cand_v, const int32_t* __restrict__ cand_i,
                                                         float* __restrict__ log_probs, int32_t* __restrict__ finished,
                                                         int64_t* __restrict__ lengths, int32_t* __restrict__ word_ids,
                                                         int32_t* __restrict__ parent_ids, float* __restrict__ scores,
                                                         int W, int V, int chunks, int end_id,
                                                         const int32_t* __restrict__ stop, int stop_t) {
  __shared__ ValIdx sh[256];
  __shared__ int s_fin[64], s_sel[64];
  if (comic_stopped(stop, stop_t)) return;
  __shared__ float s_selv[64];
  __shared__ long long s_len[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int w = tid; w < W; w += 256) {
    s_fin[w] = finished[b * W + w];
    s_len[w] = lengths[b * W + w];
  }
  __syncthreads();
  const int n = chunks * W;
  const float* cv = cand_v + (size_t)b * n;
  const int32_t* ci = cand_i + (size_t)b * n;
  for (int r = 0; r < W; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < n; j += 256) {
      const int f = ci[j];
      if (f == 0x7fffffff) continue;
      bool taken = false;
      for (int q = 0; q < r; ++q) taken |= (s_sel[q] == f);
      if (taken) continue;
      if (better(cv[j], f, bv, bi)) {
        bv = cv[j];
        bi = f;
      }
    }
    const ValIdx best = block_argmax(bv, bi, sh);
    if (tid == 0) {
      int sel = best.i;
      if (sel == 0x7fffffff) {   // all-NaN corner, as in beam_step_kernel: lowest untaken flat index
        sel = 0;
        bool again = true;
        while (again) {
          again = false;
          for (int q = 0; q < r; ++q)
            if (s_sel[q] == sel) {
              ++sel;
              again = true;
            }
        }
      }
      s_sel[r] = sel;
      s_selv[r] = best.v;
    }
    __syncthreads();
  }
  if (tid < W) {
    const int f = s_sel[tid];
    const int parent = f / V, word = f - parent * V;
    const int prev_fin = s_fin[parent];
    word_ids[b * W + tid] = word;
    parent_ids[b * W + tid] = parent;
    scores[b * W + tid] = s_selv[tid];
    log_probs[b * W + tid] = s_selv[tid];
    finished[b * W + tid] = (prev_fin || word == end_id) ? 1 : 0;
    lengths[b * W + tid] = s_len[parent] + (prev_fin ? 0 : 1);
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ parent,
                                   float* __restrict__ out, long total, int W, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  const int src = (r / W) * W + parent[r];
  out[i] = in[(size_t)src * cols + c];
}

__global__ void gather_tree_kernel(const int32_t* __restrict__ step_ids, const int32_t* __restrict__ parent_ids,
                                   const int32_t* __restrict__ max_len, int32_t* __restrict__ out, int T, int B, int W,
                                   int end_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, w = i % W;
  for (int t = 0; t < T; ++t) out[((size_t)t * B + b) * W + w] = end_id;
  const int L = min(T, max_len[b]);
  if (L <= 0) return;
  out[((size_t)(L - 1) * B + b) * W + w] = step_ids[((size_t)(L - 1) * B + b) * W + w];
  int parent = parent_ids[((size_t)(L - 1) * B + b) * W + w];
  for (int level = L - 2; level >= 0; --level) {
    if (parent < 0 || parent >= W) return;  // invalid trajectory: leave EOS (reference raises)
    out[((size_t)level * B + b) * W + w] = step_ids[((size_t)level * B + b) * W + parent];
    parent = parent_ids[((size_t)level * B + b) * W + parent];
  }
  bool fin = false;
  for (int t = 0; t < L; ++t) {
    int32_t* p = out + ((size_t)t * B + b) * W + w;
    if (fin)
      *p = end_id;
    else if (*p == end_id)
      fin = true;
  }
}

}  // namespace

// executor-internal: argmax of x + noise (noise null: of x)
int comic_argmax_rows_noise(const float* x, const float* noise, int32_t* idx, int rows, int V, hipStream_t st) {
  COMIC_REQUIRE(x && idx && rows > 0 && V > 0, "argmax_rows: bad arguments");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, st, x, noise, idx, V);
  COMIC_LAUNCH_CHECK("argmax_rows");
  return 0;
}
extern "C" int comic_argmax_rows(const float* x, int32_t* idx, int rows, int V, void* stream) {
  return comic_argmax_rows_noise(x, nullptr, idx, rows, V, (hipStream_t)stream);
}

// executor-internal: the step with BeamSearchDecoder's length penalty (length_penalty_weight; 0 = none)
int comic_beam_step_lp(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                       int32_t* parent_ids, float* scores, int B, int W, int V, int end_id, float lpw, hipStream_t st) {
  COMIC_REQUIRE(logits && log_probs && finished && lengths && word_ids && parent_ids && scores,
                "beam_step: null pointer");
  COMIC_REQUIRE(W >= 1 && W <= 64, "beam_step: beam width must be in [1,64] (got %d)", W);
  COMIC_REQUIRE((long)W * V < (1L << 31) && W <= V, "beam_step: beam*V too large or beam > V");
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(256), 0, st, logits, log_probs, finished, lengths, word_ids,
                     parent_ids, scores, W, V, end_id, lpw, g_comic_stop.p, g_comic_stop.t);
  COMIC_LAUNCH_CHECK("beam_step");
  return 0;
}
extern "C" int comic_beam_step(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths,
                               int32_t* word_ids, int32_t* parent_ids, float* scores, int B, int W, int V, int end_id,
                               void* stream) {
  return comic_beam_step_lp(logits, log_probs, finished, lengths, word_ids, parent_ids, scores, B, W, V, end_id, 0.f,
                            (hipStream_t)stream);
}

// executor-internal: with a workspace and a large vocabulary the step is split over several
// workgroups per entry (needs 2*B*W*chunks floats + B*chunks*W (float + int32))
int comic_beam_step_ws(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths, int32_t* word_ids,
                       int32_t* parent_ids, float* scores, int B, int W, int V, int end_id, void* ws, int64_t ws_bytes,
                       hipStream_t st) {
  int chunks = std::max(1, std::min(32, 1024 / std::max(1, B)));
  chunks = std::min(chunks, std::max(1, V / 1024));
  const int64_t need = ((int64_t)2 * B * W * chunks + (int64_t)2 * B * chunks * W) * 4 + 1024;
  if (!ws || ws_bytes < need || chunks < 2 || (long)W * V < 8192)
    return comic_beam_step(logits, log_probs, finished, lengths, word_ids, parent_ids, scores, B, W, V, end_id,
                           (void*)st);
  COMIC_REQUIRE(W >= 1 && W <= 64 && (long)W * V < (1L << 31) && W <= V, "beam_step: bad beam width");
  float* pmax = (float*)ws;
  float* psum = pmax + (size_t)B * W * chunks;
  float* cand_v = psum + (size_t)B * W * chunks;
  int32_t* cand_i = (int32_t*)(cand_v + (size_t)B * chunks * W);
  hipLaunchKernelGGL(beam_stats_kernel, dim3(chunks, W, B), dim3(256), 0, st, logits, pmax, psum, W, V, chunks,
                     g_comic_stop.p, g_comic_stop.t);
  {
    const int per = (V + chunks - 1) / chunks, kper = (per + 255) / 256;
    auto launch = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(chunks, B), dim3(256), 0, st, logits, (const float*)log_probs,
                         (const int32_t*)finished, (const float*)pmax, (const float*)psum, cand_v, cand_i, W, V, chunks,
                         end_id, g_comic_stop.p, g_comic_stop.t);
    };
    if (W * kper <= 16) launch(beam_chunk_topk_kernel<16>);
    else if (W * kper <= 40) launch(beam_chunk_topk_kernel<40>);
    else launch(beam_chunk_topk_kernel<0>);
  }
  hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(256), 0, st, (const float*)cand_v, (const int32_t*)cand_i,
                     log_probs, finished, lengths, word_ids, parent_ids, scores, W, V, chunks, end_id, g_comic_stop.p,
                     g_comic_stop.t);
  COMIC_LAUNCH_CHECK("beam_step (split)");
  return 0;
}

extern "C" int comic_gather_rows(const float* in, const int32_t* parent, float* out, int rows, int W, int cols,
                                 void* stream) {
  COMIC_REQUIRE(in != out, "gather_rows: in-place gather is not supported");
  const long total = (long)rows * cols;
  if (total == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     parent, out, total, W, cols);
  COMIC_LAUNCH_CHECK("gather_rows");
  return 0;
}

extern "C" int comic_gather_tree(const int32_t* step_ids, const int32_t* parent_ids, const int32_t* max_len,
                                 int32_t* out, int T, int B, int W, int end_id, void* stream) {
  hipLaunchKernelGGL(gather_tree_kernel, dim3(cdiv(B * W, 64)), dim3(64), 0, (hipStream_t)stream, step_ids,
                     parent_ids, max_len, out, T, B, W, end_id);
  COMIC_LAUNCH_CHECK("gather_tree");
  return 0;
}
