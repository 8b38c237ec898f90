// Grouped hi/lo-split bf16 GEMM with in-kernel split-K combine (see gemm_group.h).
//
// Why: after the backward time loop a training step of the decoder forms a dozen independent products -- every weight
// gradient (d K, d W_m, d W_q, d W_o, the embedding third of d gates * K^T), the bias column sums, the attention
// parameter sums.  As separate launches (profiles/r03_decoder_step_timeline.txt) they were ~30 kernels of 5-90 us on two
// lanes, each with its own split-K reduce launch and its own tail; here they are the work items of ONE launch:
//   * a work item = (problem, 128 x 128 output tile, k slice); items of about equal k length, ~2-3 per CU;
//   * a split tile's slices write their partial tile to a slab with sc1 (write-through) stores, drain (s_waitcnt vmcnt(0)
//     in every wave, workgroup barrier), and one lane takes a ticket on the tile's counter (agent-scope atomic add); the
//     workgroup whose ticket is the last one re-reads ALL S partials with sc1 loads in slice order -- its own included, so
//     the sum does not depend on who arrived last: bit-reproducible -- and runs the epilogue.  The form is row 1 of
//     MI355X_MICROARCH.md's table of hand-offs measured with sc1 loads in place of the acquire;
//   * a bias gradient is the product ones^T * dY (ones_a): one more problem of the group instead of two colsum launches.
// Arithmetic = comic_gemm_f32_split3 (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate).
#include <algorithm>

#include "gemm_group.h"
#include "gemm_x3_dev.h"

namespace {

constexpr int GB = 128;                 // tile edge
constexpr int kGemmGroupSlots = 512;    // workgroups of the four-wave kernel resident together: two per CU (80 KiB of LDS each)
constexpr int kItemOverhead = 6;        // k-tiles a work item costs beside its k loop (prologue, partial-tile store, its share of the combine)
constexpr int kSc1 = 16;                // cache-policy bit of raw buffer accesses: sc1
typedef __attribute__((ext_vector_type(4))) unsigned gg_u32x4_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gg_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 gg_load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const gg_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, kSc1);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void gg_store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  const gg_u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, kSc1);
}

constexpr int kGroupLds = 4 * (X3Tile<GB, true>::BYTES + X3Tile<GB, true>::BYTES);   // the largest of the three forms
static_assert(X3Tile<GB, true>::BYTES >= X3Tile<GB, false>::BYTES, "LDS of the k-contiguous form bounds the others");
constexpr int kGroupLdsPc = 2 * 2 * (X3TilePc<GB, false>::BYTES + X3TilePc<GB, false>::BYTES);   // two stages of the larger (row-contiguous) images
static_assert(X3TilePc<GB, false>::BYTES >= X3TilePc<GB, true>::BYTES, "LDS of the row-contiguous form bounds the others");

// four consecutive columns n .. n+3 of output row m
__device__ __forceinline__ void gg_emit(const ComicGemmProb& p, int m, int n, float4 a) {
  if (m >= p.M || n >= p.N) return;
  const int nv = min(4, p.N - n);
  float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
  float* cp = p.C + (size_t)m * p.ldc + n;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= nv) break;
    if (p.bias) v[q] += p.bias[n + q];
    if (p.mask) v[q] = (v[q] / p.keep) * p.mask[(size_t)m * p.ld_mask + n + q];
    if (p.beta != 0.f) v[q] += p.beta * cp[q];
  }
  if (nv == 4 && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0)) {
    *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nv) cp[q] = v[q];
  }
}

template <bool PC>
__global__ __launch_bounds__(PC ? 512 : 256) void gemm_group_x3_kernel(ComicGemmGroup g) {
  constexpr int NT = PC ? 512 : 256;                   // threads; PC: waves 0-3 own the output quadrants, waves 4-7 stream operands
  constexpr int LDS = PC ? kGroupLdsPc : kGroupLds;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Workgroup ids go round the 8 XCDs; with xcd_chunk the logical item index is made contiguous per XCD (items that share
  // operand tiles then share an L2).  speed only: any placement gives the same bits.
  int bid = blockIdx.x;
  if (g.xcd_chunk > 0 && bid < g.xcd_chunk * 8) bid = (bid & 7) * g.xcd_chunk + (bid >> 3);
  // the arrival flag of the split-K combine lives in the last bytes of the (by then idle) tile images: with a static
  // __shared__ word on top of the 80 KiB of dynamic LDS only ONE workgroup fitted a CU
  volatile unsigned* s_last = (volatile unsigned*)(smem + LDS - 16);
  // which problem: the launch's workgroups are laid out problem after problem
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < g.n; ++i)
    if (bid >= g.p[i].wg_begin) pi = i;
  const ComicGemmProb& p = g.p[pi];
  const int local = bid - p.wg_begin;
  // Item order inside a problem: slice-major, and inside a slice the tiles in panels of 8 tile columns walked row by row --
  // consecutive items (= one XCD under the xcd_chunk order, about 60 at a time) then cover a block of ~8 x 8 tiles of ONE
  // k slice and share their operand tiles through that XCD's L2 (tile-major order with the slices innermost read every B
  // tile from beyond the L2 once per item: 56 % L2 hits by counters).
  const int S = p.S, ntiles = p.tiles_m * p.tiles_n;
  const int slice = local / ntiles, tl = local - slice * ntiles;
  constexpr int PW = 8;
  const int panel = tl / (p.tiles_m * PW), rem = tl - panel * p.tiles_m * PW;
  const int pw = min(PW, p.tiles_n - panel * PW);
  const int mt = rem / pw, nt = panel * PW + rem % pw;
  const int tile = mt * p.tiles_n + nt;
  const int m0 = mt * GB, n0 = nt * GB;
  const int kbeg = S > 1 ? slice * p.k_per_slice : 0;
  const int kend = S > 1 ? min(p.K, kbeg + p.k_per_slice) : p.K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const long tile_id = (long)p.ticket0 + tile;
  const __amdgpu_buffer_rsrc_t sr = gg_rsrc(g.slab + ((long)p.slab_tile0 + (long)tile * S) * (GB * GB), (long)S * GB * GB * 4);
  const unsigned sbase = (unsigned)slice * (GB * GB * 4);
  if (p.ones_a) {
    // column sums in exact fp32 (a bias gradient is a sum with cancellation: the 16 mantissa bits of a hi + lo pair are
    // not enough for it): thread = (4 columns, k rows kg, kg + KG, ...), the row groups combined in a fixed order
    constexpr int KG = NT / 32;
    float4* red = (float4*)smem;                       // [KG][32]
    const int cg = tid & 31, kg = tid >> 5, n = n0 + 4 * cg;
    const bool vec = (p.ldb % 4 == 0) && (((uintptr_t)p.B & 15) == 0);
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N)
      for (int k = kbeg + kg; k < kend; k += KG) {
        const float4 v = load4(p.B + (size_t)k * p.ldb + n, p.N - n, vec);
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
    red[kg * 32 + cg] = sm;
    __syncthreads();
    if (tid < 32) {
      float4 tot = red[cg];
#pragma unroll
      for (int k = 1; k < KG; ++k) {
        const float4 v = red[k * 32 + cg];
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
      }
      if (S == 1) gg_emit(p, 0, n, tot);
      else gg_store16_sc1(sr, sbase + (unsigned)cg * 16u, tot);
    }
    if (S == 1) return;
  } else {
    f32x4_t acc[4][4];
    if constexpr (PC) {
      if (p.type == COMIC_GG_TN) x3_mainloop_pc<false, false>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
      else if (p.type == COMIC_GG_NN) x3_mainloop_pc<true, false>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
      else x3_mainloop_pc<true, true>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
    } else {
      if (p.type == COMIC_GG_TN)
        x3_mainloop<GB, false, false, GB>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
      else if (p.type == COMIC_GG_NN)
        x3_mainloop<GB, true, false, GB>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
      else
        x3_mainloop<GB, true, true, GB>(p.A, p.B, p.M, p.N, p.lda, p.ldb, m0, n0, kbeg, kend, smem, acc);
    }
    const bool owner = wave < 4;                         // (PC: the streaming waves hold no results)
    if (S == 1) {
      if (owner)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gg_emit(p, m0 + wm * 64 + j * 16 + (lane & 15), n0 + wn * 64 + i * 16 + (lane >> 4) * 4,
                  make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      return;
    }
    // partial tile -> slab tile slab_tile0 + tile * S + slice
    if (owner)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ml = wm * 64 + j * 16 + (lane & 15), nl = wn * 64 + i * 16 + (lane >> 4) * 4;
        gg_store16_sc1(sr, sbase + (unsigned)(ml * GB + nl) * 4u, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(g.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (t == (unsigned)(S - 1)) ? 1u : 0u;
    if (t == (unsigned)(S - 1)) __hip_atomic_store(g.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!*s_last) return;
  // last arriver: sum the S partials in slice order (sc1 loads: the partials were written by other CUs)
  const int n_chunks = p.ones_a ? 1 : (GB * GB / 4) / NT;     // column sums: row 0 of the tile only (32 float4)
#pragma unroll 1
  for (int c = 0; c < n_chunks; c += 4) {
    float4 sum[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = gg_load16_sc1(sr, (unsigned)s * (GB * GB * 4) + (unsigned)(tid + NT * (c + u)) * 16u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sum[u].x += v[u].x; sum[u].y += v[u].y; sum[u].z += v[u].z; sum[u].w += v[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx4 = tid + NT * (c + u);
      if (!p.ones_a || idx4 < GB / 4) gg_emit(p, m0 + idx4 / (GB / 4), n0 + (idx4 % (GB / 4)) * 4, sum[u]);
    }
  }
}


}  // namespace

int comic_gemm_group_plan(ComicGemmGroup& g, int target_items, int64_t* slab_bytes, int* n_tickets) {
  COMIC_REQUIRE(g.n > 0 && g.n <= kGemmGroupMax, "gemm_group: %d problems", g.n);
  long cost = 0;
  for (int i = 0; i < g.n; ++i) {
    ComicGemmProb& p = g.p[i];
    COMIC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.B && p.C && (p.A || p.ones_a), "gemm_group: bad problem %d", i);
    COMIC_REQUIRE(p.type >= COMIC_GG_TN && p.type <= COMIC_GG_NT, "gemm_group: bad type of problem %d", i);
    COMIC_REQUIRE(!p.ones_a || (p.type == COMIC_GG_TN && p.M == 1), "gemm_group: ones_a needs type TN and M = 1 (problem %d)", i);
    COMIC_REQUIRE(p.ldc >= p.N && (!p.mask || p.ld_mask >= p.N), "gemm_group: leading dimension of problem %d", i);
    if (!p.ones_a) COMIC_REQUIRE(p.lda >= (p.type == COMIC_GG_TN ? p.M : p.K), "gemm_group: lda of problem %d", i);
    COMIC_REQUIRE(p.ldb >= (p.type == COMIC_GG_NT ? p.K : p.N), "gemm_group: ldb of problem %d", i);
    p.tiles_m = cdiv(p.M, GB);
    p.tiles_n = cdiv(p.N, GB);
    cost += (long)cdiv(p.M, GB) * p.tiles_n * cdiv(p.K, 32);
  }
  // k-tiles (of 32) per work item.  A launch runs in ROUNDS of the workgroups that are resident together (kGemmGroupSlots:
  // two four-wave workgroups per CU), so what counts is rounds x the longest item, not the item count: the weight-gradient
  // group of a decoder step split into 580 items of 26-30 k-tiles ran two rounds for 1.13 rounds of work (112 us,
  // profiles/r05_decoder_step_timeline.txt).  Every item length from the old rule's up to the unsplit one is priced as
  // rounds x (longest item + kItemOverhead k-tiles of prologue, slab store and combine) and the cheapest taken; a smaller
  // `target_items` (the caller's retry when the slab is too small) raises the shortest length that is tried.
  auto split_of = [](int kt, long klen) { return (int)std::min<long>(16, std::max<long>(1, (kt + klen / 2) / klen)); };
  long klen = std::max<long>(8, cost / std::max(1, target_items));
  {
    long best = -1, best_klen = klen, kt_max = 8;
    for (int i = 0; i < g.n; ++i) kt_max = std::max<long>(kt_max, cdiv(g.p[i].K, 32));
    for (long kl = klen; kl <= kt_max; ++kl) {
      long items = 0, longest = 0;
      for (int i = 0; i < g.n; ++i) {
        const ComicGemmProb& p = g.p[i];
        const int kt = cdiv(p.K, 32), S = split_of(kt, kl);
        items += (long)p.tiles_m * p.tiles_n * S;
        if (!p.ones_a) longest = std::max<long>(longest, cdiv(kt, S));
      }
      const long est = cdiv64(items, kGemmGroupSlots) * (longest + kItemOverhead);
      if (best < 0 || est < best) {
        best = est;
        best_klen = kl;
      }
    }
    klen = best_klen;
  }
  int wg = 0, tickets = 0, slab_tiles = 0;
  for (int i = 0; i < g.n; ++i) {
    ComicGemmProb& p = g.p[i];
    const int nt = cdiv(p.M, GB) * p.tiles_n, kt = cdiv(p.K, 32);
    int S = split_of(kt, klen);
    p.k_per_slice = cdiv(kt, S) * 32;
    S = cdiv(p.K, p.k_per_slice);
    p.S = S;
    p.wg_begin = wg;
    p.slab_tile0 = slab_tiles;
    p.ticket0 = tickets;
    wg += nt * S;
    if (S > 1) {
      slab_tiles += nt * S;
      tickets += nt;
    }
  }
  if (slab_bytes) *slab_bytes = (int64_t)slab_tiles * GB * GB * 4;
  if (n_tickets) *n_tickets = tickets;
  return wg;
}

int comic_gemm_group_launch(const ComicGemmGroup& g_in, int n_wg, hipStream_t st) {
  ComicGemmGroup g = g_in;
  g.xcd_chunk = n_wg / 8;          // work items of one XCD are neighbours in the item order
  COMIC_REQUIRE(n_wg > 0, "gemm_group: empty launch");
  // producer / consumer kernel (one 8-wave workgroup per CU) when every product can be loaded 16 bytes at a time and the
  // launch fits the chip in one round; beyond that the four-wave kernel (two workgroups per CU) measured faster (the
  // weight-gradient group of the decoder step: 98 against 118 us; its single-round launches 19-23 against 23-29 us)
  bool pc = n_wg <= 256;
  for (int i = 0; i < g.n && pc; ++i) {
    const ComicGemmProb& p = g.p[i];
    if (p.ones_a) continue;
    const bool a_kc = p.type != COMIC_GG_TN, b_kc = p.type == COMIC_GG_NT;
    pc = p.lda % 4 == 0 && p.ldb % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0 &&
         (!(a_kc || b_kc) || p.K % 4 == 0);
  }
  static PerDeviceOnce once, once_pc;
  bool& done = pc ? once_pc.slot() : once.slot();
  const int lds = pc ? kGroupLdsPc : kGroupLds;
  if (!done) {
    const void* fn = pc ? (const void*)gemm_group_x3_kernel<true> : (const void*)gemm_group_x3_kernel<false>;
    COMIC_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess,
                  "gemm_group: cannot reserve %d bytes of LDS", lds);
    done = true;
  }
  if (pc) hipLaunchKernelGGL(gemm_group_x3_kernel<true>, dim3((unsigned)n_wg), dim3(512), lds, st, g);
  else hipLaunchKernelGGL(gemm_group_x3_kernel<false>, dim3((unsigned)n_wg), dim3(256), lds, st, g);
  COMIC_LAUNCH_CHECK("gemm_group");
  return 0;
}

namespace {
int gg_from_public(const comic_gemm_prob* probs, int n, ComicGemmGroup& g) {
  COMIC_REQUIRE(probs && n > 0 && n <= kGemmGroupMax, "gemm_group: 1..%d problems", kGemmGroupMax);
  g = ComicGemmGroup{};
  g.n = n;
  for (int i = 0; i < n; ++i) {
    const comic_gemm_prob& q = probs[i];
    ComicGemmProb& p = g.p[i];
    p.A = q.A; p.B = q.B; p.C = q.C; p.bias = q.bias; p.mask = q.mask;
    p.M = q.M; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc; p.ld_mask = q.ld_mask;
    p.alpha = q.alpha; p.beta = q.beta; p.keep = q.keep; p.type = q.type; p.ones_a = q.ones_a;
  }
  return 0;
}
#define kPublicTarget kGemmGroupTargetItems
}  // namespace

extern "C" int64_t comic_gemm_group_workspace(const comic_gemm_prob* probs, int n) {
  ComicGemmGroup g;
  if (gg_from_public(probs, n, g)) return -1;
  int64_t slab = 0;
  int nt = 0;
  if (comic_gemm_group_plan(g, kPublicTarget, &slab, &nt) < 0) return -1;
  return slab + 256 + (int64_t)nt * 4;
}

extern "C" int comic_gemm_group(const comic_gemm_prob* probs, int n, void* workspace, int64_t workspace_bytes, void* stream) {
  ComicGemmGroup g;
  if (int rc = gg_from_public(probs, n, g)) return rc;
  int64_t slab = 0;
  int nt = 0;
  const int wg = comic_gemm_group_plan(g, kPublicTarget, &slab, &nt);
  if (wg < 0) return 2;
  const int64_t slab_al = (slab + 255) / 256 * 256;
  COMIC_REQUIRE(nt == 0 || (workspace && workspace_bytes >= slab_al + (int64_t)nt * 4), "gemm_group: workspace too small");
  g.slab = (float*)workspace;
  g.tickets = nt ? (unsigned*)((char*)workspace + slab_al) : nullptr;
  if (nt) COMIC_REQUIRE(hipMemsetAsync(g.tickets, 0, (size_t)nt * 4, (hipStream_t)stream) == hipSuccess, "gemm_group: memset");
  return comic_gemm_group_launch(g, wg, (hipStream_t)stream);
}
