// Device pieces of the hi/lo-split bf16 GEMM (fp32 operands, three v_mfma_f32_16x16x32_bf16 per product), shared by
// gemm.hip (single-problem launches) and gemm_group.hip (grouped launches with in-kernel split-K combine).
#pragma once
#include <type_traits>

#include "common.h"

namespace {

// 4 floats from p[0..3] with element-wise validity
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int nvalid, bool vec_ok) {
  if (nvalid >= 4 && vec_ok) return *(const float4*)p;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

// ---------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores: every operand element x is split on the fly into
// hi = bf16(x), lo = bf16(x - hi) and the product is accumulated as hi*hi + hi*lo + lo*hi in fp32
// (the dropped lo*lo term and the 16 kept mantissa bits bound the relative error of a product by
// ~2^-15; measured 2e-6 on the decoder's shapes).  v_mfma_f32_16x16x32_bf16 does 16 384 FLOP in 16
// cycles against 2 048 in 32 for the f32-input MFMA, so three of them are ~5x faster than the exact
// path.  Used by the decoder executors for the time-batched products (keys, logits, every weight
// gradient); the per-step products and the public comic_gemm_f32 keep exact fp32 products.
// Tiles: BM x 128 x 32, operands converted while they are staged into k-contiguous LDS rows of
// 32 bf16 (+16 B pad: ds_read_b128 fragments, conflict-free), register-prefetched double buffer.
// (a, b) -> packed hi pair, packed lo pair
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}

// LDS images of a ROWS x 32 operand tile (bf16, one for hi and one for lo):
//   k-contiguous operand  -> [ROWS][32 k] rows of 64 B + 16 B pad, read with ds_read_b128; the 8-byte
//                            chunk holding physical k = 4c..4c+3 sits at chunk position 2*(c&3) + (c>>2)
//   row-contiguous operand -> [32 k][ROWS] rows of 2*ROWS + 32 B, read with ds_read_b64_tr_b16
// Both give lane group g the physical k {4g..4g+3, 16+4g..16+4g+3} as its 8 MFMA k-values (the
// conflict-free transposing-read order, see conv_wgrad_tr_kernel).
template <int ROWS, bool KC>
struct X3Tile {
  static constexpr int ROWB = 80;
  static constexpr int KSTR = 2 * ROWS + 32;
  static constexpr int BYTES = KC ? ROWS * ROWB : 32 * KSTR;
};

// chunk q of a ROWS x 32 fp32 tile.  k-contiguous operand: row = q / 8, k = (q % 8) * 4;
// row-contiguous operand: k = q / (ROWS/4), rows (q % (ROWS/4)) * 4 .. +3.
// Every load is UNCONDITIONAL at a clamped address and the value is selected afterwards: behind a per-lane condition hipcc
// waits for each load where the branches join (s_waitcnt vmcnt(0)), i.e. the two-tiles-deep register prefetch of the main
// loop paid the full memory latency eight times per k-tile (3.3 us per k-tile measured in the grouped launch).
// `vec`: rows are 16-byte aligned AND the extent of the contiguous dimension (K or rows_total) is a multiple of 4, so a
// chunk is either whole or empty.
// index of chunk q along the contiguous dimension (c), its extent, valid elements from c on (nv), row base offset
template <int ROWS, bool KC>
__device__ __forceinline__ void x3_chunk(int q, int ld, int row0, int rows_total, int k0, int K, int& c, int& ext, int& nv,
                                         size_t& base) {
  int r, k;
  if (KC) {
    r = row0 + (q >> 3); k = k0 + (q & 7) * 4;
    c = k; ext = K; nv = r < rows_total ? K - k : 0;
    base = (size_t)min(r, rows_total - 1) * ld;
  } else {
    constexpr int CPR = ROWS / 4;
    k = k0 + q / CPR; r = row0 + (q % CPR) * 4;
    c = r; ext = rows_total; nv = k < K ? rows_total - r : 0;
    base = (size_t)min(k, K - 1) * ld;
  }
}
// raw loads at clamped addresses (x3_mask_tile zeroes what lies outside the operand, later: at the LDS store)
template <int ROWS, bool KC, int NCH, bool VEC>
__device__ __forceinline__ void x3_load_tile(const float* __restrict__ g, int ld, int row0, int rows_total, int k0, int K,
                                             int tid, float4 (&out)[NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c, ext, nv;
    size_t base;
    x3_chunk<ROWS, KC>(tid + 256 * i, ld, row0, rows_total, k0, K, c, ext, nv, base);
    const float* bp = g + base;
    if (VEC) {
      out[i] = *(const float4*)(bp + max(0, min(c, ext - 4)));
    } else {
      const int e = ext - 1;
      out[i] = make_float4(bp[min(c, e)], bp[min(c + 1, e)], bp[min(c + 2, e)], bp[min(c + 3, e)]);
    }
  }
}
template <int ROWS, bool KC, int NCH, bool VEC>
__device__ __forceinline__ void x3_mask_tile(int row0, int rows_total, int k0, int K, int tid, float4 (&v)[NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int c, ext, nv;
    size_t base;
    x3_chunk<ROWS, KC>(tid + 256 * i, 0, row0, rows_total, k0, K, c, ext, nv, base);
    if (VEC) {
      const bool ok = nv >= 4;
      v[i].x = ok ? v[i].x : 0.f; v[i].y = ok ? v[i].y : 0.f; v[i].z = ok ? v[i].z : 0.f; v[i].w = ok ? v[i].w : 0.f;
    } else {
      v[i].x = nv > 0 ? v[i].x : 0.f; v[i].y = nv > 1 ? v[i].y : 0.f; v[i].z = nv > 2 ? v[i].z : 0.f; v[i].w = nv > 3 ? v[i].w : 0.f;
    }
  }
}
template <int ROWS, bool KC, int NCH>
__device__ __forceinline__ void x3_store_tile(unsigned char* hi_base, unsigned char* lo_base, int tid,
                                              const float4 (&in)[NCH]) {
  using T = X3Tile<ROWS, KC>;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = tid + 256 * i;
    uint32_t h0, l0, h1, l1;
    split_bf16x2(in[i].x, in[i].y, h0, l0);
    split_bf16x2(in[i].z, in[i].w, h1, l1);
    int off;
    if (KC) {
      const int row = q >> 3, c = q & 7;
      off = row * T::ROWB + (2 * (c & 3) + (c >> 2)) * 8;
    } else {
      constexpr int CPR = ROWS / 4;
      const int kk = q / CPR, row = (q % CPR) * 4;
      off = kk * T::KSTR + row * 2;
    }
    *(uint2*)(hi_base + off) = make_uint2(h0, h1);
    *(uint2*)(lo_base + off) = make_uint2(l0, l1);
  }
}
typedef __attribute__((ext_vector_type(4))) short x3_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short x3_s16x8_t;
// 16x32 MFMA fragment of the 16 operand rows starting at `row16`
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8_t x3_frag(const unsigned char* base, int row16, int lane) {
  using T = X3Tile<ROWS, KC>;
  if (KC) {
    const uint4 v = *(const uint4*)(base + (row16 + (lane & 15)) * T::ROWB + (lane >> 4) * 16);
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned char* p = base + (4 * g + q) * T::KSTR + (row16 + 4 * pp) * 2;
    const uint32_t a0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p);
    const x3_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4_t*)(uintptr_t)a0);
    const x3_s16x4_t hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4_t*)(uintptr_t)(a0 + 16 * T::KSTR));
    const x3_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}


// Source of an operand's k-tiles.  VEC (rows 16-byte aligned; for a k-contiguous operand also K % 4 == 0): one raw buffer
// load per 16-byte chunk, and what lies outside the operand comes back as zero from the buffer's own range check -- rows
// past the slice (row-contiguous operand: the buffer ends with row kend - 1) and chunks whose row / column is outside
// (their offset is parked at kOor) cost no instruction; a k-contiguous operand pays one compare per tile for "k >= kend".
// Columns between rows_total and the next multiple of 4 of a row-contiguous operand are READ (ld % 4 == 0 keeps that
// inside the row) and feed output columns >= N, which the epilogue drops.  !VEC: scalar loads at clamped addresses, masked
// when the tile is stored to LDS.
constexpr unsigned kOor = 0x80000000u;
template <int ROWS, bool KC, int NCH, bool VEC>
struct X3Src {
  const float* g;
  int ld, row0, rows_total, kend;
  __amdgpu_buffer_rsrc_t rs;
  unsigned base;          // !KC: byte offset of this thread's first chunk of the current tile
  unsigned rb[NCH];       // KC: byte offset of chunk i at k0 = 0 (kOor when its row is outside the operand)
  int kc;                 // KC: k offset of this thread's chunks inside a tile
  unsigned step8;         // !KC: bytes between two chunks of this thread (256 / (ROWS / 4) k rows)
  __device__ __forceinline__ void init(const float* g_, int ld_, int row0_, int rows_total_, int kbeg, int kend_, int tid) {
    g = g_; ld = ld_; row0 = row0_; rows_total = rows_total_; kend = kend_;
    if (VEC) {
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7FFFFFF0, 0x00020000);
      if (KC) {
        kc = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {               // chunk i: row row0 + (tid >> 3) + 32 i
          const int r = row0 + (tid >> 3) + 32 * i;
          rb[i] = r < rows_total ? (unsigned)(((long)r * ld + kc) * 4) : kOor;
        }
      } else {
        constexpr int CPR = ROWS / 4;                 // chunks per k row; chunk q = tid + 256 i: k row q / CPR, columns (q % CPR) * 4
        const int r = row0 + (tid % CPR) * 4;
        const long rec = ((long)(kend - 1) * ld + min(ld, (rows_total + 3) & ~3)) * 4;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)rec, 0x00020000);
        base = r < rows_total ? (unsigned)(((long)(kbeg + tid / CPR) * ld + r) * 4) : kOor;
        step8 = (unsigned)((256 / CPR) * ld * 4);
      }
    }
  }
  // tile starting at k0 (tiles must be requested in order, one call per tile, for the row-contiguous form)
  __device__ __forceinline__ void load(int k0, int tid, float4 (&out)[NCH]) {
    if (VEC) {
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      if (KC) {
        // k0 + kc >= kend -> bit 31 set (pure arithmetic: with a select here hipcc branches around two copies of the load
        // and waits vmcnt(0) in between)
        const unsigned past = (unsigned)((kend - 1 - (k0 + kc)) >> 31) & kOor;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const unsigned o = (rb[i] + (unsigned)k0 * 4u) | past;
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0);
          out[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      } else if constexpr (256 % (ROWS / 4) == 0) {   // a thread's chunks share their columns: one running offset
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + (unsigned)i * step8), 0, 0);
          out[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
        base += (unsigned)(32 * ld * 4);          // next tile: 32 k rows on (kOor + the whole matrix stays below 2^32)
      } else {
        constexpr int CPR = ROWS / 4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int q = tid + 256 * i, r = row0 + (q % CPR) * 4;
          const unsigned o = r < rows_total ? (unsigned)(((long)(k0 + q / CPR) * ld + r) * 4) : kOor;
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)o, 0, 0);
          out[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      }
    } else {
      x3_load_tile<ROWS, KC, NCH, false>(g, ld, row0, rows_total, k0, kend, tid, out);
    }
  }
  __device__ __forceinline__ void mask(int k0, int tid, float4 (&v)[NCH]) {
    if (!VEC) x3_mask_tile<ROWS, KC, NCH, false>(row0, rows_total, k0, kend, tid, v);
  }
};

// Main loop of one BM x BN output tile over k in [kbeg, kend): 2 x 2 waves, a wave owns (BM/32) x (BN/32) 16 x 16 tiles.
// Register staging TWO k-tiles deep: tile t travels in register set t & 1; at the top of iteration kt the load of tile
// kt+2 is issued into the set tile kt has just left, so a global load has two compute phases to land.
// acc[i][j]: n sub-tile i, m sub-tile j; a lane holds n = n0 + wn*(BN/2) + i*16 + (lane>>4)*4 + {0..3}, m = m0 + wm*(BM/2) + j*16 + (lane&15).
template <int BM, bool A_KC, bool B_KC, int BN, bool VEC>
__device__ __forceinline__ void x3_mainloop_v(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int lda,
                                            int ldb, int m0, int n0, int kbeg, int kend, unsigned char* smem,
                                            f32x4_t (&acc)[BN / 32][BM / 32]) {
  constexpr int BKx = 32;
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int ACH = BM * BKx / 4 / 256, BCH = BN * BKx / 4 / 256;   // float4 chunks per thread
  constexpr int ABYTES = X3Tile<BM, A_KC>::BYTES, BBYTES = X3Tile<BN, B_KC>::BYTES;
  unsigned char* Ah = smem;                              // [2] tile images each
  unsigned char* Al = Ah + 2 * ABYTES;
  unsigned char* Bh = Al + 2 * ABYTES;
  unsigned char* Bl = Bh + 2 * BBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float4 ar[2][ACH], br[2][BCH];
  // The loop body is straight-line code (no per-tile conditions): the k-tile count is rounded up to an even number and
  // tiles past kend are loaded from clamped addresses and masked to zero.  With conditional loads / stores inside the loop
  // hipcc's wait-count pass merged the pending-load state at the joins and waited for every load right where it was issued.
  const int nk = ((kend - kbeg + BKx - 1) / BKx + 1) & ~1;
  X3Src<BM, A_KC, ACH, VEC> sa;
  X3Src<BN, B_KC, BCH, VEC> sb;
  sa.init(A, lda, m0, M, kbeg, kend, tid);
  sb.init(B, ldb, n0, N, kbeg, kend, tid);
  sa.load(kbeg, tid, ar[0]);
  sb.load(kbeg, tid, br[0]);
  sa.load(kbeg + BKx, tid, ar[1]);
  sb.load(kbeg + BKx, tid, br[1]);
  sa.mask(kbeg, tid, ar[0]);
  sb.mask(kbeg, tid, br[0]);
  x3_store_tile<BM, A_KC, ACH>(Ah, Al, tid, ar[0]);
  x3_store_tile<BN, B_KC, BCH>(Bh, Bl, tid, br[0]);
  __syncthreads();
  auto step = [&](const int kt, auto par) {
    constexpr int P = decltype(par)::value;          // kt & 1: LDS buffer of tile kt, register set of tile kt + 2
    sa.load(kbeg + (kt + 2) * BKx, tid, ar[P]);
    sb.load(kbeg + (kt + 2) * BKx, tid, br[P]);
    bf16x8_t bh[TN], bl[TN], ah[TM], al[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      bh[i] = x3_frag<BN, B_KC>(Bh + P * BBYTES, wn * (BN / 2) + i * 16, lane);
      bl[i] = x3_frag<BN, B_KC>(Bl + P * BBYTES, wn * (BN / 2) + i * 16, lane);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      ah[j] = x3_frag<BM, A_KC>(Ah + P * ABYTES, wm * (BM / 2) + j * 16, lane);
      al[j] = x3_frag<BM, A_KC>(Al + P * ABYTES, wm * (BM / 2) + j * 16, lane);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[i], ah[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[i], al[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[i], ah[j], acc[i][j], 0, 0, 0);
      }
    // tile kt + 1 (register set P ^ 1, loaded one iteration ago) -> the other LDS buffer
    sa.mask(kbeg + (kt + 1) * BKx, tid, ar[P ^ 1]);
    sb.mask(kbeg + (kt + 1) * BKx, tid, br[P ^ 1]);
    x3_store_tile<BM, A_KC, ACH>(Ah + (P ^ 1) * ABYTES, Al + (P ^ 1) * ABYTES, tid, ar[P ^ 1]);
    x3_store_tile<BN, B_KC, BCH>(Bh + (P ^ 1) * BBYTES, Bl + (P ^ 1) * BBYTES, tid, br[P ^ 1]);
#ifdef COMIC_X3_SCHED
    // Issue order of the step: loads, fragment reads, then the conversion of tile kt + 1 (VALU + LDS stores, independent
    // of this tile's products) threaded between the MFMAs instead of behind them: a wave alone on its SIMD otherwise runs
    // MFMA (768 cycles), conversion (~500) and stores (~200) one after the other.
    __builtin_amdgcn_sched_group_barrier(0x020, ACH + BCH, 0);          // VMEM reads
    __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);                 // DS reads (as many as there are)
#pragma unroll
    for (int r = 0; r < TM * TN; ++r) {
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);                // VALU
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                // DS write
    }
#endif
    __syncthreads();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>());
    step(kt + 1, std::integral_constant<int, 1>());
  }
}

// 16-byte loads when both operands allow them (aligned rows, contiguous extents multiples of 4), scalar loads otherwise
template <int BM, bool A_KC, bool B_KC, int BN>
__device__ __forceinline__ void x3_mainloop(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int lda,
                                            int ldb, int m0, int n0, int kbeg, int kend, unsigned char* smem,
                                            f32x4_t (&acc)[BN / 32][BM / 32]) {
  const bool a_vec = (lda % 4 == 0) && (((uintptr_t)A & 15) == 0) && (!A_KC || kend % 4 == 0);
  const bool b_vec = (ldb % 4 == 0) && (((uintptr_t)B & 15) == 0) && (!B_KC || kend % 4 == 0);
  if (a_vec && b_vec) x3_mainloop_v<BM, A_KC, B_KC, BN, true>(A, B, M, N, lda, ldb, m0, n0, kbeg, kend, smem, acc);
  else x3_mainloop_v<BM, A_KC, B_KC, BN, false>(A, B, M, N, lda, ldb, m0, n0, kbeg, kend, smem, acc);
}

// ---- producer / consumer form of the main loop (grouped launches; 512 threads, 16-byte loads only) -----------------------
// In the form above all four waves of a workgroup do the same thing at the same time -- load, wait, convert, store, barrier,
// read fragments, multiply -- so a k-tile costs the SUM of those phases (about 2000 cycles alone on a CU, 2.5 us per k-tile
// with two workgroups per CU: 20 % of the matrix peak on d K).  Here the roles are split (the loader-wave idea of the
// convolution kernels): waves 0-3 own the 2 x 2 output quadrants and only read fragments and issue MFMAs; waves 4-7 only
// stream: buffer loads four k-tiles deep into four register sets, hi / lo split, LDS stores into the other of two stages.
// One workgroup barrier per k-tile: behind it tile kt + 1 is complete and stage kt % 2 is free again.  The k-contiguous LDS
// image has 64-byte rows (no pad) and the 16-byte chunk j of row r sits at j ^ swz(r >> 2), swz = {0, 3, 2, 1}: the lane
// groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) then hit sixteen distinct bank quads (the padded 80-byte rows gave
// 33 % conflict cycles by counters).
template <int ROWS, bool KC>
struct X3TilePc {
  static constexpr int KSTR = 2 * ROWS + 32;
  static constexpr int BYTES = KC ? ROWS * 64 : 32 * KSTR;
};
__device__ __forceinline__ int x3_swz(int row) { return (0x6C >> (((row >> 2) & 3) * 2)) & 3; }   // {0, 3, 2, 1}

template <int ROWS, bool KC, int NCH>
__device__ __forceinline__ void x3pc_store_tile(unsigned char* hi_base, unsigned char* lo_base, int tid,
                                                const float4 (&in)[NCH]) {
  using T = X3TilePc<ROWS, KC>;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int q = tid + 256 * i;
    uint32_t h0, l0, h1, l1;
    split_bf16x2(in[i].x, in[i].y, h0, l0);
    split_bf16x2(in[i].z, in[i].w, h1, l1);
    int off;
    if (KC) {       // 8-byte piece c (k = 4c .. 4c+3) -> chunk c & 3, half c >> 2 (lane group g multiplies k {4g.., 16+4g..})
      const int row = q >> 3, c = q & 7;
      off = row * 64 + (((c & 3) ^ x3_swz(row)) * 2 + (c >> 2)) * 8;
    } else {
      constexpr int CPR = ROWS / 4;
      const int kk = q / CPR, row = (q % CPR) * 4;
      off = kk * T::KSTR + row * 2;
    }
    *(uint2*)(hi_base + off) = make_uint2(h0, h1);
    *(uint2*)(lo_base + off) = make_uint2(l0, l1);
  }
}
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8_t x3pc_frag(const unsigned char* base, int row16, int lane) {
  using T = X3TilePc<ROWS, KC>;
  if (KC) {
    const int row = row16 + (lane & 15);
    const uint4 v = *(const uint4*)(base + row * 64 + (((lane >> 4) ^ x3_swz(row)) * 16));
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned char* p = base + (4 * g + q) * T::KSTR + (row16 + 4 * pp) * 2;
    const uint32_t a0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const unsigned char*)p);
    const x3_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4_t*)(uintptr_t)a0);
    const x3_s16x4_t hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4_t*)(uintptr_t)(a0 + 16 * T::KSTR));
    const x3_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// 128 x 128 tile, 512 threads.  acc is meaningful in waves 0-3 only (quadrant wm = wave >> 1, wn = wave & 1, layout as above).
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void x3_mainloop_pc(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int lda,
                                               int ldb, int m0, int n0, int kbeg, int kend, unsigned char* smem,
                                               f32x4_t (&acc)[4][4]) {
  constexpr int BM = 128, BN = 128, BKx = 32, NCH = 4;
  constexpr int ABYTES = X3TilePc<BM, A_KC>::BYTES, BBYTES = X3TilePc<BN, B_KC>::BYTES;
  constexpr int STAGE = 2 * (ABYTES + BBYTES);          // A hi | A lo | B hi | B lo
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // k-tiles, rounded up to a multiple of 4 (the register sets rotate with period 4; tiles past kend load zeros)
  const int nk = ((kend - kbeg + BKx - 1) / BKx + 3) & ~3;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if (wave >= 4) {
    // ---------------------------------------------------------------- producers ------------
    const int pt = tid - 256;
    X3Src<BM, A_KC, NCH, true> sa;
    X3Src<BN, B_KC, NCH, true> sb;
    sa.init(A, lda, m0, M, kbeg, kend, pt);
    sb.init(B, ldb, n0, N, kbeg, kend, pt);
    float4 ar[4][NCH], br[4][NCH];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sa.load(kbeg + s * BKx, pt, ar[s]);
      sb.load(kbeg + s * BKx, pt, br[s]);
    }
    auto put = [&](int stage, const float4 (&a)[NCH], const float4 (&b)[NCH]) {
      unsigned char* st = smem + stage * STAGE;
      x3pc_store_tile<BM, A_KC, NCH>(st, st + ABYTES, pt, a);
      x3pc_store_tile<BN, B_KC, NCH>(st + 2 * ABYTES, st + 2 * ABYTES + BBYTES, pt, b);
    };
    put(0, ar[0], br[0]);
    sa.load(kbeg + 4 * BKx, pt, ar[0]);
    sb.load(kbeg + 4 * BKx, pt, br[0]);
    __syncthreads();
    // iteration kt: tile kt + 1 (set (kt + 1) % 4) -> stage (kt + 1) % 2, then tile kt + 5 is requested into that set
    auto step = [&](int kt, auto s_) {
      constexpr int S = decltype(s_)::value;             // (kt + 1) % 4
      put(S & 1, ar[S], br[S]);
      sa.load(kbeg + (kt + 5) * BKx, pt, ar[S]);
      sb.load(kbeg + (kt + 5) * BKx, pt, br[S]);
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 4) {
      step(kt, std::integral_constant<int, 1>());
      step(kt + 1, std::integral_constant<int, 2>());
      step(kt + 2, std::integral_constant<int, 3>());
      step(kt + 3, std::integral_constant<int, 0>());
    }
  } else {
    // ---------------------------------------------------------------- consumers ------------
    const int wm = wave >> 1, wn = wave & 1;
    __syncthreads();
    auto step = [&](auto p_) {
      constexpr int P = decltype(p_)::value;             // kt % 2
      const unsigned char* st = smem + P * STAGE;
      bf16x8_t bh[4], bl[4], ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bh[i] = x3pc_frag<BN, B_KC>(st + 2 * ABYTES, wn * 64 + i * 16, lane);
        bl[i] = x3pc_frag<BN, B_KC>(st + 2 * ABYTES + BBYTES, wn * 64 + i * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ah[j] = x3pc_frag<BM, A_KC>(st, wm * 64 + j * 16, lane);
        al[j] = x3pc_frag<BM, A_KC>(st + ABYTES, wm * 64 + j * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[i], ah[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[i], al[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[i], ah[j], acc[i][j], 0, 0, 0);
        }
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      step(std::integral_constant<int, 0>());
      step(std::integral_constant<int, 1>());
    }
  }
}

}  // namespace
