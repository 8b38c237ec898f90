// Exact-fp32 GEMM on gfx950's f32-input MFMA (v_mfma_f32_16x16x4_f32) for the decoder's
// dense layers.  Replaces the tf MatMul call-sites of common/ops.py:200-238 (linear),
// common/ops_rnn.py:440-447,:545 (memory/query layers), src/model_base.py:541-543 (output
// projection), :618-621 (LSTM kernel) and their autodiff transposes.
//
//   C[M,N] = alpha * op(A) * op(B) + beta * C + bias[N]
//
// Orientation: the MFMA A-operand is the B matrix (rows = n), the MFMA B-operand is the A
// matrix (cols = m), so a lane ends up with 4 consecutive n of one m (float4 store).
// Each operand tile is staged in LDS in its NATIVE global orientation (no transposes):
//   k-contiguous  [rows][16 k]  (+pad): fragment = one 16-byte read, element s used at step s
//   row-contiguous [16 k][rows] (+pad): fragment = four ds_read_b32 at k = 4*(lane>>4)+s
// Both forms give lane-group g the physical k = 4g+s at MFMA step s, so the k-sums agree.
#include <algorithm>

#include <stdlib.h>

#include "common.h"
#include "gemm_x3_dev.h"

namespace {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
  int k_per_slice = 0;   // bf16x3 kernel: > 0 = blockIdx.z owns k in [z*k_per_slice, ...) and writes a raw
  float* slab = nullptr; //   partial tile to slab[z][M][N] (combined by splitk_reduce_kernel)
  const int32_t* stop = nullptr;   // decode loops: see comic_stopped (common.h)
  int stop_t = 0;
};

constexpr int BK = 16;
constexpr int KC_ROW = 20;  // floats per LDS row, k-contiguous form (16 + 4 pad)

// ROWS = tile extent along the operand's non-k dimension (64 or 32)
template <int ROWS, bool KCONTIG>
struct Stage {
  static constexpr int RC_ROW = ROWS + 4;  // floats per k-row in the row-contiguous form
  static constexpr int LDS_FLOATS = KCONTIG ? ROWS * KC_ROW : BK * RC_ROW;
  static constexpr int CHUNKS = ROWS * BK / 4;  // 16-byte chunks per tile (256 or 128)

  // global -> register (one chunk per thread; threads >= CHUNKS idle)
  static __device__ __forceinline__ float4 load(const float* __restrict__ g, int ld, int row0, int rows_total, int k0,
                                                int K, int tid, bool vec_ok) {
    if (tid >= CHUNKS) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
      const int r = tid >> 2, c = tid & 3;
      const int row = row0 + r, k = k0 + c * 4;
      if (row >= rows_total || k >= K) return make_float4(0.f, 0.f, 0.f, 0.f);
      return load4(g + (size_t)row * ld + k, K - k, vec_ok);
    } else {
      constexpr int CPR = ROWS / 4;  // chunks per k-row
      const int kr = tid / CPR, c = tid % CPR;
      const int k = k0 + kr, row = row0 + c * 4;
      if (k >= K || row >= rows_total) return make_float4(0.f, 0.f, 0.f, 0.f);
      return load4(g + (size_t)k * ld + row, rows_total - row, vec_ok);
    }
  }
  static __device__ __forceinline__ void store(float* lds, float4 v, int tid) {
    if (tid >= CHUNKS) return;
    if (KCONTIG) {
      const int r = tid >> 2, c = tid & 3;
      *(float4*)(lds + r * KC_ROW + c * 4) = v;
    } else {
      constexpr int CPR = ROWS / 4;
      const int kr = tid / CPR, c = tid % CPR;
      *(float4*)(lds + kr * RC_ROW + c * 4) = v;
    }
  }
  // fragment of 16x16 sub-tile `t` (rows t*16 .. t*16+15): 4 values, one per MFMA k-step
  static __device__ __forceinline__ float4 frag(const float* lds, int rowbase, int lane) {
    const int r = rowbase + (lane & 15), g = lane >> 4;
    if (KCONTIG) return *(const float4*)(lds + r * KC_ROW + g * 4);
    return make_float4(lds[(4 * g + 0) * RC_ROW + r], lds[(4 * g + 1) * RC_ROW + r], lds[(4 * g + 2) * RC_ROW + r],
                       lds[(4 * g + 3) * RC_ROW + r]);
  }
};

template <int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  constexpr int BM = 64;
  using SA = Stage<BM, A_KC>;
  using SB = Stage<BN, B_KC>;
  constexpr int TNt = BN / 2 / 16;  // n sub-tiles per wave (2 wave columns)
  constexpr int TMt = 2;            // m sub-tiles per wave (2 wave rows x 32)
  __shared__ __attribute__((aligned(16))) float As[2][SA::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[2][SB::LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool a_vec = (a.lda % 4 == 0) && (((uintptr_t)a.A & 15) == 0) && (A_KC || true);
  const bool b_vec = (a.ldb % 4 == 0) && (((uintptr_t)a.B & 15) == 0);

  f32x4_t acc[TNt][TMt];
#pragma unroll
  for (int i = 0; i < TNt; ++i)
#pragma unroll
    for (int j = 0; j < TMt; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K + BK - 1) / BK;
  float4 ar = SA::load(a.A, a.lda, m0, a.M, 0, a.K, tid, a_vec);
  float4 br = SB::load(a.B, a.ldb, n0, a.N, 0, a.K, tid, b_vec);
  SA::store(As[0], ar, tid);
  SB::store(Bs[0], br, tid);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      ar = SA::load(a.A, a.lda, m0, a.M, (kt + 1) * BK, a.K, tid, a_vec);
      br = SB::load(a.B, a.ldb, n0, a.N, (kt + 1) * BK, a.K, tid, b_vec);
    }
    float4 bf[TNt], af[TMt];
#pragma unroll
    for (int i = 0; i < TNt; ++i) bf[i] = SB::frag(Bs[buf], wn * (BN / 2) + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TMt; ++j) af[j] = SA::frag(As[buf], wm * 32 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < TNt; ++i)
#pragma unroll
      for (int j = 0; j < TMt; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i].x, af[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i].y, af[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i].z, af[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i].w, af[j].w, acc[i][j], 0, 0, 0);
      }
    if (kt + 1 < nk) {
      SA::store(As[buf ^ 1], ar, tid);
      SB::store(Bs[buf ^ 1], br, tid);
    }
    __syncthreads();
  }

  // epilogue: lane holds n = nb + (lane>>4)*4 + {0..3}, m = mb + (lane&15)
  const bool c_vec = (a.ldc % 4 == 0) && (((uintptr_t)a.C & 15) == 0);
#pragma unroll
  for (int i = 0; i < TNt; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
    if (n >= a.N) continue;
    const int nv = min(4, a.N - n);
#pragma unroll
    for (int j = 0; j < TMt; ++j) {
      const int m = m0 + wm * 32 + j * 16 + (lane & 15);
      if (m >= a.M) continue;
      float* cp = a.C + (size_t)m * a.ldc + n;
      float v[4] = {acc[i][j][0] * a.alpha, acc[i][j][1] * a.alpha, acc[i][j][2] * a.alpha, acc[i][j][3] * a.alpha};
      if (a.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nv) v[q] += a.bias[n + q];
      }
      if (a.beta != 0.f) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nv) v[q] += a.beta * cp[q];
      }
      if (nv == 4 && c_vec) {
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nv) cp[q] = v[q];
      }
    }
  }
}


template <int BM, bool A_KC, bool B_KC, int BN = 128>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  constexpr int TM = BM / 32, TN = BN / 32;               // 16x16 tiles per wave (2 x 2 waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = a.k_per_slice > 0 ? blockIdx.z * a.k_per_slice : 0;
  const int kend = a.k_per_slice > 0 ? min(a.K, kbeg + a.k_per_slice) : a.K;
  f32x4_t acc[TN][TM];
  x3_mainloop<BM, A_KC, B_KC, BN>(a.A, a.B, a.M, a.N, a.lda, a.ldb, m0, n0, kbeg, kend, smem, acc);

  // epilogue: lane holds n = nb + (lane>>4)*4 + {0..3}, m = mb + (lane&15)
  const bool to_slab = a.k_per_slice > 0;
  float* Cbase = to_slab ? a.slab + (size_t)blockIdx.z * a.M * a.N : a.C;
  const int ldc = to_slab ? a.N : a.ldc;
  const bool c_vec = (ldc % 4 == 0) && (((uintptr_t)Cbase & 15) == 0);
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
    if (n >= a.N) continue;
    const int nv = min(4, a.N - n);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / 2) + j * 16 + (lane & 15);
      if (m >= a.M) continue;
      float* cp = Cbase + (size_t)m * ldc + n;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (!to_slab) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] *= a.alpha;
          if (a.bias && q < nv) v[q] += a.bias[n + q];
          if (a.beta != 0.f && q < nv) v[q] += a.beta * cp[q];
        }
      }
      if (nv == 4 && c_vec) {
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nv) cp[q] = v[q];
      }
    }
  }
}

template <int BM, int BN = 128>
int launch_x3(const GemmArgs& a, int ta, int tb, hipStream_t st) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), a.k_per_slice > 0 ? cdiv(a.K, a.k_per_slice) : 1);
  const bool a_kc = (ta == 0), b_kc = (tb == 1);
#define COMIC_X3(AK, BK_)                                                                                              \
  do {                                                                                                                 \
    constexpr int lds = 4 * (X3Tile<BM, AK>::BYTES + X3Tile<BN, BK_>::BYTES);                                          \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute((const void*)gemm_bf16x3_kernel<BM, AK, BK_, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              lds) != hipSuccess) {                                                                    \
        comic_set_error("gemm_bf16x3: cannot reserve %d bytes of LDS", lds);                                           \
        return 1;                                                                                                      \
      }                                                                                                                \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, AK, BK_, BN>), grid, dim3(256), lds, st, a);                                \
  } while (0)
  if (a_kc && b_kc) COMIC_X3(true, true);
  else if (a_kc && !b_kc) COMIC_X3(true, false);
  else if (!a_kc && b_kc) COMIC_X3(false, true);
  else COMIC_X3(false, false);
#undef COMIC_X3
  return 0;
}

// ---------------------------------------------------------------------------------------
// Skinny GEMM (M <= 64 per row block): the decoder's per-step products at batch 64 are
// weight-streaming problems (0.34 GFLOP over a 10.5 MB LSTM kernel), so the tiled kernel
// above (64 workgroups, one 6 KB tile in flight each) is latency-bound.  Here every wave
// streams its own operands straight into MFMA fragments (no LDS staging, no barriers in
// the k loop, next k-block prefetched in registers), the k range is split over the 4 waves
// of a workgroup (reduced through LDS in a fixed order -> deterministic) and, when the
// caller provides a workspace, additionally over gridDim.y workgroups (partials + a
// reduce/epilogue kernel) so that ~all 1024 SIMDs take part.
//
// A is [M,K] k-contiguous.  B_KC=false: B is [K,N] (n-contiguous): a lane's float4 along n
// feeds FOUR column-strided 16-wide tiles (tile j = columns n0+4c+j), so the 4 results a
// lane holds for one row are 4 consecutive columns (float4 store).  B_KC=true: B is [N,K]
// (k-contiguous): a lane's float4 along k is the 4 MFMA k-steps of tile j = columns
// n0+16j+c.
struct SkinnyArgs {
  const float* A;
  const float* B;
  float* C;          // output (S == 1) or partial buffer [S][M][N] (S > 1)
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
  int kb_per_slice;  // k16-blocks per gridDim.y slice
  int direct;        // 1: write alpha*acc + bias + beta*C to C ; 0: write raw partial
  const int32_t* stop = nullptr;   // decode loops: see comic_stopped (common.h)
  int stop_t = 0;
};

__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ base, long off, int nvalid, bool vec) {
  if (nvalid >= 4 && vec) return *(const float4*)(base + off);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = base[off];
  if (nvalid > 1) r.y = base[off + 1];
  if (nvalid > 2) r.z = base[off + 2];
  if (nvalid > 3) r.w = base[off + 3];
  return r;
}

template <bool B_KC>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  constexpr int MT = 4;
  __shared__ __attribute__((aligned(16))) float red[3 * 64 * 16 * MT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int m0 = blockIdx.z * 64;
  const int KB = (a.K + 15) >> 4;
  const int kb_begin = blockIdx.y * a.kb_per_slice;
  const int kb_end = min(KB, kb_begin + a.kb_per_slice);
  const bool a_vec = (a.lda % 4 == 0) && (((uintptr_t)a.A & 15) == 0);
  const bool b_vec = (a.ldb % 4 == 0) && (((uintptr_t)a.B & 15) == 0);

  f32x4_t acc[4][MT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto load_block = [&](int kb, float4* xa, float4* wb) {
    const int k = kb * 16 + 4 * g;
    const int kv = a.K - k;  // valid k elements from k on (may be <= 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + 16 * i + r;
      xa[i] = (m < a.M && kv > 0) ? ld4_guard(a.A, (long)m * a.lda + k, kv, a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (B_KC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + r;
        wb[j] = (n < a.N && kv > 0) ? ld4_guard(a.B, (long)n * a.ldb + k, kv, b_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int n = n0 + 4 * r;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wb[s] = (k + s < a.K && n < a.N) ? ld4_guard(a.B, (long)(k + s) * a.ldb + n, a.N - n, b_vec)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto mfma_block = [&](const float4* xa, const float4* wb) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float w;
        if (B_KC) {
          w = s == 0 ? wb[j].x : s == 1 ? wb[j].y : s == 2 ? wb[j].z : wb[j].w;
        } else {
          w = j == 0 ? wb[s].x : j == 1 ? wb[s].y : j == 2 ? wb[s].z : wb[s].w;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float x = s == 0 ? xa[i].x : s == 1 ? xa[i].y : s == 2 ? xa[i].z : xa[i].w;
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, acc[j][i], 0, 0, 0);
        }
      }
  };

  // this wave's k-blocks: kb_begin + wave, +4, ...   (register double buffering)
  float4 xa0[MT], wb0[4], xa1[MT], wb1[4];
  int kb = kb_begin + wave;
  if (kb < kb_end) load_block(kb, xa0, wb0);
  while (kb < kb_end) {
    const int kb1 = kb + 4;
    if (kb1 < kb_end) load_block(kb1, xa1, wb1);
    mfma_block(xa0, wb0);
    if (kb1 >= kb_end) break;
    const int kb2 = kb1 + 4;
    if (kb2 < kb_end) load_block(kb2, xa0, wb0);
    mfma_block(xa1, wb1);
    kb = kb2;
  }

  // fixed-order reduction over the 4 waves: waves 1..3 park their tiles, wave 0 adds them
  if (wave > 0) {
    float* dst = red + (size_t)(wave - 1) * 64 * 16 * MT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) *(float4*)(dst + ((j * MT + i) * 64 + lane) * 4) = *(float4*)&acc[j][i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float* src = red + (size_t)w * 64 * 16 * MT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4 t = *(const float4*)(src + ((j * MT + i) * 64 + lane) * 4);
        acc[j][i][0] += t.x; acc[j][i][1] += t.y; acc[j][i][2] += t.z; acc[j][i][3] += t.w;
      }
  }
  // store: lane holds, for output row m = m0+16i+r, tile-column index cidx = 4g+q
  float* Cout = a.direct ? a.C : a.C + (size_t)blockIdx.y * a.M * a.ldc;
  const bool c_vec = (a.ldc % 4 == 0) && (((uintptr_t)Cout & 15) == 0);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + 16 * i + r;
    if (m >= a.M) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // four consecutive columns: NN -> cidx = 4g+u over tiles j=0..3 ; NT -> tile j=u, q=0..3
      int n;
      float v[4];
      if (B_KC) {
        n = n0 + 16 * u + 4 * g;
        v[0] = acc[u][i][0]; v[1] = acc[u][i][1]; v[2] = acc[u][i][2]; v[3] = acc[u][i][3];
      } else {
        n = n0 + 4 * (4 * g + u);
        v[0] = acc[0][i][u]; v[1] = acc[1][i][u]; v[2] = acc[2][i][u]; v[3] = acc[3][i][u];
      }
      if (n >= a.N) continue;
      const int nv = min(4, a.N - n);
      float* cp = Cout + (size_t)m * a.ldc + n;
      if (a.direct) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] *= a.alpha;
          if (a.bias && q < nv) v[q] += a.bias[n + q];
          if (a.beta != 0.f && q < nv) v[q] += a.beta * cp[q];
        }
      }
      if (nv == 4 && c_vec) {
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int q = 0; q < nv; ++q) cp[q] = v[q];
      }
    }
  }
}

// C = alpha * sum_s P[s] + bias + beta*C
__global__ void splitk_reduce_kernel(const float* __restrict__ P, float* __restrict__ C, const float* __restrict__ bias,
                                     int M, int N, int ldc, int S, float alpha, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += P[(size_t)k * M * N + i];
  s *= alpha;
  if (bias) s += bias[n];
  float* cp = C + (size_t)m * ldc + n;
  if (beta != 0.f) s += beta * *cp;
  *cp = s;
}

template <int BN>
void launch(const GemmArgs& a, int ta, int tb, hipStream_t st) {
  dim3 grid(cdiv(a.M, 64), cdiv(a.N, BN));
  const bool a_kc = (ta == 0), b_kc = (tb == 1);
  if (a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<BN, true, true>), grid, dim3(256), 0, st, a);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<BN, true, false>), grid, dim3(256), 0, st, a);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL((gemm_f32_kernel<BN, false, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BN, false, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

// Skinny product left as split-K partials: ws receives [S][M][N] (ld = N) raw partial sums and
// *S_out their count; the consumer kernel adds them up (saves a reduce launch per GEMM).
int comic_gemm_f32_partial(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int trans_b,
                           void* ws, int64_t ws_bytes, int* S_out, hipStream_t st) {
  COMIC_REQUIRE(A && B && ws && S_out, "gemm_partial: null pointer");
  COMIC_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldb >= (trans_b ? K : N), "gemm_partial: bad shape");
  const int KB = (K + 15) / 16, NB = cdiv(N, 64), MB = cdiv(M, 64);
  int S = std::max(1, std::min(std::min(256 / std::max(1, NB * MB), KB / 8), 32));
  while (S > 1 && (int64_t)S * M * N * 4 > ws_bytes) --S;
  COMIC_REQUIRE((int64_t)S * M * N * 4 <= ws_bytes, "gemm_partial: workspace too small");
  SkinnyArgs a{A, B, (float*)ws, nullptr, M, N, K, lda, ldb, N, 1.f, 0.f, cdiv(KB, S), 0};
  a.stop = g_comic_stop.p;
  a.stop_t = g_comic_stop.t;
  S = cdiv(KB, a.kb_per_slice);
  dim3 grid(NB, S, MB);
  if (trans_b)
    hipLaunchKernelGGL((gemm_skinny_kernel<true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<false>), grid, dim3(256), 0, st, a);
  COMIC_LAUNCH_CHECK("gemm_skinny(partial)");
  *S_out = S;
  return 0;
}

int comic_gemm_f32_ws(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                      int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta, void* ws, int64_t ws_bytes,
                      hipStream_t st) {
  COMIC_REQUIRE(A && B && C, "gemm: null pointer");
  COMIC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape %d %d %d", M, N, K);
  COMIC_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm: leading dimension too small");
  if (!trans_a && M <= 2048) {
    // skinny path
    const int KB = (K + 15) / 16, NB = cdiv(N, 64), MB = cdiv(M, 64);
    int S = 1;
    if (ws) {
      S = std::max(1, std::min(std::min(256 / std::max(1, NB * MB), KB / 8), 32));
      while (S > 1 && (int64_t)S * M * N * 4 > ws_bytes) --S;
    }
    SkinnyArgs a{A, B, S > 1 ? (float*)ws : C, bias, M, N, K, lda, ldb, S > 1 ? N : ldc, alpha, beta, cdiv(KB, S),
                 S > 1 ? 0 : 1};
    S = cdiv(KB, a.kb_per_slice);
    dim3 grid(NB, S, MB);
    if (trans_b)
      hipLaunchKernelGGL((gemm_skinny_kernel<true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((gemm_skinny_kernel<false>), grid, dim3(256), 0, st, a);
    if (a.direct == 0) {
      const long total = (long)M * N;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (const float*)ws,
                         C, bias, M, N, ldc, S, alpha, beta);
    }
    COMIC_LAUNCH_CHECK("gemm_skinny");
    return 0;
  }
  GemmArgs a{A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta};
  const long blocks64 = (long)cdiv(M, 64) * cdiv(N, 64);
  if (blocks64 >= 192)
    launch<64>(a, trans_a, trans_b, st);
  else
    launch<32>(a, trans_a, trans_b, st);
  COMIC_LAUNCH_CHECK("gemm_f32");
  return 0;
}

// C = alpha * op(A) * op(B) + beta * C + bias with 3-term bf16-split products (see gemm_bf16x3_kernel).
// With a workspace the k range is split over gridDim.z workgroups when there are too few output tiles to
// fill the chip (weight gradients: K = T*B rows, small M x N); the raw partial slabs are combined in a
// fixed order (deterministic).
int comic_gemm_bf16x3_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta, void* ws,
                           int64_t ws_bytes, hipStream_t st) {
  COMIC_REQUIRE(A && B && C, "gemm_bf16x3: null pointer");
  COMIC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_bf16x3: bad shape %d %d %d", M, N, K);
  COMIC_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm_bf16x3: leading dimension too small");
  GemmArgs a{A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta};
  a.stop = g_comic_stop.p;
  a.stop_t = g_comic_stop.t;
  const long blocks128 = (long)cdiv(M, 128) * cdiv(N, 128);
  constexpr int big_min = 40;  // 128-row tiles from this many 128 x 128 output blocks on (measured on the decoder step's
                               // products: 314 -> 288 us per step against 200)
  const bool big = blocks128 >= big_min;
  const long tiles = big ? blocks128 : (long)cdiv(M, 64) * cdiv(N, 128);
  int S = 1;
  if (ws && tiles < 512) {
    S = (int)std::min<long>(std::min<long>(16, 768 / tiles), K / 128);
    while (S > 1 && (int64_t)S * M * N * 4 > ws_bytes) --S;
    S = std::max(S, 1);
  }
  if (S > 1) {
    a.k_per_slice = cdiv(cdiv(K, S), 32) * 32;
    a.slab = (float*)ws;
    S = cdiv(K, a.k_per_slice);
  }
  // decode-step shapes (beam rows x hidden x vocabulary): a tile as tall as ALL rows streams the big operand once
  const bool tall = S == 1 && M > 128 && M <= 192 && cdiv(N, 128) >= 64;
  int rc;
  // ... in 64-column tiles: twice the workgroups (two per CU) hide each other's load latency (PMC: a wave of the
  // 128-column form spent half its life in s_waitcnt at one wave per SIMD)
  if (tall && M <= 160) rc = launch_x3<160, 64>(a, trans_a, trans_b, st);
  else if (tall) rc = launch_x3<192, 64>(a, trans_a, trans_b, st);
  else rc = big ? launch_x3<128>(a, trans_a, trans_b, st) : launch_x3<64>(a, trans_a, trans_b, st);
  if (rc) return rc;
  if (S > 1) {
    const long total = (long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, (const float*)ws, C,
                       bias, M, N, ldc, S, alpha, beta);
  }
  COMIC_LAUNCH_CHECK("gemm_bf16x3");
  return 0;
}

extern "C" int comic_gemm_f32_split3(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                     int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  return comic_gemm_bf16x3_impl(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, beta, workspace,
                                workspace_bytes, (hipStream_t)stream);
}

extern "C" int comic_gemm_f32_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                     int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  return comic_gemm_f32_ws(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, beta, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

extern "C" int comic_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                              int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta,
                              void* stream) {
  return comic_gemm_f32_ws(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, alpha, beta, nullptr, 0,
                           (hipStream_t)stream);
}
