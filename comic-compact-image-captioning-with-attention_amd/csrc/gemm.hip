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
#include "common.h"

namespace {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
};

constexpr int BK = 16;
constexpr int KC_ROW = 20;  // floats per LDS row, k-contiguous form (16 + 4 pad)

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

extern "C" int comic_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                              int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta,
                              void* stream) {
  COMIC_REQUIRE(A && B && C, "gemm: null pointer");
  COMIC_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape %d %d %d", M, N, K);
  COMIC_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "gemm: leading dimension too small");
  GemmArgs a{A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta};
  hipStream_t st = (hipStream_t)stream;
  const long blocks64 = (long)cdiv(M, 64) * cdiv(N, 64);
  if (blocks64 >= 192)
    launch<64>(a, trans_a, trans_b, st);
  else
    launch<32>(a, trans_a, trans_b, st);
  COMIC_LAUNCH_CHECK("gemm_f32");
  return 0;
}
