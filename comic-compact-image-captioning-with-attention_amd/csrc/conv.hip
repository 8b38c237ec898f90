// CNN encoder kernels for gfx950: implicit-GEMM convolution on MFMA with a folded
// BatchNorm + ReLU epilogue, the Cin=3 stem convolution, and the pooling kernels.
//
// Replaces the TF-1.9 op call-sites of slim.conv2d / slim.max_pool2d / slim.avg_pool2d in
// common/nets/inception_v3.py:100-415 under inception_arg_scope
// (common/nets/inception_utils.py:32-82): Conv2D (no bias) -> FusedBatchNorm(inference,
// no gamma, eps 1e-3) -> Relu, NHWC.
//
// Implicit GEMM, "swapped" orientation so that the accumulator holds 4 consecutive output
// channels of one pixel per lane (one 8-byte bf16 store):
//     D[n][m] = sum_k  Wp[n][k] * im2col(X)[m][k]        n = cout, m = (b,ho,wo), k = (kh,kw,cin)
// MFMA A-operand rows = weights, B-operand columns = pixels.  Both operands are staged in
// LDS as [row][64 bytes of k] (+16 B row padding); a lane's fragment is one 16-byte chunk
// (row = lane&15, chunk = lane>>4): 8 bf16 for v_mfma_f32_16x16x32_bf16, or 4 floats fed to
// four v_mfma_f32_16x16x4_f32 (exact fp32 parity mode; both operands use the same
// k-permutation so the sum over k is unchanged).
#include "conv_common.h"
#include "conv_ws.h"
#include "conv_stem.h"
#include "conv_img.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

namespace {

constexpr int kRowBytes = 80;  // 64 B of k + 16 B pad (spreads ds_read_b128 over the banks)

template <typename T, int BM, int BN, int WM, int WN, bool MASK = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;  // k elements per tile (64 bytes per row)
  constexpr int A_PASSES = (BM + 63) / 64;
  constexpr int B_PASSES = (BN + 63) / 64;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  static_assert(WM * WN == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * kRowBytes];
  unsigned char* Xs = smem;                       // [2][BM][kRowBytes]
  unsigned char* Ws = smem + 2 * BM * kRowBytes;  // [2][BN][kRowBytes]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bm0 = blockIdx.x * BM;
  const int bn0 = blockIdx.y * BN;
  const int cq = tid & 3;     // 16-byte chunk inside the 64-byte k row
  const int lrow = tid >> 2;  // 0..63

  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ wg = (const T*)a.w;

  // ---- per-thread im2col row state (one per pass) ------------------------------------
  int xbase[A_PASSES], hi0[A_PASSES], wi0[A_PASSES];
  bool mok[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int r = lrow + 64 * i;
    const int m = bm0 + r;
    mok[i] = (r < BM) && (m < a.M);
    int mm = mok[i] ? m : 0;
    const int wo = mm % a.Wo;
    mm /= a.Wo;
    const int ho = mm % a.Ho;
    const int b = mm / a.Ho;
    hi0[i] = ho * a.SH - a.PT;
    wi0[i] = wo * a.SW - a.PL;
    xbase[i] = ((b * a.H + hi0[i]) * a.W + wi0[i]) * a.x_cs + a.x_co;
  }
  // k state of this thread's chunk
  int kc, kkw, kkh;
  {
    const int kk = cq * EPC;
    kc = kk % a.Cin;
    const int tap = kk / a.Cin;
    kkw = tap % a.KW;
    kkh = tap / a.KW;
  }
  // weight rows
  const T* wrow[B_PASSES];
  bool nok[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int r = lrow + 64 * i;
    const int n = bn0 + r;
    nok[i] = (r < BN) && (n < a.Cout);
    wrow[i] = wg + (size_t)(nok[i] ? n : 0) * a.Kpad + cq * EPC;
  }

  const int nk = a.Kpad / BKE;
  uint4 areg[A_PASSES], breg[B_PASSES];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto load_tile = [&](int kt) {
    const bool kvalid = kkh < a.KH;
    const int koff = (kkh * a.W + kkw) * a.x_cs + kc;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int hi = hi0[i] + kkh, wi = wi0[i] + kkw;
      const bool ok = mok[i] && kvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
      areg[i] = ok ? *(const uint4*)(xg + (xbase[i] + koff)) : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      breg[i] = nok[i] ? *(const uint4*)(wrow[i] + (size_t)kt * BKE) : zero4;
    }
    // advance k state by one tile
    kc += BKE;
    while (kc >= a.Cin) {
      kc -= a.Cin;
      if (++kkw == a.KW) {
        kkw = 0;
        ++kkh;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int r = lrow + 64 * i;
      if (r < BM) *(uint4*)(Xs + (buf * BM + r) * kRowBytes + cq * 16) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
      const int r = lrow + 64 * i;
      if (r < BN) *(uint4*)(Ws + (buf * BN + r) * kRowBytes + cq * 16) = breg[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fchunk = lane >> 4;
  const int wbase = (wn * (BN / WN) + frow) * kRowBytes + fchunk * 16;
  const int xbase_l = (wm * (BM / WM) + frow) * kRowBytes + fchunk * 16;
  // n-tiles entirely beyond Cout are skipped (wave-uniform)
  int tn_live = 0;
#pragma unroll
  for (int i = 0; i < TN; ++i)
    if (bn0 + wn * (BN / WN) + i * 16 < a.Cout) tn_live = i + 1;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    uint4 wf[TN], xf[TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
      wf[i] = *(const uint4*)(Ws + buf * BN * kRowBytes + wbase + i * 16 * kRowBytes);
#pragma unroll
    for (int j = 0; j < TM; ++j)
      xf[j] = *(const uint4*)(Xs + buf * BM * kRowBytes + xbase_l + j * 16 * kRowBytes);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (i < tn_live) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, wf[i]), __builtin_bit_cast(bf16x8_t, xf[j]), acc[i][j], 0, 0, 0);
          } else {
            const f32x4_t wv = __builtin_bit_cast(f32x4_t, wf[i]);
            const f32x4_t xv = __builtin_bit_cast(f32x4_t, xf[j]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], xv[0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], xv[1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], xv[2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], xv[3], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: y = relu(acc * scale[n] + shift[n]) ----------------------------------
  const int mcol = lane & 15, nq = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n0 = bn0 + wn * (BN / WN) + i * 16 + nq;
    if (n0 >= a.Cout) continue;
    const float4 sc = a.scale ? *(const float4*)(a.scale + n0) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = a.scale ? *(const float4*)(a.shift + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = bm0 + wm * (BM / WM) + j * 16 + mcol;
      if (m >= a.M) continue;
      float v0 = acc[i][j][0] * sc.x + sh.x;
      float v1 = acc[i][j][1] * sc.y + sh.y;
      float v2 = acc[i][j][2] * sc.z + sh.z;
      float v3 = acc[i][j][3] * sc.w + sh.w;
      if (a.relu) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
        v2 = fmaxf(v2, 0.f);
        v3 = fmaxf(v3, 0.f);
      }
      const size_t off = (size_t)m * a.y_cs + a.y_co + n0;
      if constexpr (MASK) {    // fused activation gradient of the producer conv (ConvArgs::mask_y); T is the plan dtype
        const size_t yoff = (size_t)m * a.mask_cs + a.mask_co + n0;
        float yv[4];
        if (sizeof(T) == 4) {
          const float4 t = *(const float4*)((const float*)a.mask_y + yoff);
          yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w;
        } else {
          const uint2 t = *(const uint2*)((const bf16_t*)a.mask_y + yoff);
          yv[0] = __uint_as_float(t.x << 16); yv[1] = __uint_as_float(t.x & 0xFFFF0000u);
          yv[2] = __uint_as_float(t.y << 16); yv[3] = __uint_as_float(t.y & 0xFFFF0000u);
        }
        const float4 bs = *(const float4*)(a.mask_scale + n0);
        const float g0 = yv[0] > 0.f ? v0 : 0.f, g1 = yv[1] > 0.f ? v1 : 0.f, g2 = yv[2] > 0.f ? v2 : 0.f, g3 = yv[3] > 0.f ? v3 : 0.f;
        float* db = a.mask_dbeta + (size_t)(blockIdx.x & (kMaskCopies - 1)) * a.Cout + n0;
        atomicAdd(db + 0, g0);
        atomicAdd(db + 1, g1);
        atomicAdd(db + 2, g2);
        atomicAdd(db + 3, g3);
        if (sizeof(T) == 4) *(float4*)((float*)a.y + off) = make_float4(g0 * bs.x, g1 * bs.y, g2 * bs.z, g3 * bs.w);
        else *(uint2*)((bf16_t*)a.y + off) = make_uint2(pack_bf16x2(g0 * bs.x, g1 * bs.y), pack_bf16x2(g2 * bs.z, g3 * bs.w));
        continue;
      }
      if (sizeof(T) == 4 || a.out_f32) {
        float4* yp = (float4*)((float*)a.y + off);
        if (a.accum) {
          const float4 o = *yp;
          v0 += o.x; v1 += o.y; v2 += o.z; v3 += o.w;
        }
        *yp = make_float4(v0, v1, v2, v3);
      } else {
        uint2* yp = (uint2*)((bf16_t*)a.y + off);
        if (a.accum) {
          const uint2 o = *yp;
          v0 += __uint_as_float(o.x << 16); v1 += __uint_as_float(o.x & 0xFFFF0000u);
          v2 += __uint_as_float(o.y << 16); v3 += __uint_as_float(o.y & 0xFFFF0000u);
        }
        *yp = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
}

// ---- stem convolution: fp32 NHWC input with Cin <= 4 (images), direct form ------------
// weights fp32 [K = KH*KW*Cin][Cout] staged in LDS; one output pixel per thread, 32 output
// channels per pass.
template <typename T>
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  const float* __restrict__ wg = (const float*)a.w;
  for (int i = threadIdx.x; i < a.K * a.Cout; i += blockDim.x) wsm[i] = wg[i];
  __syncthreads();
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= a.M) return;
  int mm = pix;
  const int wo = mm % a.Wo;
  mm /= a.Wo;
  const int ho = mm % a.Ho;
  const int b = mm / a.Ho;
  const int hi0 = ho * a.SH - a.PT, wi0 = wo * a.SW - a.PL;
  const float* __restrict__ xg = (const float*)a.x;
  for (int co0 = 0; co0 < a.Cout; co0 += 32) {
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (int kh = 0; kh < a.KH; ++kh) {
      const int hi = hi0 + kh;
      if ((unsigned)hi >= (unsigned)a.H) continue;
      for (int kw = 0; kw < a.KW; ++kw) {
        const int wi = wi0 + kw;
        if ((unsigned)wi >= (unsigned)a.W) continue;
        const float* xp = xg + ((size_t)(b * a.H + hi) * a.W + wi) * a.x_cs + a.x_co;
        for (int ci = 0; ci < a.Cin; ++ci) {
          const float xv = xp[ci];
          const float* wr = wsm + ((kh * a.KW + kw) * a.Cin + ci) * a.Cout + co0;
#pragma unroll
          for (int j = 0; j < 32; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
        }
      }
    }
    const size_t off = (size_t)pix * a.y_cs + a.y_co + co0;
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[j + q] * a.scale[co0 + j + q] + a.shift[co0 + j + q];
        if (a.relu) v[q] = fmaxf(v[q], 0.f);
      }
      if (sizeof(T) == 4 || a.out_f32) {
        *(float4*)((float*)a.y + off + j) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        *(uint2*)((bf16_t*)a.y + off + j) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
  }
}

// Stem convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, the same
// arithmetic type as the kernel above): K = KH*KW*Cin <= 32 (3x3x3 = 27 -> 7 k-steps of 4), no padding.
// A wave owns 64 consecutive output pixels = 4 column tiles of the swapped product D[cout][pixel];
// the whole filter lives in registers as A fragments, the pixel operand is gathered straight from the
// fp32 image (lane = (pixel, k & 3): 4 consecutive k are (almost always) 4 consecutive floats, and the
// 16 pixels of a tile are SW*Cin floats apart, so a wave load covers a dense span of the image row).
template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_stem_mfma_kernel(ConvArgs a) {
  constexpr int KS = 8;                      // k-steps of 4 (K <= 32)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const float* __restrict__ wg = (const float*)a.w;      // [K][Cout]
  const float* __restrict__ xg = (const float*)a.x;
  // bf16 plans: the 32-deep product runs on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi of the fp32 operands split
  // into two bf16 halves (three 16x16x32 MFMAs of 16 cycles instead of eight fp32 16x16x4 MFMAs of 32: the fp32 form
  // kept the stem matrix-bound at 180 us per 320 images); the dropped lo*lo terms are 2^-16 relative, far below the
  // bf16 rounding of the store.  fp32 plans keep the exact fp32 MFMA.  SPLIT changes the lane -> k map: a lane holds
  // k = 8*fg .. 8*fg+7 (one bf16x8 operand) instead of k = 4*s + fg.
  constexpr bool SPLIT = sizeof(T) == 2;
  float wf[NT][KS];
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int k = SPLIT ? 8 * fg + s : 4 * s + fg;
    const bool kv = k < a.K;
    const int kk = kv ? k : 0;
    const int ci = kk % a.Cin, tap = kk / a.Cin;
    const int kw = tap % a.KW, kh = tap / a.KW;
    koff[s] = kv ? (kh * a.W + kw) * a.x_cs + ci : -1;
#pragma unroll
    for (int i = 0; i < NT; ++i) wf[i][s] = kv ? wg[kk * a.Cout + i * 16 + fr] : 0.f;
  }
  // a workgroup walks kStemChunks chunks of 256 pixels with the weight fragments it loaded once
  constexpr int kStemChunks = 4;
  for (int chunk = 0; chunk < kStemChunks; ++chunk) {
  const int pix0 = ((blockIdx.x * kStemChunks + chunk) * 4 + wave) * 64;
  if (pix0 >= a.M) break;
  float xv[4][KS];
  int mrow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int pix = pix0 + t * 16 + fr;
    const bool ok = pix < a.M;
    int mm = ok ? pix : 0;
    const int wo = mm % a.Wo;
    mm /= a.Wo;
    const int ho = mm % a.Ho;
    const int b = mm / a.Ho;
    const int base = ((b * a.H + ho * a.SH) * a.W + wo * a.SW) * a.x_cs + a.x_co;
    mrow[t] = ok ? pix : -1;
#pragma unroll
    for (int s = 0; s < KS; ++s) xv[t][s] = (ok & (koff[s] >= 0)) ? xg[base + koff[s]] : 0.f;
  }
  f32x4_t acc[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if constexpr (SPLIT) {
    auto split8 = [](const float (&v)[KS], bf16x8_t& hi, bf16x8_t& lo) {
      uint32_t h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
        l[e] = pack_bf16x2(v[2 * e] - __uint_as_float(h[e] << 16), v[2 * e + 1] - __uint_as_float(h[e] & 0xFFFF0000u));
      }
      hi = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
      lo = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
    };
    bf16x8_t wh[NT], wl[NT], xh[4], xl[4];
#pragma unroll
    for (int i = 0; i < NT; ++i) split8(wf[i], wh[i], wl[i]);
#pragma unroll
    for (int t = 0; t < 4; ++t) split8(xv[t], xh[t], xl[t]);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh[t], acc[i][t], 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl[t], acc[i][t], 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh[t], acc[i][t], 0, 0, 0);
      }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][s], xv[t][s], acc[i][t], 0, 0, 0);
  }
  // epilogue: lane holds channels i*16 + fg*4 .. +3 of pixel mrow[t]
  if constexpr (NT == 2 && sizeof(T) == 2) {
    // 32 bf16 channels: the two tiles are exchanged row-wise (see conv_store_tiles) and a lane stores 16 bytes
    if (!a.out_f32 && !a.x3 && (((a.y_cs | a.y_co) & 7) == 0)) {
      float4 sc[2], sh[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        sc[i] = *(const float4*)(a.scale + i * 16 + fg * 4);
        sh[i] = *(const float4*)(a.shift + i * 16 + fg * 4);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t pk[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v0 = fmaf(acc[i][t][0], sc[i].x, sh[i].x), v1 = fmaf(acc[i][t][1], sc[i].y, sh[i].y);
          float v2 = fmaf(acc[i][t][2], sc[i].z, sh[i].z), v3 = fmaf(acc[i][t][3], sc[i].w, sh[i].w);
          if (a.relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
          }
          pk[i][0] = pack_bf16x2(v0, v1);
          pk[i][1] = pack_bf16x2(v2, v3);
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        if (mrow[t] >= 0)
          *(uint4*)((bf16_t*)a.y + (size_t)mrow[t] * a.y_cs + a.y_co + (fg & 1) * 16 + (fg >> 1) * 8) =
              make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
      continue;
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n0 = i * 16 + fg * 4;
    const float4 sc = *(const float4*)(a.scale + n0), sh = *(const float4*)(a.shift + n0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v0 = fmaf(acc[i][t][0], sc.x, sh.x), v1 = fmaf(acc[i][t][1], sc.y, sh.y);
      float v2 = fmaf(acc[i][t][2], sc.z, sh.z), v3 = fmaf(acc[i][t][3], sc.w, sh.w);
      if (a.relu) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      }
      if (mrow[t] < 0) continue;
      const size_t off = (size_t)mrow[t] * a.y_cs + a.y_co + n0;
      if (sizeof(T) == 4 || a.out_f32) {
        *(float4*)((float*)a.y + off) = make_float4(v0, v1, v2, v3);
      } else if (a.x3) {                  // COMIC_OP_X3: [hi | lo | hi] regions
        const uint32_t h01 = pack_bf16x2(v0, v1), h23 = pack_bf16x2(v2, v3);
        const uint32_t l01 = pack_bf16x2(v0 - __uint_as_float(h01 << 16), v1 - __uint_as_float(h01 & 0xFFFF0000u));
        const uint32_t l23 = pack_bf16x2(v2 - __uint_as_float(h23 << 16), v3 - __uint_as_float(h23 & 0xFFFF0000u));
        bf16_t* yp = (bf16_t*)a.y + off;
        *(uint2*)yp = make_uint2(h01, h23);
        *(uint2*)(yp + a.x3) = make_uint2(l01, l23);
        *(uint2*)(yp + 2 * a.x3) = make_uint2(h01, h23);
      } else {
        *(uint2*)((bf16_t*)a.y + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
  }
}

// The same product for stems with a LARGE filter and padding (Inception-V1's Conv2d_1a_7x7: 7x7 / stride 2, SAME, 3 -> 64,
// common/nets/inception_v1.py:59-60; K = 147).  The direct kernel above spent 4.3 ms on 640 images (47 % of the
// Inception-V1 forward).  bf16 plans only.
//   * a workgroup owns kStemRows output rows of one image; the input rows they need are staged ONCE in LDS as fp32 with
//     zeroed halo columns / rows (SAME padding costs no per-element test);
//   * k is re-indexed as k' = 24 kh + j, j = kw * Cin + ci < KW * Cin <= 24 (weights of the pad positions are zero): the
//     eight k' of an MFMA operand lane then lie inside ONE filter row, i.e. they are eight consecutive floats of a staged
//     input row -- four ds_read_b64 instead of eight bounds-checked gathers;
//   * every 32-deep chunk runs as hi*hi + hi*lo + lo*hi of bf16 halves; the whole filter lives in registers
//     (KC x NT hi / lo fragment pairs).
constexpr int kStemRows = 2;          // output rows per workgroup
constexpr int kStemKRow = 24;         // k' per filter row
constexpr int kStemFetch = 12;        // float2 a thread moves per staged unit: (rows with weights) * W * Cin / 2 <= 12 * 256
template <int NT, int KC>
__global__ __launch_bounds__(256) void conv_stem_wide_kernel(ConvArgs a, int lrow, int blocks_per_image, int total_units) {
  extern __shared__ __attribute__((aligned(16))) float xs[];      // two buffers of [(kStemRows - 1) * SH + 8][lrow]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const float* __restrict__ wg = (const float*)a.w;      // [K][Cout]
  auto split8 = [](const float (&v)[8], bf16x8_t& hi, bf16x8_t& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
      l[e] = pack_bf16x2(v[2 * e] - __uint_as_float(h[e] << 16), v[2 * e + 1] - __uint_as_float(h[e] & 0xFFFF0000u));
    }
    hi = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
  };
  const int nrows = (kStemRows - 1) * a.SH + 8;           // staged rows (filter rows >= KH carry zero weights: never loaded)
  const int lrows = (kStemRows - 1) * a.SH + a.KH;        // ... of which these are loaded
  const int jw = a.KW * a.Cin;                            // floats of one filter row
  const int half = a.W * a.x_cs / 2;                      // float2 per image row (x_cs == Cin: dense NHWC image)
  const float* __restrict__ xg = (const float*)a.x + a.x_co;
  // a unit = kStemRows output rows of one image.  Staged row r = image row ho0*SH - PT + r behind PL*Cin zero floats; rows
  // outside the image are written as zeros, the halo columns are zeroed once and never written again.
  auto fetch = [&](int u, float2 (&v)[kStemFetch]) {
    const int b = u / blocks_per_image, ho0 = (u - b * blocks_per_image) * kStemRows;
#pragma unroll
    for (int q = 0; q < kStemFetch; ++q) {
      const int i = threadIdx.x + 256 * q;
      const int r = i / half, c2 = i - r * half;
      const int hi = ho0 * a.SH - a.PT + r;
      v[q] = make_float2(0.f, 0.f);
      if (r < lrows && (unsigned)hi < (unsigned)a.H) v[q] = *(const float2*)(xg + ((size_t)(b * a.H + hi) * a.W) * a.x_cs + 2 * c2);
    }
  };
  auto stash = [&](float* buf, const float2 (&v)[kStemFetch]) {
#pragma unroll
    for (int q = 0; q < kStemFetch; ++q) {
      const int i = threadIdx.x + 256 * q;
      const int r = i / half, c2 = i - r * half;
      if (r < lrows) *(float2*)(buf + r * lrow + a.PL * a.Cin + 2 * c2) = v[q];
    }
  };
  for (int i = threadIdx.x; i < 2 * nrows * lrow; i += 256) xs[i] = 0.f;
  // ---- the filter as fragments: lane holds k' = 32 c + 8 fg + s of chunk c ------------------------------------------------
  bf16x8_t wh[KC][NT], wl[KC][NT];
  int off[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int k0 = 32 * c + 8 * fg;
    const int kh = k0 / kStemKRow, j0 = k0 - kh * kStemKRow;
    off[c] = kh * lrow + j0;
    float wf[NT][8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bool kv = kh < a.KH && j0 + s < jw;
      const int k = kv ? kh * jw + j0 + s : 0;
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i][s] = kv ? wg[k * a.Cout + i * 16 + fr] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) split8(wf[i], wh[c][i], wl[c][i]);
  }
  // ---- this wave's pixel tiles: tile = wave + 4 t over the kStemRows * Wo pixels of a unit --------------------------------
  const int npix = kStemRows * a.Wo;
  int xbase[4], prow[4], pcol[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = (wave + 4 * t) * 16 + fr;
    const int r = p / a.Wo, wo = p - r * a.Wo;
    const bool ok = p < npix;
    xbase[t] = ok ? r * a.SH * lrow + wo * a.SW * a.Cin : 0;
    prow[t] = ok ? r : (1 << 20);
    pcol[t] = wo;
  }
  float2 pre[kStemFetch];
  __syncthreads();                                        // the zero fill is complete
  int unit = blockIdx.x;
  if (unit < total_units) {
    fetch(unit, pre);
    stash(xs, pre);
  }
  __syncthreads();
  for (int it = 0; unit < total_units; unit += gridDim.x, ++it) {
    const float* buf = xs + (it & 1) * nrows * lrow;
    const int next = unit + gridDim.x;
    if (next < total_units) fetch(next, pre);             // in flight under this unit's matrix work
    const int b = unit / blocks_per_image, ho0 = (unit - b * blocks_per_image) * kStemRows;
    f32x4_t acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      bf16x8_t xh[4], xl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float2* sp = (const float2*)(buf + xbase[t] + off[c]);      // even float offset: 8-byte aligned
        const float2 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
        const float xv[8] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, v3.x, v3.y};
        split8(xv, xh[t], xl[t]);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c][i], xh[t], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[c][i], xl[t], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[c][i], xh[t], acc[i][t], 0, 0, 0);
        }
    }
    // epilogue: lane holds channels i*16 + fg*4 .. +3 of its pixel of tile t
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n0 = i * 16 + fg * 4;
      const float4 sc = *(const float4*)(a.scale + n0), sh = *(const float4*)(a.shift + n0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v0 = fmaf(acc[i][t][0], sc.x, sh.x), v1 = fmaf(acc[i][t][1], sc.y, sh.y);
        float v2 = fmaf(acc[i][t][2], sc.z, sh.z), v3 = fmaf(acc[i][t][3], sc.w, sh.w);
        if (a.relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        if (ho0 + prow[t] >= a.Ho) continue;
        const size_t o = ((size_t)(b * a.Ho + ho0 + prow[t]) * a.Wo + pcol[t]) * a.y_cs + a.y_co + n0;
        if (a.out_f32)
          *(float4*)((float*)a.y + o) = make_float4(v0, v1, v2, v3);
        else
          *(uint2*)((bf16_t*)a.y + o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
    if (next < total_units) stash(xs + ((it + 1) & 1) * nrows * lrow, pre);
    __syncthreads();                                      // the other buffer is complete; this one is free
  }
}

// ---- pooling ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float* v);
template <>
__device__ __forceinline__ void load_vec<float>(const float* p, float* v) {
  const float4 t = *(const float4*)p;
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float* v) {
  const uint4 t = *(const uint4*)p;
  const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(u[i] << 16);
    v[2 * i + 1] = __uint_as_float(u[i] & 0xFFFF0000u);
  }
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float* v);
template <>
__device__ __forceinline__ void store_vec<float>(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* v) {
  *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                          pack_bf16x2(v[6], v[7]));
}

// MODE 0: max-pool (VALID or padded with -inf)   MODE 1: avg-pool dividing by valid taps
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pool_kernel(ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cvecs = a.Cin / EPC;
  const long total = (long)a.M * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  int mm = (int)(idx / cvecs);
  const int pix = mm;
  const int wo = mm % a.Wo;
  mm /= a.Wo;
  const int ho = mm % a.Ho;
  const int b = mm / a.Ho;
  const T* __restrict__ xg = (const T*)a.x;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = MODE == 0 ? -INFINITY : 0.f;
  int cnt = 0;
  for (int kh = 0; kh < a.KH; ++kh) {
    const int hi = ho * a.SH - a.PT + kh;
    if ((unsigned)hi >= (unsigned)a.H) continue;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int wi = wo * a.SW - a.PL + kw;
      if ((unsigned)wi >= (unsigned)a.W) continue;
      float v[EPC];
      load_vec<T>(xg + ((size_t)(b * a.H + hi) * a.W + wi) * a.x_cs + a.x_co + cv * EPC, v);
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] = MODE == 0 ? fmaxf(acc[j], v[j]) : acc[j] + v[j];
      ++cnt;
    }
  }
  if (MODE == 1) {
    const float c = (float)cnt;
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] = acc[j] / c;
  }
  store_vec<T>((T*)a.y + (size_t)pix * a.y_cs + a.y_co + cv * EPC, acc);
}

// The same pools over COMIC_OP_X3 buffers: a value is hi + lo of two channel regions.  Max: the pair with the larger hi, then
// the larger lo (|lo| <= ulp(hi) / 2, so this is the order of hi + lo); the winning pair is copied.  Average: fp32 sum of
// hi + lo over the valid taps, divided, split again.  Output: the three regions [hi | lo | hi].
template <int MODE>
__global__ __launch_bounds__(256) void pool_x3_kernel(ConvArgs a) {
  constexpr int EPC = 8;
  const int cvecs = a.Cin / EPC;
  const long total = (long)a.M * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  int mm = (int)(idx / cvecs);
  const int pix = mm;
  const int wo = mm % a.Wo;
  mm /= a.Wo;
  const int ho = mm % a.Ho;
  const int b = mm / a.Ho;
  const bf16_t* __restrict__ xg = (const bf16_t*)a.x;
  float hi[EPC], lo[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    hi[j] = MODE == 0 ? -INFINITY : 0.f;
    lo[j] = 0.f;
  }
  int cnt = 0;
  for (int kh = 0; kh < a.KH; ++kh) {
    const int hy = ho * a.SH - a.PT + kh;
    if ((unsigned)hy >= (unsigned)a.H) continue;
    for (int kw = 0; kw < a.KW; ++kw) {
      const int wx = wo * a.SW - a.PL + kw;
      if ((unsigned)wx >= (unsigned)a.W) continue;
      float vh[EPC], vl[EPC];
      const bf16_t* src = xg + ((size_t)(b * a.H + hy) * a.W + wx) * a.x_cs + a.x_co + cv * EPC;
      load_vec<bf16_t>(src, vh);
      load_vec<bf16_t>(src + a.x3_src, vl);
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        if (MODE == 0) {
          const bool take = vh[j] > hi[j] || (vh[j] == hi[j] && vl[j] > lo[j]);
          hi[j] = take ? vh[j] : hi[j];
          lo[j] = take ? vl[j] : lo[j];
        } else {
          hi[j] += vh[j] + vl[j];
        }
      }
      ++cnt;
    }
  }
  if (MODE == 1) {
    const float c = (float)cnt;
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const float v = hi[j] / c;
      hi[j] = bf16_to_f32(f32_to_bf16(v));
      lo[j] = v - hi[j];
    }
  }
  bf16_t* dst = (bf16_t*)a.y + (size_t)pix * a.y_cs + a.y_co + cv * EPC;
  store_vec<bf16_t>(dst, hi);
  store_vec<bf16_t>(dst + a.x3, lo);
  store_vec<bf16_t>(dst + 2 * a.x3, hi);
}

// 3x3 stride-1 SAME max-pool (the pool branches of the Inception-V1 / V3 blocks) as one thread per (image row, channel
// vector): the thread walks the row and keeps the maxima of the last three columns over the (up to three) valid source
// rows, so every source element is loaded once per output ROW that needs it (3 loads per output instead of 9; the
// generic kernel above took 22 % of the Inception-V1 forward).  max is exact in any order: same results.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3_rows_kernel(ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cvecs = a.Cin / EPC;
  const long total = (long)a.B * a.H * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  const int q = (int)(idx / cvecs);
  const int ho = q % a.H, b = q / a.H;
  const T* __restrict__ xg = (const T*)a.x;
  const int r0 = max(ho - 1, 0), r1 = min(ho + 1, a.H - 1);
  auto colmax = [&](int w, float (&m)[EPC]) {
#pragma unroll
    for (int j = 0; j < EPC; ++j) m[j] = -INFINITY;
    for (int r = r0; r <= r1; ++r) {
      float v[EPC];
      load_vec<T>(xg + ((size_t)(b * a.H + r) * a.W + w) * a.x_cs + a.x_co + cv * EPC, v);
#pragma unroll
      for (int j = 0; j < EPC; ++j) m[j] = fmaxf(m[j], v[j]);
    }
  };
  float c0[EPC], c1[EPC], c2[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) c0[j] = -INFINITY;
  colmax(0, c1);
  for (int wo = 0; wo < a.W; ++wo) {
    if (wo + 1 < a.W) {
      colmax(wo + 1, c2);
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) c2[j] = -INFINITY;
    }
    float o[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) o[j] = fmaxf(fmaxf(c0[j], c1[j]), c2[j]);
    store_vec<T>((T*)a.y + ((size_t)(b * a.H + ho) * a.W + wo) * a.y_cs + a.y_co + cv * EPC, o);
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      c0[j] = c1[j];
      c1[j] = c2[j];
    }
  }
}

// bf16 store of four channels of a kind-7 output; a.x3 (COMIC_OP_X3): as the three regions [hi | lo | hi]
__device__ __forceinline__ void pbr_store_bf16(const ConvArgs& a, size_t off, float v0, float v1, float v2, float v3) {
  bf16_t* yp = (bf16_t*)a.y + off;
  const uint32_t h01 = pack_bf16x2(v0, v1), h23 = pack_bf16x2(v2, v3);
  *(uint2*)yp = make_uint2(h01, h23);
  if (a.x3) {
    const uint32_t l01 = pack_bf16x2(v0 - __uint_as_float(h01 << 16), v1 - __uint_as_float(h01 & 0xFFFF0000u));
    const uint32_t l23 = pack_bf16x2(v2 - __uint_as_float(h23 << 16), v3 - __uint_as_float(h23 & 0xFFFF0000u));
    *(uint2*)(yp + a.x3) = make_uint2(l01, l23);
    *(uint2*)(yp + 2 * a.x3) = make_uint2(h01, h23);
  }
}

// kind 7: 3x3 s1 SAME average (divisor = taps inside the image) of an fp32 map, then the folded
// BatchNorm + ReLU of the projection that produced it.  One thread per (pixel, 4 channels).
__device__ __forceinline__ void pool_bn_relu_item(const ConvArgs& a, const long idx, const int cvecs) {
  const int cv = (int)(idx % cvecs);
  int mm = (int)(idx / cvecs);
  const int pix = mm;
  const int wo = mm % a.Wo;
  mm /= a.Wo;
  const int ho = mm % a.Ho;
  const int b = mm / a.Ho;
  const float* __restrict__ xg = (const float*)a.x;
  // all nine taps are loaded unconditionally from clamped coordinates (nine independent 16-byte loads in flight);
  // taps outside the image are then skipped in the sum, in the same (kh, kw) order as a branchy loop
  float4 v[9];
  bool ok[9];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho - 1 + kh;
    const int hc = min(max(hi, 0), a.H - 1);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo - 1 + kw;
      const int wc = min(max(wi, 0), a.W - 1);
      ok[kh * 3 + kw] = ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
      v[kh * 3 + kw] = *(const float4*)(xg + ((size_t)(b * a.H + hc) * a.W + wc) * a.x_cs + a.x_co + cv * 4);
    }
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (ok[t]) {
      acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w;
      ++cnt;
    }
  }
  const float c = (float)cnt;
  const float4 sc = *(const float4*)(a.scale + cv * 4), sh = *(const float4*)(a.shift + cv * 4);
  float v0 = acc.x / c * sc.x + sh.x, v1 = acc.y / c * sc.y + sh.y, v2 = acc.z / c * sc.z + sh.z, v3 = acc.w / c * sc.w + sh.w;
  if (a.relu) {
    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
  }
  const size_t off = (size_t)pix * a.y_cs + a.y_co + cv * 4;
  if (a.out_f32)
    *(float4*)((float*)a.y + off) = make_float4(v0, v1, v2, v3);
  else
    pbr_store_bf16(a, off, v0, v1, v2, v3);
}

__global__ __launch_bounds__(256) void pool_bn_relu_kernel(ConvArgs a) {
  const int cvecs = a.Cin / 4;
  const long total = (long)a.M * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) pool_bn_relu_item(a, idx, cvecs);
}

// The same op as one thread per (image row, 4 channels): the thread walks the row and keeps the sums of the last three
// columns over the (up to three) valid source rows, so every source element is loaded once per output ROW that needs it
// (3 loads per output instead of 9).  Summation order: rows inside a column first, then the columns left to right.
__global__ __launch_bounds__(256) void pool_bn_relu_rows_kernel(ConvArgs a) {
  const int cvecs = a.Cin / 4;
  const long total = (long)a.B * a.H * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  const int q = (int)(idx / cvecs);
  const int ho = q % a.H, b = q / a.H;
  const float* __restrict__ xg = (const float*)a.x;
  const int r0 = max(ho - 1, 0), r1 = min(ho + 1, a.H - 1);
  const float nrows = (float)(r1 - r0 + 1);
  const float4 sc = *(const float4*)(a.scale + cv * 4), sh = *(const float4*)(a.shift + cv * 4);
  const float lo = a.relu ? 0.f : -INFINITY;
  // The row is walked in blocks of kAhead columns whose source loads are all issued before the first of them is used: with
  // the loads of column wo + 1 requested in the iteration that consumes them, every output pixel waited a full memory
  // round trip (84 us for 1280 x 25 x 25 x 64: twice the HBM floor).  Same sums in the same order.
  constexpr int kAhead = 4;
  auto col_loads = [&](int w, float4 (&v)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = r0 + k;
      v[k] = (w < a.W && r <= r1) ? *(const float4*)(xg + ((size_t)(b * a.H + r) * a.W + w) * a.x_cs + a.x_co + cv * 4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto col_sum = [&](const float4 (&v)[3]) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (r0 + k <= r1) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    return s;
  };
  float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1, c2;
  {
    float4 v[3];
    col_loads(0, v);
    c1 = col_sum(v);
  }
  for (int w0 = 0; w0 < a.W; w0 += kAhead) {
    float4 nx[kAhead][3];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) col_loads(w0 + u + 1, nx[u]);      // columns w0 + 1 .. w0 + kAhead (zeros past the row)
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int wo = w0 + u;
      if (wo >= a.W) break;
      const bool right = wo + 1 < a.W;
      c2 = right ? col_sum(nx[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float cnt = nrows * (float)(1 + (wo > 0) + right);
      float v0 = ((c0.x + c1.x) + c2.x) / cnt * sc.x + sh.x, v1 = ((c0.y + c1.y) + c2.y) / cnt * sc.y + sh.y;
      float v2 = ((c0.z + c1.z) + c2.z) / cnt * sc.z + sh.z, v3 = ((c0.w + c1.w) + c2.w) / cnt * sc.w + sh.w;
      v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
      const size_t off = ((size_t)(b * a.H + ho) * a.W + wo) * a.y_cs + a.y_co + cv * 4;
      if (a.out_f32)
        *(float4*)((float*)a.y + off) = make_float4(v0, v1, v2, v3);
      else
        pbr_store_bf16(a, off, v0, v1, v2, v3);
      c0 = c1;
      c1 = c2;
    }
  }
}

// The same work as a member of a grouped conv launch: workgroup `local` of this member handles
// kPoolItemsPerThread x blockDim.x consecutive (pixel, 4-channel) items.
constexpr int kPoolItemsPerThread = 4;
__device__ __forceinline__ void pool_bn_relu_member(const ConvArgs& a, const int local) {
  const int cvecs = a.Cin / 4;
  const long total = (long)a.M * cvecs;
  const long base = (long)local * blockDim.x * kPoolItemsPerThread + threadIdx.x;
#pragma unroll
  for (int r = 0; r < kPoolItemsPerThread; ++r) {
    const long idx = base + (long)r * blockDim.x;
    if (idx < total) pool_bn_relu_item(a, idx, cvecs);
  }
}

// global KHxKW VALID average -> fp32 [B, Ho*Wo, C]; one thread per (pixel, channel vec)
template <typename T>
__global__ __launch_bounds__(256) void global_avgpool_kernel(ConvArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cvecs = a.Cin / EPC;
  const long total = (long)a.M * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  int mm = (int)(idx / cvecs);
  const int pix = mm;
  const int wo = mm % a.Wo;
  mm /= a.Wo;
  const int ho = mm % a.Ho;
  const int b = mm / a.Ho;
  const T* __restrict__ xg = (const T*)a.x;
  float acc[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
  for (int kh = 0; kh < a.KH; ++kh)
    for (int kw = 0; kw < a.KW; ++kw) {
      float v[EPC];
      load_vec<T>(xg + ((size_t)(b * a.H + ho * a.SH + kh) * a.W + wo * a.SW + kw) * a.x_cs + a.x_co + cv * EPC, v);
#pragma unroll
      for (int j = 0; j < EPC; ++j) acc[j] += v[j];
    }
  const float inv = (float)(a.KH * a.KW);
  float* yp = (float*)a.y + (size_t)pix * a.y_cs + a.y_co + cv * EPC;
#pragma unroll
  for (int j = 0; j < EPC; ++j) yp[j] = acc[j] / inv;
}

// ---- weight packing / BN folding -----------------------------------------------------------
template <typename T>
__global__ void pack_conv_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int K, int Kpad,
                                         int cout) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)cout * Kpad) return;
  const int k = (int)(idx % Kpad), n = (int)(idx / Kpad);
  const float v = k < K ? w[(size_t)k * cout + n] : 0.f;
  if (sizeof(T) == 4)
    ((float*)out)[idx] = v;
  else
    ((bf16_t*)out)[idx] = f32_to_bf16(v);
}

__global__ void fold_bn_kernel(const float* beta, const float* mean, const float* var, float eps, float* scale,
                               float* shift, int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float s = 1.0f / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = beta[i] - mean[i] * s;
}


// ---------------------------------------------------------------------------------------------
// bf16 throughput path: LDS-DMA pipelined implicit GEMM.
//
// Same math and orientation as conv_igemm_kernel, but the operand tiles go global -> LDS with
// `global_load_lds_dwordx4` (no VGPR staging) into a ring of NSTAGE stages of [rows][128 B of
// k] (BK = 64 bf16), so two whole k-tiles stay in flight behind the one being multiplied.
// One raw s_barrier per k-tile; DMA completion is tracked with counted `s_waitcnt vmcnt(N)`;
// fragment reads are inline-asm ds_read_b128 (hipcc would otherwise drain vmcnt(0) in front of
// every LDS read that may alias an in-flight DMA -- cdna_hip_programming.md §5).
//
// LDS image: 128-byte rows, 16-byte slot = chunk ^ ((row >> 1) & 7).  The DMA writes LDS
// linearly (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE chunk; the
// ds_read_b128 lane groups of the 16x16x32 operand fetch then hit 16 distinct slots.
// Padding / out-of-range rows read from a 16-byte zero page.
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};


// ALIGNED (Cin % 64 == 0): a whole k-tile lies inside one filter tap, so the tap walk (kh, kw, c0)
// is wave-uniform scalar state and the per-lane part of a source address is a constant.
// NSTAGE_ = kLoaderWaves + n: n ring stages and WAVE SPECIALISATION -- the WM x WN waves only read fragments and issue
// MFMAs, four more waves only issue the LDS-DMA pieces.  An LDS-DMA instruction holds its wave for ~70 cycles (stamps:
// 700 of the 1950 cycles of a 128x192 k-tile went into issuing its 10 pieces); on a wave of its own that time runs
// beside the matrix work instead of in front of it.  Both roles execute one s_barrier per k-tile.
constexpr int kLoaderWaves = 20;
constexpr int ring_stages(int nstage) { return nstage % 10; }
constexpr int dma_threads(int wm, int wn, int nstage) { return (wm * wn + (nstage >= kLoaderWaves ? 4 : 0)) * 64; }

template <int BM, int BN, int WM, int WN, int NSTAGE_, bool ALIGNED, bool MASK = false>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvArgs& a, const int block_m, const int block_n) {
  constexpr int BKE = 64;                       // bf16 elements per k-tile = 128 bytes per row
  constexpr int ROWS = BM + BN;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int NSTAGE = ring_stages(NSTAGE_);
  constexpr bool SPEC = NSTAGE_ >= kLoaderWaves;
  constexpr int NWC = WM * WN;                  // waves that own output tiles (4 or 8)
  constexpr int NW = SPEC ? 4 : NWC;            // waves that issue the LDS-DMA pieces
  constexpr int IPW_A = BM / (8 * NW);          // 8-row DMA instructions per wave per stage (A)
  constexpr int IPW_B = BN / (8 * NW);
  constexpr int LPT = IPW_A + IPW_B;            // DMA instructions per wave per stage
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  static_assert((NWC == 4 || NWC == 8) && BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && BM % (16 * WM) == 0 &&
                    BN % (16 * WN) == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS DMA bases become SGPR math
  const bool loader = SPEC && wave_id >= NWC;
  const int wave = loader ? wave_id - NWC : wave_id;           // index inside its role
  const int wm = wave / WN, wn = wave % WN;
  const int bm0 = block_m * BM, bn0 = block_n * BN;
  const bf16_t* __restrict__ xg = (const bf16_t*)a.x;
  const bf16_t* __restrict__ wg = (const bf16_t*)a.w;
  const bf16_t* zp = (const bf16_t*)a.zero;
  const int slot = lane & 7, rsub = lane >> 3;

  // ---- im2col state of the rows this lane feeds ------------------------------------------
  int xbase[IPW_A], hi0[IPW_A], wi0[IPW_A];
  bool mok[IPW_A];
#pragma unroll
  for (int i = 0; i < IPW_A; ++i) {
    const int r = (wave * IPW_A + i) * 8 + rsub;
    const int m = bm0 + r;
    mok[i] = m < a.M;
    int mm = mok[i] ? m : 0;
    const int wo = mm % a.Wo;
    mm /= a.Wo;
    const int ho = mm % a.Ho;
    const int b = mm / a.Ho;
    hi0[i] = ho * a.SH - a.PT;
    wi0[i] = wo * a.SW - a.PL;
    xbase[i] = ((b * a.H + hi0[i]) * a.W + wi0[i]) * a.x_cs + a.x_co;
  }
  // the source chunk of a lane alternates between two values with the parity of the 8-row group
  // (swizzle term (row>>1)&7 = 4*(group&1) + (rsub>>1)); keep one k-state per parity
  int kc[2], kkw[2], kkh[2];
  int chunk8[2];          // ALIGNED: element offset of this lane's chunk inside the k-tile, per parity
  int s_kh = 0, s_kw = 0, s_c0 = 0;   // ALIGNED: wave-uniform tap walk
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int chunk = slot ^ ((4 * par + (rsub >> 1)) & 7);
    const int kk = chunk * 8;
    chunk8[par] = kk;
    kc[par] = kk % a.Cin;
    const int tap = kk / a.Cin;
    kkw[par] = tap % a.KW;
    kkh[par] = tap / a.KW;
  }
  const bf16_t* wsrc[IPW_B];
  bool nok[IPW_B];
#pragma unroll
  for (int i = 0; i < IPW_B; ++i) {
    const int gb = wave * IPW_B + i;
    const int n = bn0 + gb * 8 + rsub;
    nok[i] = n < a.Cout;
    const int chunk = slot ^ ((4 * (gb & 1) + (rsub >> 1)) & 7);
    wsrc[i] = wg + (size_t)(nok[i] ? n : 0) * a.Kpad + chunk * 8;
  }
  const int nk = a.Kpad / BKE;

  auto issue = [&](int kt, int stage) {
    unsigned char* sbase = smem + stage * STAGE_BYTES;
    if constexpr (ALIGNED) {
      const bool kvalid = s_kh < a.KH;
      const int koff = (s_kh * a.W + s_kw) * a.x_cs + s_c0;   // scalar
#pragma unroll
      for (int i = 0; i < IPW_A; ++i) {
        const int ga = wave * IPW_A + i;
        const int hi = hi0[i] + s_kh, wi = wi0[i] + s_kw;
        const bool ok = mok[i] & kvalid & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
        const bf16_t* src = xg + (xbase[i] + koff + chunk8[ga & 1]);
        src = ok ? src : zp;
        dma16(src, sbase + ga * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < IPW_A; ++i) {
        const int ga = wave * IPW_A + i;
        const int par = ga & 1;
        const int hi = hi0[i] + kkh[par], wi = wi0[i] + kkw[par];
        const bool ok = mok[i] & (kkh[par] < a.KH) & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
        const bf16_t* src = xg + (xbase[i] + (kkh[par] * a.W + kkw[par]) * a.x_cs + kc[par]);
        src = ok ? src : zp;
        dma16(src, sbase + ga * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < IPW_B; ++i) {
      const int gb = wave * IPW_B + i;
      const bf16_t* src = wsrc[i] + (size_t)kt * BKE;
      src = nok[i] ? src : zp;
      dma16(src, sbase + BM * 128 + gb * 1024);
    }
    if constexpr (ALIGNED) {
      s_c0 += BKE;
      if (s_c0 >= a.Cin) {
        s_c0 = 0;
        if (++s_kw == a.KW) {
          s_kw = 0;
          ++s_kh;
        }
      }
    } else {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        kc[par] += BKE;
        while (kc[par] >= a.Cin) {
          kc[par] -= a.Cin;
          if (++kkw[par] == a.KW) {
            kkw[par] = 0;
            ++kkh[par];
          }
        }
      }
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane (r, g) reads row r of a 16-row tile, k-chunk ks*4+g, at slot chunk^((r>>1)&7)
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t sw = (fr >> 1) & 7;
  const uint32_t x_off0 = (wm * (BM / WM) + fr) * 128 + (((0 + fg) ^ sw) & 7) * 16;
  const uint32_t x_off1 = (wm * (BM / WM) + fr) * 128 + (((4 + fg) ^ sw) & 7) * 16;
  const uint32_t w_off0 = (BM + wn * (BN / WN) + fr) * 128 + (((0 + fg) ^ sw) & 7) * 16;
  const uint32_t w_off1 = (BM + wn * (BN / WN) + fr) * 128 + (((4 + fg) ^ sw) & 7) * 16;

  // NSTAGE-1 tiles are in flight before the loop; at the top of iteration kt the tiles
  // kt .. min(kt+NSTAGE-2, nk-1) have been issued and tile kt must have landed.
  constexpr int AHEAD = NSTAGE - 1;
  STAMP(0);
  if constexpr (SPEC) {
    if (loader) {
      // loader waves: keep AHEAD tiles in flight; tile kt must have landed before barrier kt releases its readers
#pragma unroll
      for (int p = 0; p < AHEAD; ++p)
        if (p < nk) issue(p, p);
      for (int kt = 0; kt < nk; ++kt) {
        const int pending = min(AHEAD - 1, nk - 1 - kt);
        if (pending >= 2)
          wait_vmcnt<2 * LPT>();
        else if (pending == 1)
          wait_vmcnt<LPT>();
        else
          wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % NSTAGE);
      }
      return;
    }
  } else {
#pragma unroll
    for (int p = 0; p < AHEAD; ++p)
      if (p < nk) issue(p, p);
  }
  STAMP(1);
#ifdef COMIC_STAMPS
  unsigned long long t_issue = 0, t_wait = 0;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt % NSTAGE;
#ifdef COMIC_STAMPS
    const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (!SPEC) {
      const int pending = min(AHEAD - 1, nk - 1 - kt);   // tiles allowed to stay in flight
      if (pending >= 2)
        wait_vmcnt<2 * LPT>();
      else if (pending == 1)
        wait_vmcnt<LPT>();
      else
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();   // every wave's share of tile kt landed; the stage read in kt-1 is free
#ifdef COMIC_STAMPS
    const unsigned long long ti0 = __builtin_amdgcn_s_memtime();
    t_wait += ti0 - tw0;
    if (kt == 0 && tid == 0) g_stamps[(blockIdx.x & 16383) * 8 + 7] = ti0 - tw0;    // the wait for the FIRST k-tile
#endif
    if constexpr (!SPEC) {
      if (kt + AHEAD < nk) issue(kt + AHEAD, (kt + AHEAD) % NSTAGE);
    }
#ifdef COMIC_STAMPS
    t_issue += __builtin_amdgcn_s_memtime() - ti0;
#endif
    const uint32_t sb = lds0 + stage * STAGE_BYTES;
    u32x4_t xf0[TM], xf1[TM], wf0[TN], wf1[TN];
    // both 32-deep halves of the k-tile are requested at once; the first half's MFMAs start as soon as ITS fragments
    // have landed (the LDS returns in order: lgkmcnt <= TM+TN leaves exactly the second half outstanding), so the
    // second half's read latency runs under them.  Per accumulator the order stays half 0, then half 1.
    static_assert(TM + TN <= 15, "lgkmcnt holds 4 bits");
    static_for<0, TN>([&](auto i) { wf0[i] = lds_read128<decltype(i)::value * 2048>(sb + w_off0); });
    static_for<0, TM>([&](auto j) { xf0[j] = lds_read128<decltype(j)::value * 2048>(sb + x_off0); });
    static_for<0, TN>([&](auto i) { wf1[i] = lds_read128<decltype(i)::value * 2048>(sb + w_off1); });
    static_for<0, TM>([&](auto j) { xf1[j] = lds_read128<decltype(j)::value * 2048>(sb + x_off1); });
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // n-tiles beyond Cout multiply zero-filled weight rows (no branch: keeps the accumulators in place)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf0[i]),
                                                            __builtin_bit_cast(bf16x8_t, xf0[j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf1[i]),
                                                            __builtin_bit_cast(bf16x8_t, xf1[j]), acc[i][j], 0, 0, 0);
  }

  STAMP(2);
#ifdef COMIC_STAMPS
  if (tid == 0) { g_stamps[(blockIdx.x & 16383) * 8 + 4] = t_issue; g_stamps[(blockIdx.x & 16383) * 8 + 5] = t_wait; g_stamps[(blockIdx.x & 16383) * 8 + 6] = nk; }
#endif
  // ---- epilogue: y = relu(acc * scale[n] + shift[n]) ----------------------------------
  int mrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = bm0 + wm * (BM / WM) + j * 16 + (lane & 15);
    mrow[j] = m < a.M ? m : -1;
  }
  conv_store_tiles<TN, TM, MASK>(a, acc, bn0 + wn * (BN / WN), (lane >> 4) * 4, mrow);
  STAMP(3);
}



// ---------------------------------------------------------------------------------------------
// "Walk" form of the loader-wave kernel for launches whose members read the SAME im2col matrix (the 1x1 convs at the head
// of an Inception block; ConvArgs::remap == 3): ONE workgroup per pixel tile walks over the out-channel tiles of ALL
// members.  Why (phase stamps of the 12x12 768 -> 704 groups at 1280 images, one tile per workgroup): the wait for the first
// k-tile is 14 % of a workgroup's life, 43 % of the k loop is spent at the barrier waiting for rows that come from HBM again
// although three sibling workgroups fetch the same rows (L2 hit rate of the pixel operand 0.54), the epilogue another 14 %.
// Here the k-tile stream of the loader waves runs on across the out-channel tiles (the ring never drains: the first
// k-tiles of the next tile land under the epilogue of this one), and after the first tile the pixel rows come from the
// L2 this workgroup has just filled.  Same operands in the same order per accumulator as conv_igemm_dma_body: identical bits.
template <int BM, int BN, int WM, int WN, int NSTAGE_, bool ALIGNED>
__device__ __forceinline__ void conv_igemm_dma_walk_body(const ConvArgs* __restrict__ args, const int n_members, const int block_m,
                                                         const int r_begin, const int r_end) {   // out-channel tiles [r_begin, r_end) of the walk
  constexpr int BKE = 64;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int NSTAGE = ring_stages(NSTAGE_);
  static_assert(NSTAGE_ >= kLoaderWaves, "the walk form needs the loader waves");
  constexpr int NWC = WM * WN;
  constexpr int NW = 4;
  constexpr int IPW_A = BM / (8 * NW);
  constexpr int IPW_B = BN / (8 * NW);
  constexpr int LPT = IPW_A + IPW_B;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int AHEAD = NSTAGE - 1;
  static_assert(NWC == 4 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const ConvArgs& a = args[0];                   // everything about the pixel operand is common to the members
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave_id >= NWC;
  const int wave = loader ? wave_id - NWC : wave_id;
  const int wm = wave / WN, wn = wave % WN;
  const int bm0 = block_m * BM;
  const int nk = a.Kpad / BKE;
  const int total = (r_end - r_begin) * nk;      // k-tiles of the whole walk
  const int slot = lane & 7, rsub = lane >> 3;

  if (loader) {
    const bf16_t* __restrict__ xg = (const bf16_t*)a.x;
    const bf16_t* zp = (const bf16_t*)a.zero;
    int xbase[IPW_A], hi0[IPW_A], wi0[IPW_A];
    bool mok[IPW_A];
#pragma unroll
    for (int i = 0; i < IPW_A; ++i) {
      const int r = (wave * IPW_A + i) * 8 + rsub;
      const int m = bm0 + r;
      mok[i] = m < a.M;
      int mm = mok[i] ? m : 0;
      const int wo = mm % a.Wo;
      mm /= a.Wo;
      const int ho = mm % a.Ho;
      const int b = mm / a.Ho;
      hi0[i] = ho * a.SH - a.PT;
      wi0[i] = wo * a.SW - a.PL;
      xbase[i] = ((b * a.H + hi0[i]) * a.W + wi0[i]) * a.x_cs + a.x_co;
    }
    int kc[2], kkw[2], kkh[2], chunk8[2];
    int s_kh = 0, s_kw = 0, s_c0 = 0;
    auto reset_k = [&]() {
      s_kh = s_kw = s_c0 = 0;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int chunk = slot ^ ((4 * par + (rsub >> 1)) & 7);
        const int kk = chunk * 8;
        chunk8[par] = kk;
        kc[par] = kk % a.Cin;
        const int tap = kk / a.Cin;
        kkw[par] = tap % a.KW;
        kkh[par] = tap / a.KW;
      }
    };
    const bf16_t* wsrc[IPW_B];
    bool nok[IPW_B];
    auto set_tile = [&](int r) {                 // weight rows of out-channel tile r (member p, its tile r - blk0)
      int p = 0;
      for (int i = 1; i < n_members; ++i)
        if (r >= args[i].blk0) p = i;
      const bf16_t* wg = (const bf16_t*)args[p].w;
      const int bn0 = (r - args[p].blk0) * BN, cout = args[p].Cout;
#pragma unroll
      for (int i = 0; i < IPW_B; ++i) {
        const int gb = wave * IPW_B + i;
        const int nn = bn0 + gb * 8 + rsub;
        nok[i] = nn < cout;
        const int chunk = slot ^ ((4 * (gb & 1) + (rsub >> 1)) & 7);
        wsrc[i] = wg + (size_t)(nok[i] ? nn : 0) * a.Kpad + chunk * 8;
      }
    };
    int gi = 0, gi_kt = 0, gi_r = r_begin;       // issue pointer: stream index, its k-tile and out-channel tile
    auto issue = [&]() {
      if (gi_kt == 0) {
        set_tile(gi_r);
        reset_k();
      }
      unsigned char* sbase = smem + (gi % NSTAGE) * STAGE_BYTES;
      if constexpr (ALIGNED) {
        const bool kvalid = s_kh < a.KH;
        const int koff = (s_kh * a.W + s_kw) * a.x_cs + s_c0;
#pragma unroll
        for (int i = 0; i < IPW_A; ++i) {
          const int ga = wave * IPW_A + i;
          const int hi = hi0[i] + s_kh, wi = wi0[i] + s_kw;
          const bool ok = mok[i] & kvalid & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
          const bf16_t* src = xg + (xbase[i] + koff + chunk8[ga & 1]);
          src = ok ? src : zp;
          dma16(src, sbase + ga * 1024);
        }
      } else {
#pragma unroll
        for (int i = 0; i < IPW_A; ++i) {
          const int ga = wave * IPW_A + i;
          const int par = ga & 1;
          const int hi = hi0[i] + kkh[par], wi = wi0[i] + kkw[par];
          const bool ok = mok[i] & (kkh[par] < a.KH) & ((unsigned)hi < (unsigned)a.H) & ((unsigned)wi < (unsigned)a.W);
          const bf16_t* src = xg + (xbase[i] + (kkh[par] * a.W + kkw[par]) * a.x_cs + kc[par]);
          src = ok ? src : zp;
          dma16(src, sbase + ga * 1024);
        }
      }
#pragma unroll
      for (int i = 0; i < IPW_B; ++i) {
        const int gb = wave * IPW_B + i;
        const bf16_t* src = wsrc[i] + (size_t)gi_kt * BKE;
        src = nok[i] ? src : zp;
        dma16(src, sbase + BM * 128 + gb * 1024);
      }
      if constexpr (ALIGNED) {
        s_c0 += BKE;
        if (s_c0 >= a.Cin) {
          s_c0 = 0;
          if (++s_kw == a.KW) {
            s_kw = 0;
            ++s_kh;
          }
        }
      } else {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          kc[par] += BKE;
          while (kc[par] >= a.Cin) {
            kc[par] -= a.Cin;
            if (++kkw[par] == a.KW) {
              kkw[par] = 0;
              ++kkh[par];
            }
          }
        }
      }
      ++gi;
      if (++gi_kt == nk) {
        gi_kt = 0;
        ++gi_r;
      }
    };
#pragma unroll
    for (int p = 0; p < AHEAD; ++p)
      if (gi < total) issue();
    for (int g = 0; g < total; ++g) {
      const int pending = min(AHEAD - 1, total - 1 - g);
      if (pending >= 2)
        wait_vmcnt<2 * LPT>();
      else if (pending == 1)
        wait_vmcnt<LPT>();
      else
        wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (gi < total) issue();
    }
    return;
  }

  // ---- MFMA waves ---------------------------------------------------------------------------------------------------------
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t sw = (fr >> 1) & 7;
  const uint32_t x_off0 = (wm * (BM / WM) + fr) * 128 + (((0 + fg) ^ sw) & 7) * 16;
  const uint32_t x_off1 = (wm * (BM / WM) + fr) * 128 + (((4 + fg) ^ sw) & 7) * 16;
  const uint32_t w_off0 = (BM + wn * (BN / WN) + fr) * 128 + (((0 + fg) ^ sw) & 7) * 16;
  const uint32_t w_off1 = (BM + wn * (BN / WN) + fr) * 128 + (((4 + fg) ^ sw) & 7) * 16;
  int mrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = bm0 + wm * (BM / WM) + j * 16 + (lane & 15);
    mrow[j] = m < a.M ? m : -1;
  }
  int g = 0;
  for (int r = r_begin; r < r_end; ++r) {
    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt, ++g) {
      __builtin_amdgcn_s_barrier();
      const uint32_t sb = lds0 + (g % NSTAGE) * STAGE_BYTES;
      u32x4_t xf0[TM], xf1[TM], wf0[TN], wf1[TN];
      static_assert(TM + TN <= 15, "lgkmcnt holds 4 bits");
      static_for<0, TN>([&](auto i) { wf0[i] = lds_read128<decltype(i)::value * 2048>(sb + w_off0); });
      static_for<0, TM>([&](auto j) { xf0[j] = lds_read128<decltype(j)::value * 2048>(sb + x_off0); });
      static_for<0, TN>([&](auto i) { wf1[i] = lds_read128<decltype(i)::value * 2048>(sb + w_off1); });
      static_for<0, TM>([&](auto j) { xf1[j] = lds_read128<decltype(j)::value * 2048>(sb + x_off1); });
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf0[i]),
                                                              __builtin_bit_cast(bf16x8_t, xf0[j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf1[i]),
                                                              __builtin_bit_cast(bf16x8_t, xf1[j]), acc[i][j], 0, 0, 0);
    }
    int p = 0;
    for (int i = 1; i < n_members; ++i)
      if (r >= args[i].blk0) p = i;
    conv_store_tiles<TN, TM>(args[p], acc, (r - args[p].blk0) * BN + wn * (BN / WN), (lane >> 4) * 4, mrow);
  }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(dma_threads(WM, WN, NSTAGE)) void conv_igemm_dma_walk_kernel(const ConvArgs* __restrict__ args, int n, int total,
                                                                                         int split) {
  // split 2 ("paired walk"): the walk of a pixel tile is shared by two workgroups with consecutive logical ids -- they start
  // together on ONE XCD, stream the pixel rows at the same time (one fetch from the memory side for both) and re-read them
  // from an L2 that holds 16 pixel tiles per XCD instead of 32 (K = 768: 196 KB each)
  const int l = xcd_tile_index(total);
  if (l < 0) return;
  const int bm = l / split, part = l - bm * split;
  const int R = args[0].grp_nt;                  // out-channel tiles of all members
  const int r0 = R * part / split, r1 = R * (part + 1) / split;
  if (r0 >= r1) return;
  if (args[0].Cin % 64 == 0)
    conv_igemm_dma_walk_body<BM, BN, WM, WN, NSTAGE, true>(args, n, bm, r0, r1);
  else
    conv_igemm_dma_walk_body<BM, BN, WM, WN, NSTAGE, false>(args, n, bm, r0, r1);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, bool ALIGNED>
__global__ __launch_bounds__(dma_threads(WM, WN, NSTAGE)) void conv_igemm_dma_kernel(ConvArgs a) {
  const int tiles_n = (a.Cout + BN - 1) / BN;
  int bm, bn;
  if (a.remap) {
    const int l = xcd_tile_index(a.tiles_m * tiles_n);
    if (l < 0) return;
    bn = l % tiles_n;
    bm = l / tiles_n;
  } else {
    if ((int)blockIdx.x >= a.tiles_m * tiles_n) return;
    bm = blockIdx.x % a.tiles_m;
    bn = blockIdx.x / a.tiles_m;
  }
  conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, ALIGNED>(a, bm, bn);
}

// Grouped launch: blockIdx.z selects one of several independent convolutions (the same-depth
// ops of the parallel Inception branches) whose argument records live in device memory.  One
// launch then carries 2-4x the workgroups of a single 12x12 / 5x5 layer, which is what those
// layers lack to fill 256 CUs at batch 64.
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(dma_threads(WM, WN, NSTAGE)) void conv_igemm_dma_grouped_kernel(const ConvArgs* __restrict__ args, int n, int total) {
  int bid = blockIdx.x;
  const int remap = args[0].remap;
  if (remap) {
    bid = xcd_tile_index(total);
    if (bid < 0) return;
  } else if (bid >= total) {
    return;
  }
  if (remap == 2) {
    // shared input: logical id = (pixel tile, out-channel tile over ALL members), pixel tile slowest, and every
    // XCD owns a contiguous range of logical ids -- the members' workgroups of one pixel tile run together on one
    // XCD, so the activation rows come from the memory side once instead of once per member
    const int R = args[0].grp_nt;
    const int bm = bid / R, r = bid - bm * R;
    int p = 0;
    for (int i = 1; i < n; ++i)
      if (r >= args[i].blk0) p = i;
    const ConvArgs a = args[p];
    if (a.Cin % 64 == 0)
      conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, true>(a, bm, r - a.blk0);
    else
      conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, false>(a, bm, r - a.blk0);
    return;
  }
  int p = 0;
  for (int i = 1; i < n; ++i)
    if (bid >= args[i].blk0) p = i;
  const ConvArgs a = args[p];
  const int local = bid - a.blk0;
  if (a.member_kind == 1) {
    pool_bn_relu_member(a, local);
    return;
  }
  int bm, bn;
  if (remap) {
    const int tiles_n = (a.Cout + BN - 1) / BN;
    bn = local % tiles_n;
    bm = local / tiles_n;
  } else {
    // the out-channel tiles of a pixel tile 8 workgroups apart: consecutive workgroup ids go round the 8 XCDs, so b and
    // b + 8 start together on ONE XCD and the second reads the im2col rows the first has just pulled into that L2
    // (3x3 / 2 288 -> 384 of Mixed_6a at 1280 images: 2.5 GB of reads per launch, every tap of either tile from the memory side)
    // The pairing is between `local` and `local + 8`, i.e. hardware ids bid and bid + 8 (this branch runs without the
    // xcd_tile_index remap: bid == blockIdx.x), which share an XCD whatever a.blk0 is -- a member need not start on a
    // multiple of 8; only WHICH XCD a pair lands on moves with blk0.
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int full = a.tiles_m & ~7;
    if (local < full * tiles_n) {
      bn = (local >> 3) % tiles_n;
      bm = (local / (8 * tiles_n)) * 8 + (local & 7);
    } else {
      const int l2 = local - full * tiles_n, r = a.tiles_m - full;
      bm = full + l2 % r;
      bn = l2 / r;
    }
  }
  if (a.Cin % 64 == 0)
    conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, true>(a, bm, bn);
  else
    conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, false>(a, bm, bn);
}

// comic_cnn_op::min_lds: lower bound on the dynamic LDS a DMA conv workgroup requests.  84 KiB admits one
// workgroup per CU instead of 2-3: a forward that runs on a second stream UNDER the decoder step then leaves
// wave slots, registers and LDS for the latency-bound decoder kernels (measured: forward alone 1.44 -> 1.79 ms,
// overlapped training step 3.67 -> 3.53 ms).  Carried per op (ConvArgs::min_lds), no process state.

#include "conv_patch.inc"

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_dma_grouped(const ConvArgs* args_dev, int n, int total_blocks, int min_lds, hipStream_t st) {
  constexpr int lds0 = ring_stages(NSTAGE) * (BM + BN) * 128;
  const int lds = std::max(lds0, min_lds);
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_igemm_dma_grouped_kernel<BM, BN, WM, WN, NSTAGE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("conv: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_dma_grouped_kernel<BM, BN, WM, WN, NSTAGE>), dim3((total_blocks + 7) / 8 * 8),
                     dim3(dma_threads(WM, WN, NSTAGE)), lds, st, args_dev, n, total_blocks);
  return 0;
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_dma_walk(const ConvArgs* args_dev, int n, int total_blocks, int min_lds, hipStream_t st, int split = 1) {
  constexpr int lds0 = ring_stages(NSTAGE) * (BM + BN) * 128;
  const int lds = std::max(lds0, min_lds);
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_igemm_dma_walk_kernel<BM, BN, WM, WN, NSTAGE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("conv: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm_dma_walk_kernel<BM, BN, WM, WN, NSTAGE>), dim3((total_blocks + 7) / 8 * 8),
                     dim3(dma_threads(WM, WN, NSTAGE)), lds, st, args_dev, n, total_blocks, split);
  return 0;
}

template <int BM, int BN, int WM, int WN, int NSTAGE = 3>
int launch_dma(const ConvArgs& a, hipStream_t st) {
  static_assert(ring_stages(NSTAGE) >= 2 && ring_stages(NSTAGE) <= 4, "pipeline depth");
  constexpr int lds0 = ring_stages(NSTAGE) * (BM + BN) * 128;
  const int lds = std::max(lds0, a.min_lds);
  static_assert(lds0 <= 160 * 1024, "LDS");
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<BM, BN, WM, WN, NSTAGE, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<BM, BN, WM, WN, NSTAGE, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("conv: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  ConvArgs b = a;
  b.tiles_m = cdiv(a.M, BM);
  const long total = (long)b.tiles_m * cdiv(a.Cout, BN);
  if (total >= (1L << 31) - 8) {
    comic_set_error("conv: too many tiles");
    return 1;
  }
  dim3 grid((unsigned)((total + 7) / 8 * 8));
  if (a.Cin % 64 == 0)
    hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WM, WN, NSTAGE, true>), grid, dim3(dma_threads(WM, WN, NSTAGE)), lds, st, b);
  else
    hipLaunchKernelGGL((conv_igemm_dma_kernel<BM, BN, WM, WN, NSTAGE, false>), grid, dim3(dma_threads(WM, WN, NSTAGE)), lds, st, b);
  return 0;
}

// Backward-data launches fused with the producer's activation gradient (ConvArgs::mask_y): the im2col kernel with the MASK
// epilogue, the tile shapes the backward of small batches uses.
template <int BM, int BN, int WM, int WN, int NSTAGE, bool ALIGNED>
__global__ __launch_bounds__(dma_threads(WM, WN, NSTAGE)) void conv_igemm_dma_mask_kernel(ConvArgs a) {
  const int tiles_n = (a.Cout + BN - 1) / BN;
  const int l = xcd_tile_index(a.tiles_m * tiles_n);
  if (l < 0) return;
  conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, ALIGNED, true>(a, l / tiles_n, l % tiles_n);
}

template <int BM, int BN, int WM, int WN, int NSTAGE = 3>
int launch_dma_mask(const ConvArgs& a, hipStream_t st) {
  constexpr int lds0 = ring_stages(NSTAGE) * (BM + BN) * 128;
  const int lds = std::max(lds0, a.min_lds);
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_igemm_dma_mask_kernel<BM, BN, WM, WN, NSTAGE, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv_igemm_dma_mask_kernel<BM, BN, WM, WN, NSTAGE, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      comic_set_error("conv: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  ConvArgs b = a;
  b.remap = 1;
  b.tiles_m = cdiv(a.M, BM);
  const long total = (long)b.tiles_m * cdiv(a.Cout, BN);
  if (total >= (1L << 31) - 8) {
    comic_set_error("conv: too many tiles");
    return 1;
  }
  dim3 grid((unsigned)((total + 7) / 8 * 8));
  if (a.Cin % 64 == 0)
    hipLaunchKernelGGL((conv_igemm_dma_mask_kernel<BM, BN, WM, WN, NSTAGE, true>), grid, dim3(dma_threads(WM, WN, NSTAGE)), lds, st, b);
  else
    hipLaunchKernelGGL((conv_igemm_dma_mask_kernel<BM, BN, WM, WN, NSTAGE, false>), grid, dim3(dma_threads(WM, WN, NSTAGE)), lds, st, b);
  return 0;
}

int dispatch_igemm_dma_mask(const ConvArgs& a, hipStream_t st) {
  const long b128x128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
  const long b128x64 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 64);
  const long b64x64 = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64);
  if (a.Cout <= 32) return launch_dma_mask<128, 32, 4, 1>(a, st);
  if (a.Cout % 128 == 0 && b128x128 >= 512) return launch_dma_mask<128, 128, 2, 2>(a, st);
  if (b128x64 >= 512) return launch_dma_mask<128, 64, 2, 2>(a, st);
  if (b64x64 >= 384) return launch_dma_mask<64, 64, 2, 2>(a, st);
  return launch_dma_mask<32, 64, 1, 4>(a, st);
}

// Image-resident kernel (conv_img.hip): n convs of ONE shape (same source geometry, Cin, Cout) as one launch.
int img_config(const ConvArgs& a) {
  if (!a.w_frag || a.accum || a.mask_y || a.x_cs % 8 != 0 || a.x_co % 8 != 0) return -1;
  return comic_img_config(a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.SH, a.SW, a.Ho, a.Wo);
}
bool img_same_shape(const ConvArgs& p, const ConvArgs& q) {
  return p.H == q.H && p.W == q.W && p.Cin == q.Cin && p.Cout == q.Cout && p.B == q.B;
}
int launch_img_convs(const ConvArgs* a, int n, hipStream_t st) {
  const int cfg = img_config(a[0]);
  if (cfg < 0 || n < 1 || n > kImgMaxMembers) {
    comic_set_error("conv: layer not eligible for the image-resident kernel (%dx%d, %dx%d, Cin %d, Cout %d%s)", a[0].H, a[0].W,
                    a[0].KH, a[0].KW, a[0].Cin, a[0].Cout, a[0].w_frag ? "" : ", no fragment-order weights");
    return 3;
  }
  ComicImgArgs ia;
  memset(&ia, 0, sizeof(ia));
  ia.n_members = n;
  ia.B = a[0].B; ia.H = a[0].H; ia.W = a[0].W; ia.Cin = a[0].Cin; ia.Cout = a[0].Cout;
  ia.G = comic_img_images_per_group(cfg);
  ia.groups = cdiv(ia.B, ia.G);
  const int pxb = a[0].Cin * 2;
  ia.PXBp = pxb + ((pxb % 64 == 0) ? 32 : 0);
  ia.KS32 = a[0].Kpad / 32;
  for (int j = 0; j < n; ++j) {
    if (img_config(a[j]) != cfg || !img_same_shape(a[j], a[0])) {
      comic_set_error("conv: the members of an image-resident launch must have one shape");
      return 3;
    }
    ComicImgMember& m = ia.m[j];
    m.x = (const bf16_t*)a[j].x; m.wf = (const bf16_t*)a[j].w_frag; m.scale = a[j].scale; m.shift = a[j].shift; m.y = a[j].y;
    m.x_cs = a[j].x_cs; m.x_co = a[j].x_co; m.y_cs = a[j].y_cs; m.y_co = a[j].y_co;
    m.KH = a[j].KH; m.KW = a[j].KW; m.PT = a[j].PT; m.PL = a[j].PL; m.relu = a[j].relu; m.out_f32 = a[j].out_f32;
  }
  return comic_img_launch(cfg, ia, st);
}

int validate_grouped_conv(const comic_cnn_op* op, int xc, int yc, const comic_conv_weight* wt, int batch);
// Chains of image-resident convs (COMIC_CHAIN_TILE, conv_img.hip conv_img_chain_kernel): the n ops of the group are one or
// two chains, chain after chain; an op with COMIC_OP_CHAIN_LINK hands its output to the NEXT op of the table through the LDS
// (its dst buffer is not written), an op without it ends its chain and stores to its dst slice.
int launch_img_chains(const comic_cnn_op* op, int n, void* const* buffers, const int32_t* buf_channels,
                      const comic_conv_weight* weights, int batch, hipStream_t st) {
  ComicChainArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.B = batch; ca.H = op[0].H; ca.W = op[0].W; ca.Cin = op[0].Cin;
  const int pxb = ca.Cin * 2;
  ca.PXBp = pxb + ((pxb % 64 == 0) ? 32 : 0);
  COMIC_REQUIRE(comic_img_chain_supported(ca.H, ca.W, ca.Cin), "conv chain: %dx%d maps of %d channels are not supported", ca.H,
                ca.W, ca.Cin);
  int nm = 0;
  for (int j = 0; j < n;) {
    COMIC_REQUIRE(nm < kChainMaxMembers, "conv chain: more than %d chains in one launch", kChainMaxMembers);
    ComicChainMember& m = ca.m[nm];
    int len = 0;
    for (;; ++len) {
      COMIC_REQUIRE(j + len < n && len < kChainMaxConvs, "conv chain: a chain of more than %d convs, or a link flag on the last op",
                    kChainMaxConvs);
      const comic_cnn_op* o = op + j + len;
      const comic_conv_weight* wt = weights + o->weight;
      const bool link = (o->flags & COMIC_OP_CHAIN_LINK) != 0;
      if (int rc = validate_grouped_conv(o, buf_channels[o->src], buf_channels[o->dst], wt, batch)) return rc;
      COMIC_REQUIRE(o->SH == 1 && o->SW == 1 && o->Ho == o->H && o->Wo == o->W && o->H == ca.H && o->W == ca.W &&
                        o->Cin == ca.Cin && o->KH * o->KW >= 2 && o->KH * o->KW <= 32 && !(o->flags & (COMIC_OP_RAW | COMIC_OP_X3)),
                    "conv chain: every conv must be a stride-1 SAME conv with BatchNorm over the chain's %dx%dx%d input", ca.H, ca.W,
                    ca.Cin);
      COMIC_REQUIRE(wt->w_frag, "conv chain: no fragment-order weights");
      COMIC_REQUIRE(o->Cout == (link ? ca.Cin : 192), "conv chain: inner convs keep the channel count, the last one has 192 outputs");
      COMIC_REQUIRE(!link || (!o->out_f32 && o->dst == o[1].src && o->dst_coff == 0 && o[1].src_coff == 0 &&
                              buf_channels[o->dst] == ca.Cin),
                    "conv chain: a linked conv must feed the next op of the table");
      ComicChainConv& c = m.c[len];
      c.wf = (const bf16_t*)wt->w_frag; c.scale = wt->scale; c.shift = wt->shift;
      c.KH = o->KH; c.KW = o->KW; c.PT = o->PT; c.PL = o->PL; c.Cout = o->Cout; c.relu = o->relu;
      c.KS32 = ((o->KH * o->KW * o->Cin + 63) / 64 * 64) / 32;
      c.keep = nullptr;
      if (link && (o->flags & COMIC_OP_CHAIN_KEEP)) {
        COMIC_REQUIRE(buffers[o->dst] && buf_channels[o->dst] % 4 == 0, "conv chain: a kept intermediate map needs its buffer");
        c.keep = buffers[o->dst]; c.keep_cs = buf_channels[o->dst]; c.keep_co = o->dst_coff;
      }
      if (len == 0) {
        COMIC_REQUIRE(buf_channels[o->src] % 8 == 0 && o->src_coff % 8 == 0 && buffers[o->src], "conv chain: bad source slice");
        m.x = (const bf16_t*)buffers[o->src]; m.x_cs = buf_channels[o->src]; m.x_co = o->src_coff;
      }
      if (!link) {
        COMIC_REQUIRE(buffers[o->dst], "conv chain: null destination");
        m.y = buffers[o->dst]; m.y_cs = buf_channels[o->dst]; m.y_co = o->dst_coff; m.out_f32 = o->out_f32;
        ++len;
        break;
      }
    }
    m.n_convs = len;
    j += len;
    ++nm;
  }
  if (nm == 2 && ca.m[1].n_convs > ca.m[0].n_convs) std::swap(ca.m[0], ca.m[1]);     // the longer chain's workgroups first
  ca.n_members = nm;
  return comic_img_chain_launch(ca, st);
}

// explicit tile selection (comic_cnn_op.tile, filled by the host-side autotuner)
constexpr int kNumConvTiles = 12;
// ids 26..28: two-stage wide im2col tiles.  The L2 -> LDS fill (about 30 B/clk/CU) bounds the im2col kernel: a k-tile
// costs (BM+BN)*128 bytes of fill for BM*BN/32 MFMA cycles, so 64x128 cannot pass ~31 % of the MFMA peak, 128x128
// 47 %, 128x192 56 %; with two stages (instead of three) two such workgroups still share a CU.
constexpr int kWideTile0 = 26, kNumWideTiles = 22;      // 35..47: four MFMA waves + four loader waves      // 29..31: 8 waves, one workgroup per CU (fill bound 62 / 80 / 94 %)
// 56..58: "walk" forms of 44 / 38 / 35 for launches whose members share their im2col matrix (conv_igemm_dma_walk_body)
constexpr int kWalkTile0 = 56, kNumWalkTiles = 6;      // 56..58: one workgroup per pixel tile; 59..61: two (paired walk)
inline int walk_split(int t) { return t >= kWalkTile0 + 3 ? 2 : 1; }
inline bool is_walk_tile(int t) { return t >= kWalkTile0 && t < kWalkTile0 + kNumWalkTiles; }
inline bool is_im2col_tile(int t) { return t <= kNumConvTiles || (t >= kWideTile0 && t < kWideTile0 + kNumWideTiles) || is_walk_tile(t); }
int launch_dma_tile(int tile, const ConvArgs& a, hipStream_t st) {
  switch (tile) {
    case 26: return launch_dma<128, 128, 2, 2, 2>(a, st);
    case 27: return launch_dma<128, 192, 2, 2, 2>(a, st);
    case 28: return launch_dma<192, 128, 2, 2, 2>(a, st);
    case 29: return launch_dma<256, 128, 4, 2, 2>(a, st);
    case 30: return launch_dma<256, 192, 2, 4, 2>(a, st);
    case 31: return launch_dma<256, 256, 2, 4, 2>(a, st);
    case 32: return launch_dma<128, 160, 2, 2, 2>(a, st);      // 160-channel layers without a ragged column tile
    case 33: return launch_dma<256, 64, 4, 1, 2>(a, st);
    case 34: return launch_dma<192, 96, 2, 2, 2>(a, st);
    case 35: return launch_dma<128, 192, 2, 2, kLoaderWaves + 3>(a, st);
    case 36: return launch_dma<128, 128, 2, 2, kLoaderWaves + 3>(a, st);
    case 37: return launch_dma<128, 256, 2, 2, kLoaderWaves + 3>(a, st);
    case 38: return launch_dma<192, 128, 2, 2, kLoaderWaves + 3>(a, st);
    case 39: return launch_dma<128, 160, 2, 2, kLoaderWaves + 3>(a, st);
    case 40: return launch_dma<192, 192, 2, 2, kLoaderWaves + 2>(a, st);
    case 41: return launch_dma<128, 192, 2, 2, kLoaderWaves + 4>(a, st);
    case 42: return launch_dma<160, 192, 2, 2, kLoaderWaves + 3>(a, st);
    case 43: return launch_dma<192, 160, 2, 2, kLoaderWaves + 3>(a, st);
    case 44: return launch_dma<192, 192, 2, 2, kLoaderWaves + 3>(a, st);
    case 45: return launch_dma<64, 128, 2, 2, kLoaderWaves + 3>(a, st);        // small-batch shapes
    case 46: return launch_dma<128, 64, 2, 2, kLoaderWaves + 3>(a, st);
    case 47: return launch_dma<64, 64, 2, 2, kLoaderWaves + 4>(a, st);
    case 1: return launch_dma<128, 128, 2, 2, 3>(a, st);
    case 2: return launch_dma<128, 64, 2, 2, 3>(a, st);
    case 3: return launch_dma<64, 64, 2, 2, 3>(a, st);
    case 4: return launch_dma<32, 64, 1, 4, 3>(a, st);
    case 5: return launch_dma<128, 32, 4, 1, 3>(a, st);
    case 6: return launch_dma<64, 128, 2, 2, 3>(a, st);
    case 7: return launch_dma<256, 64, 4, 1, 3>(a, st);
    case 8: return launch_dma<128, 64, 2, 2, 4>(a, st);
    case 9: return launch_dma<64, 64, 2, 2, 4>(a, st);
    case 10: return launch_dma<32, 64, 1, 4, 4>(a, st);
    case 11: return launch_dma<64, 128, 2, 2, 4>(a, st);
    case 12: return launch_dma<128, 32, 4, 1, 4>(a, st);
    // patch-resident variants (stride 1, Cin >= 32): 64*TM pixels x 16*TN channels
    case 13: return launch_patch<4, 4>(a, st);
    case 14: return launch_patch<4, 2>(a, st);
    case 15: return launch_patch<4, 6>(a, st);
    case 16: return launch_patch<2, 4>(a, st);
    case 17: return launch_patch<2, 2>(a, st);
    case 18: return launch_patch<2, 6>(a, st);
    // patch-resident, 8 / 12 waves per workgroup (2-3 per SIMD) over one patch
    case 19: return launch_patch<4, 4, 4, 2>(a, st);    // 256 px x 128 ch
    case 20: return launch_patch<4, 4, 4, 3>(a, st);    // 256 px x 192 ch
    case 21: return launch_patch<4, 4, 8, 1>(a, st);    // 512 px x 64 ch
    case 22: return launch_patch<4, 2, 4, 2>(a, st);    // 256 px x 64 ch, 8 waves
    case 23: return launch_patch<4, 6, 4, 2>(a, st);    // 256 px x 192 ch, 8 waves
    case 24: return launch_patch<2, 4, 4, 2>(a, st);    // 128 px x 128 ch
    case 25: return launch_patch<2, 6, 4, 2>(a, st);    // 128 px x 192 ch, 8 waves
    // patch-resident, four MFMA waves + four loader waves
    case 48: return launch_patch<4, 4, 4, 1, kLoaderWaves + 3>(a, st);   // 256 px x 64 ch
    case 49: return launch_patch<4, 2, 4, 1, kLoaderWaves + 3>(a, st);   // 256 px x 32 ch
    case 50: return launch_patch<4, 6, 4, 1, kLoaderWaves + 3>(a, st);   // 256 px x 96 ch
    case 51: return launch_patch<2, 6, 4, 1, kLoaderWaves + 3>(a, st);   // 128 px x 96 ch
    case 52: return launch_patch<4, 6, 2, 2, kLoaderWaves + 3>(a, st);   // 128 px x 192 ch
    case 53: return launch_patch<4, 4, 2, 2, kLoaderWaves + 3>(a, st);   // 128 px x 128 ch
    case COMIC_IMG_TILE: return launch_img_convs(&a, 1, st);             // image-resident (conv_img.hip)
    default:
      comic_set_error("conv: unknown tile id %d", tile);
      return 2;
  }
}

// (BM, BN) of the explicit tile ids above
constexpr int kTileBM[kNumConvTiles + 1] = {0, 128, 128, 64, 32, 128, 64, 256, 128, 64, 32, 64, 128};
constexpr int kTileBN[kNumConvTiles + 1] = {0, 128, 64, 64, 64, 32, 128, 64, 64, 64, 64, 128, 32};
constexpr int kWideBM[kNumWideTiles] = {128, 128, 192, 256, 256, 256, 128, 256, 192, 128, 128, 128, 192, 128, 192, 128, 160, 192, 192, 64, 128, 64};
constexpr int kWideBN[kNumWideTiles] = {128, 192, 128, 128, 192, 256, 160, 64, 96, 192, 128, 256, 128, 160, 192, 192, 192, 160, 192, 128, 64, 64};
constexpr int kWalkBase[kNumWalkTiles] = {44, 38, 35, 44, 38, 35};
inline int im2col_tile_threads(int t) { return (t >= 29 && t <= 31) || t >= 35 ? 512 : 256; }
inline int tile_bm(int t) { return is_walk_tile(t) ? kWideBM[kWalkBase[t - kWalkTile0] - kWideTile0] : t >= kWideTile0 ? kWideBM[t - kWideTile0] : kTileBM[t]; }
inline int tile_bn(int t) { return is_walk_tile(t) ? kWideBN[kWalkBase[t - kWalkTile0] - kWideTile0] : t >= kWideTile0 ? kWideBN[t - kWideTile0] : kTileBN[t]; }

int launch_dma_grouped_tile(int tile, const ConvArgs* args_dev, int n, int total_blocks, int min_lds, hipStream_t st) {
  switch (tile) {
    case 26: return launch_dma_grouped<128, 128, 2, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 27: return launch_dma_grouped<128, 192, 2, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 28: return launch_dma_grouped<192, 128, 2, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 29: return launch_dma_grouped<256, 128, 4, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 30: return launch_dma_grouped<256, 192, 2, 4, 2>(args_dev, n, total_blocks, min_lds, st);
    case 31: return launch_dma_grouped<256, 256, 2, 4, 2>(args_dev, n, total_blocks, min_lds, st);
    case 32: return launch_dma_grouped<128, 160, 2, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 33: return launch_dma_grouped<256, 64, 4, 1, 2>(args_dev, n, total_blocks, min_lds, st);
    case 34: return launch_dma_grouped<192, 96, 2, 2, 2>(args_dev, n, total_blocks, min_lds, st);
    case 35: return launch_dma_grouped<128, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 36: return launch_dma_grouped<128, 128, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 37: return launch_dma_grouped<128, 256, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 38: return launch_dma_grouped<192, 128, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 39: return launch_dma_grouped<128, 160, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 40: return launch_dma_grouped<192, 192, 2, 2, kLoaderWaves + 2>(args_dev, n, total_blocks, min_lds, st);
    case 41: return launch_dma_grouped<128, 192, 2, 2, kLoaderWaves + 4>(args_dev, n, total_blocks, min_lds, st);
    case 42: return launch_dma_grouped<160, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 43: return launch_dma_grouped<192, 160, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 44: return launch_dma_grouped<192, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 45: return launch_dma_grouped<64, 128, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 46: return launch_dma_grouped<128, 64, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 47: return launch_dma_grouped<64, 64, 2, 2, kLoaderWaves + 4>(args_dev, n, total_blocks, min_lds, st);
    case 56: return launch_dma_walk<192, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 57: return launch_dma_walk<192, 128, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 58: return launch_dma_walk<128, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st);
    case 59: return launch_dma_walk<192, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st, 2);
    case 60: return launch_dma_walk<192, 128, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st, 2);
    case 61: return launch_dma_walk<128, 192, 2, 2, kLoaderWaves + 3>(args_dev, n, total_blocks, min_lds, st, 2);
    case 1: return launch_dma_grouped<128, 128, 2, 2, 3>(args_dev, n, total_blocks, min_lds, st);
    case 2: return launch_dma_grouped<128, 64, 2, 2, 3>(args_dev, n, total_blocks, min_lds, st);
    case 3: return launch_dma_grouped<64, 64, 2, 2, 3>(args_dev, n, total_blocks, min_lds, st);
    case 4: return launch_dma_grouped<32, 64, 1, 4, 3>(args_dev, n, total_blocks, min_lds, st);
    case 5: return launch_dma_grouped<128, 32, 4, 1, 3>(args_dev, n, total_blocks, min_lds, st);
    case 6: return launch_dma_grouped<64, 128, 2, 2, 3>(args_dev, n, total_blocks, min_lds, st);
    case 7: return launch_dma_grouped<256, 64, 4, 1, 3>(args_dev, n, total_blocks, min_lds, st);
    case 8: return launch_dma_grouped<128, 64, 2, 2, 4>(args_dev, n, total_blocks, min_lds, st);
    case 9: return launch_dma_grouped<64, 64, 2, 2, 4>(args_dev, n, total_blocks, min_lds, st);
    case 10: return launch_dma_grouped<32, 64, 1, 4, 4>(args_dev, n, total_blocks, min_lds, st);
    case 11: return launch_dma_grouped<64, 128, 2, 2, 4>(args_dev, n, total_blocks, min_lds, st);
    case 12: return launch_dma_grouped<128, 32, 4, 1, 4>(args_dev, n, total_blocks, min_lds, st);
    default:
      comic_set_error("conv: unknown tile id %d", tile);
      return 2;
  }
}

// Tile of a group: the explicit id of its first member, else the same fill rule as the
// single-conv heuristic applied to the group's total tile count.

// ---- weight-stationary 1x1 groups (conv_ws.hip) ---------------------------------------------------------------
// ops[0..n) qualify when they are 1x1 / stride-1 / unpadded convolutions over the SAME source slice (the convs at
// the head of an Inception block, or a single conv), Cin and the concatenated Cout fit the register-resident
// weight layout, and -- with COMIC_OP_POOLED_SRC -- every member reads through the same 3x3 / 2 VALID max-pool.
bool ws_group_eligible(const comic_cnn_op* ops, int n) {
  if (n < 1 || n > 4) return false;
  int tiles = 0;
  for (int j = 0; j < n; ++j) {
    const comic_cnn_op& o = ops[j];
    if (o.kind != 0 || o.KH != 1 || o.KW != 1 || o.SH != 1 || o.SW != 1 || o.PT != 0 || o.PL != 0) return false;
    if (o.src != ops[0].src || o.src_coff != ops[0].src_coff || o.Cin != ops[0].Cin || o.H != ops[0].H ||
        o.W != ops[0].W || o.Ho != ops[0].Ho || o.Wo != ops[0].Wo || (o.flags & COMIC_OP_POOLED_SRC) != (ops[0].flags & COMIC_OP_POOLED_SRC))
      return false;
    if (o.Cout % 16 != 0 || (o.flags & COMIC_OP_X3)) return false;   // (conv_ws.hip has its own epilogue: plain stores)
    tiles += o.Cout / 16;
  }
  const comic_cnn_op& o = ops[0];
  if (o.flags & COMIC_OP_POOLED_SRC) {
    if (o.Ho != (o.H - 3) / 2 + 1 || o.Wo != (o.W - 3) / 2 + 1) return false;
  } else if (o.Ho != o.H || o.Wo != o.W) {
    return false;
  }
  return comic_ws_supported(o.Cin, tiles);
}

int run_ws_group(const comic_cnn_op* ops, int n, void* const* buffers, const int32_t* buf_channels,
                 const comic_conv_weight* weights, int batch, hipStream_t st) {
  COMIC_REQUIRE(ws_group_eligible(ops, n), "conv_ws: ops are not an eligible 1x1 group (kind %d, %dx%d, Cin %d)", ops[0].kind,
                ops[0].KH, ops[0].KW, ops[0].Cin);
  ComicWsArgs a;
  memset(&a, 0, sizeof(a));
  const comic_cnn_op& o = ops[0];
  const int xc = buf_channels[o.src];
  COMIC_REQUIRE(buffers[o.src] && o.src_coff + o.Cin <= xc && xc % 8 == 0 && o.src_coff % 8 == 0, "conv_ws: bad source slice");
  COMIC_REQUIRE((long)batch * o.H * o.W * xc * 2 < (1L << 31), "conv_ws: activation tensor too large");
  a.x = (const bf16_t*)buffers[o.src];
  a.B = batch; a.H = o.H; a.W = o.W; a.x_cs = xc; a.x_co = o.src_coff; a.Cin = o.Cin;
  a.Kpad = (o.Cin + 63) / 64 * 64;
  a.Ho = o.Ho; a.Wo = o.Wo; a.M = batch * o.Ho * o.Wo;
  a.pooled = (o.flags & COMIC_OP_POOLED_SRC) ? 1 : 0;
  a.n_members = n;
  a.tiles_m = cdiv(a.M, 64);
  int t0 = 0;
  for (int j = 0; j < n; ++j) {
    const comic_cnn_op& q = ops[j];
    const comic_conv_weight* wt = weights + q.weight;
    const bool raw = (q.flags & COMIC_OP_RAW) != 0;
    const int yc = buf_channels[q.dst];
    COMIC_REQUIRE(wt->w && (raw || (wt->scale && wt->shift)), "conv_ws: missing weights");
    COMIC_REQUIRE(buffers[q.dst] && q.dst_coff + q.Cout <= yc && q.dst_coff % 4 == 0 && yc % 4 == 0, "conv_ws: bad destination slice");
    COMIC_REQUIRE((long)a.M * yc < (1L << 31), "conv_ws: output tensor too large");
    ComicWsMember& m = a.m[j];
    m.w = (const bf16_t*)wt->w;
    m.scale = raw ? nullptr : wt->scale;
    m.shift = raw ? nullptr : wt->shift;
    m.y = buffers[q.dst];
    m.y_cs = yc; m.y_co = q.dst_coff; m.cout = q.Cout; m.relu = q.relu; m.out_f32 = q.out_f32;
    m.tile0 = t0;
    t0 += q.Cout / 16;
  }
  a.n_tiles = t0;
  if (int rc = comic_ws_launch(a, st)) return rc;
  COMIC_LAUNCH_CHECK("conv_ws");
  return 0;
}
// WS is the default (tile id 0) for an eligible group with enough pixel tiles to occupy every CU
bool ws_group_selected(const comic_cnn_op* ops, int n, int batch) {
  if (!ws_group_eligible(ops, n)) return false;
  if (ops[0].flags & COMIC_OP_POOLED_SRC) return true;
  if (ops[0].tile == COMIC_WS_TILE) return true;
  return ops[0].tile == 0 && (long)batch * ops[0].Ho * ops[0].Wo >= 64L * 256;
}

int group_tile(const comic_cnn_op* ops, int n, int batch) {
  if (ops[0].tile == COMIC_IMG_TILE || ops[0].tile == COMIC_CHAIN_TILE) return ops[0].tile;
  if (ops[0].tile > 0 && ops[0].tile < COMIC_WS_TILE) return ops[0].tile;
  if (is_walk_tile(ops[0].tile)) return ops[0].tile;
  bool all128 = true;
  for (int i = 0; i < n; ++i) all128 = all128 && ops[i].Cout % 128 == 0;
  auto blocks = [&](int t) {
    long b = 0;
    for (int i = 0; i < n; ++i)
      b += (long)cdiv(batch * ops[i].Ho * ops[i].Wo, kTileBM[t]) * cdiv(ops[i].Cout, kTileBN[t]);
    return b;
  };
  if (all128 && blocks(1) >= 512) return 1;
  if (blocks(2) >= 512) return 2;
  if (blocks(3) >= 384) return 3;
  return 4;
}

// Workgroup layout of one group member under tile id `tile`: sets a.tiles_m (and the patch geometry for the
// patch-resident ids), returns its workgroup count or -1 when the member is not eligible; *lds = LDS it needs.
long member_blocks(int tile, ConvArgs& a, int* lds) {
  if (a.member_kind == 1) {          // pool + BN + ReLU items, kPoolItemsPerThread per thread of the launch's workgroup size
    const int threads = is_im2col_tile(tile) ? im2col_tile_threads(tile) : kPatchTiles[patch_tile_index(tile)].threads;
    *lds = 0;
    return cdiv64((long)a.M * (a.Cin / 4), (long)threads * kPoolItemsPerThread);
  }
  if (is_im2col_tile(tile)) {
    a.tiles_m = cdiv(a.M, tile_bm(tile));
    *lds = 0;
    return (long)a.tiles_m * cdiv(a.Cout, tile_bn(tile));
  }
  const PatchTile pt = kPatchTiles[patch_tile_index(tile)];
  PatchGeo g;
  if (!patch_geometry(a, pt.BM, pt.BN, 3, g)) return -1;
  apply_geometry(a, g);
  *lds = g.lds_bytes;
  return (long)a.tiles_m * cdiv(a.Cout, pt.BN);
}

int dispatch_igemm_dma(const ConvArgs& a, hipStream_t st) {
  // thin-channel stride-1 layers with many pixels (the 109x109 / 52x52 / 25x25 3x3 and 5x5 convs, forward and
  // backward-data): the patch-resident kernel, variants as the autotuner picks them at batch 64
  if (a.SH == 1 && a.SW == 1 && a.Cin >= 32 && a.Cin <= 96 && a.KH * a.KW > 1 && a.M >= 16384) {
    PatchGeo g;
    if (a.Cout <= 32 && patch_geometry(a, 256, 32, 3, g)) return launch_patch<4, 2>(a, st);
    if (a.Cout == 96 && patch_geometry(a, 128, 96, 3, g)) return launch_patch<2, 6>(a, st);
    if (patch_geometry(a, 256, 64, 3, g)) return launch_patch<4, 2, 4, 2>(a, st);
  }
  const long b128x128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
  const long b128x64 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 64);
  const long b64x64 = (long)cdiv(a.M, 64) * cdiv(a.Cout, 64);
  if (a.Cout <= 32) return launch_dma<128, 32, 4, 1>(a, st);
  if (a.Cout % 128 == 0 && b128x128 >= 512) return launch_dma<128, 128, 2, 2>(a, st);
  if (b128x64 >= 512) return launch_dma<128, 64, 2, 2>(a, st);
  if (b64x64 >= 384) return launch_dma<64, 64, 2, 2>(a, st);
  return launch_dma<32, 64, 1, 4>(a, st);
}

const void* zero_page_address() {
  static const void* p[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!p[dev]) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page)) == hipSuccess) p[dev] = q;
  }
  return p[dev];
}

template <typename T, int BM, int BN, int WM, int WN>
void launch_igemm(const ConvArgs& a, hipStream_t st) {
  dim3 grid(cdiv(a.M, BM), cdiv(a.Cout, BN));
  hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN>), grid, dim3(256), 0, st, a);
}

template <typename T>
void launch_igemm_mask(const ConvArgs& a, hipStream_t st) {     // fused activation gradient (backward of the fp32 plan)
  dim3 grid(cdiv(a.M, 64), cdiv(a.Cout, 64));
  hipLaunchKernelGGL((conv_igemm_kernel<T, 64, 64, 2, 2, true>), grid, dim3(256), 0, st, a);
}

template <typename T>
int dispatch_igemm(const ConvArgs& a, hipStream_t st) {
  // choose the largest tile that still gives >= ~2 workgroups per CU; small layers fall
  // back to 64x64 (XCD note: grid.x = pixel tiles, so workgroups b, b+8 share an L2 and a
  // neighbouring input window).
  const int n64 = cdiv(a.Cout, 64);
  const long blocks_128x128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
  const long blocks_128x64 = (long)cdiv(a.M, 128) * n64;
  if (a.Cout <= 32) {
    launch_igemm<T, 128, 32, 4, 1>(a, st);
  } else if (a.Cout % 128 == 0 && blocks_128x128 >= 512) {
    launch_igemm<T, 128, 128, 2, 2>(a, st);
  } else if (blocks_128x64 >= 512) {
    launch_igemm<T, 128, 64, 2, 2>(a, st);
  } else {
    launch_igemm<T, 64, 64, 2, 2>(a, st);
  }
  return 0;
}

int fill_args(ConvArgs& a, const comic_cnn_op* op, const void* x, int x_channels, void* y, int y_channels,
              const comic_conv_weight* wt, int batch) {
  a.x = x;
  a.y = y;
  a.w = wt ? wt->w : nullptr;
  const bool raw = op->kind == 0 && (op->flags & COMIC_OP_RAW);     // epilogue deferred to a kind-7 op
  a.scale = (wt && !raw) ? wt->scale : nullptr;
  a.shift = (wt && !raw) ? wt->shift : nullptr;
  a.B = batch;
  a.H = op->H; a.W = op->W; a.Cin = op->Cin; a.Cout = op->Cout;
  a.KH = op->KH; a.KW = op->KW; a.SH = op->SH; a.SW = op->SW; a.PT = op->PT; a.PL = op->PL;
  a.Ho = op->Ho; a.Wo = op->Wo;
  a.x_cs = x_channels; a.x_co = op->src_coff; a.y_cs = y_channels; a.y_co = op->dst_coff;
  a.K = op->KH * op->KW * op->Cin;
  a.Kpad = (a.K + 63) / 64 * 64;
  a.M = batch * op->Ho * op->Wo;
  a.relu = op->relu;
  a.out_f32 = op->out_f32;
  a.zero = zero_page_address();
  a.blk0 = 0;
  a.tiles_m = 0;
  a.accum = 0;
  a.mask_y = nullptr;
  a.mask_scale = nullptr;
  a.mask_dbeta = nullptr;
  a.mask_cs = a.mask_co = 0;
  a.member_kind = 0;
  a.w_frag = wt ? wt->w_frag : nullptr;
  a.min_lds = std::min(std::max(op->min_lds, 0), 160 * 1024);
  a.remap = 1;
  // COMIC_OP_X3: bf16 buffers hold [hi | lo | hi] regions of a third of their channels each
  const bool x3 = (op->flags & COMIC_OP_X3) != 0;
  a.x3 = (x3 && !op->out_f32) ? y_channels / 3 : 0;
  a.x3_src = (x3 && op->kind >= 2 && op->kind <= 3) ? x_channels / 3 : 0;
  return 0;
}

// the activation gradient a backward-data launch applies to its result (ConvArgs::mask_*), and where that result goes
struct ConvMask {
  const void* y;            // forward output of the producer conv, rows of y_cs channels, slice from y_co
  int y_cs, y_co;
  const float* scale;       // its BatchNorm scale
  float* dbeta;             // its d beta accumulator
  void* dz;                 // its d-conv tensor [B][Ho][Wo][Cout] (stride 1: not dilated)
};

template <typename T>
int run_op(const comic_cnn_op* op, const void* x, int xc, void* y, int yc, const comic_conv_weight* wt, int batch,
           hipStream_t st, int accum = 0, const ConvMask* mask = nullptr) {
  constexpr int EPC = Elem<T>::EPC;
  ConvArgs a;
  fill_args(a, op, x, xc, y, yc, wt, batch);
  a.accum = accum;
  if (mask) {
    COMIC_REQUIRE(op->kind == 0 && !accum && !op->out_f32 && !a.x3 && !a.scale, "conv: fused activation gradient on a plain backward-data launch only");
    a.mask_y = mask->y;
    a.mask_cs = mask->y_cs;
    a.mask_co = mask->y_co;
    a.mask_scale = mask->scale;
    a.mask_dbeta = mask->dbeta;
  }
  COMIC_REQUIRE(x && y, "cnn op %d: null buffer", op->kind);
  COMIC_REQUIRE((long)batch * op->H * op->W * xc < (1L << 31) && (long)a.M * yc < (1L << 31),
                "cnn op: tensor exceeds 2^31 elements");
  COMIC_REQUIRE(op->src_coff + op->Cin <= xc, "cnn op: source channel slice out of range");
  switch (op->kind) {
    case 0: {
      COMIC_REQUIRE(wt && wt->w && (accum || mask || (op->flags & COMIC_OP_RAW) || (wt->scale && wt->shift)), "conv: missing weights");
      COMIC_REQUIRE(op->Cin % EPC == 0 && op->src_coff % EPC == 0 && xc % EPC == 0,
                    "conv: Cin/offset/stride must be multiples of %d", EPC);
      COMIC_REQUIRE(op->Cout % 16 == 0 && op->dst_coff % 4 == 0 && yc % 4 == 0,
                    "conv: Cout must be a multiple of 16 (got %d)", op->Cout);
      COMIC_REQUIRE(op->dst_coff + op->Cout <= (a.x3 ? a.x3 : yc), "conv: destination channel slice out of range");
      COMIC_REQUIRE(!(op->flags & COMIC_OP_X3) || (sizeof(T) == 2 && yc % 3 == 0 && (yc / 3) % 4 == 0 && !accum),
                    "conv: COMIC_OP_X3 needs a bf16 plan and a destination of three equal channel regions");
      if constexpr (sizeof(T) == 2) {
        COMIC_REQUIRE(a.zero, "conv: zero page symbol not resolvable");
        COMIC_REQUIRE((long)batch * op->H * op->W * xc * 2 < (1L << 31), "conv: activation tensor too large");
        if (mask) {
          if (int rc = dispatch_igemm_dma_mask(a, st)) return rc;
        } else if (op->tile > 0) {
          if (int rc = launch_dma_tile(op->tile, a, st)) return rc;
        } else if (int rc = dispatch_igemm_dma(a, st)) {
          return rc;
        }
      } else if (mask) {
        launch_igemm_mask<T>(a, st);
      } else {
        dispatch_igemm<T>(a, st);
      }
      break;
    }
    case 1: {
      COMIC_REQUIRE(wt && wt->w, "stem conv: missing weights");
      COMIC_REQUIRE(op->Cin <= 4 && op->Cout % 32 == 0, "stem conv: needs Cin<=4, Cout%%32==0");
      COMIC_REQUIRE(op->dst_coff % 4 == 0 && yc % 4 == 0, "stem conv: misaligned destination");
      const size_t lds = (size_t)a.K * a.Cout * sizeof(float);
      COMIC_REQUIRE(lds <= 64 * 1024, "stem conv: weights do not fit LDS");
      if (a.K <= 32 && a.Cout == 32 && a.PT == 0 && a.PL == 0 && (op->Ho - 1) * op->SH + op->KH <= op->H &&
          (op->Wo - 1) * op->SW + op->KW <= op->W && op->tile != 1)
        hipLaunchKernelGGL((conv_stem_mfma_kernel<T, 2>), dim3(cdiv(a.M, 1024)), dim3(256), 0, st, a);   // VALID 3x3x3
      else if (const int lrow = ((op->PL + op->W + op->KW) * op->Cin + kStemKRow + 1) / 2 * 2;   // staged row, floats
               sizeof(T) == 2 && op->tile != 1 && a.Cout == 64 && op->KW * op->Cin <= kStemKRow && op->KH <= 8 &&
               xc == op->Cin && op->src_coff == 0 && (op->W * op->Cin) % 2 == 0 && (op->PL * op->Cin) % 2 == 0 &&
               (op->SW * op->Cin) % 2 == 0 && kStemRows * op->Wo <= 256 &&
               (op->Wo - 1) * op->SW * op->Cin + kStemKRow <= lrow &&          // the last pixel's widest operand read
               ((kStemRows - 1) * op->SH + op->KH) * (op->W * op->Cin / 2) <= kStemFetch * 256 &&
               2 * lrow * ((kStemRows - 1) * op->SH + 8) * 4 <= 64 * 1024) {
        // Inception-V1 Conv2d_1a_7x7 (7x7 / 2, SAME) -- conv_stem_wide_kernel, persistent workgroups (one per CU: the
        // filter fragments take most of the register file)
        const int bpi = cdiv(op->Ho, kStemRows), total = batch * bpi;
        const int lds_w = 2 * lrow * ((kStemRows - 1) * op->SH + 8) * 4;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        hipLaunchKernelGGL((conv_stem_wide_kernel<4, 6>), dim3(std::min(total, cus)), dim3(256), lds_w, st, a, lrow, bpi, total);
      }
      else
        hipLaunchKernelGGL((conv_stem_kernel<T>), dim3(cdiv(a.M, 256)), dim3(256), lds, st, a);
      break;
    }
    case 2:
    case 3: {
      COMIC_REQUIRE(op->Cin % EPC == 0 && op->src_coff % EPC == 0 && xc % EPC == 0 && op->dst_coff % EPC == 0 &&
                        yc % EPC == 0,
                    "pool: channel counts/offsets must be multiples of %d", EPC);
      COMIC_REQUIRE(op->dst_coff + op->Cin <= yc, "pool: destination channel slice out of range");
      const long total = (long)a.M * (op->Cin / EPC);
      if (op->flags & COMIC_OP_X3) {
        if constexpr (sizeof(T) == 2) {
          COMIC_REQUIRE(xc % 3 == 0 && yc % 3 == 0 && a.x3 > 0 && a.x3_src > 0 && op->src_coff + op->Cin <= a.x3_src &&
                            op->dst_coff + op->Cin <= a.x3 && a.x3 % 8 == 0 && a.x3_src % 8 == 0,
                        "pool (x3): channel regions do not fit the buffers");
          if (op->kind == 2)
            hipLaunchKernelGGL((pool_x3_kernel<0>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
          else
            hipLaunchKernelGGL((pool_x3_kernel<1>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
          break;
        } else {
          COMIC_REQUIRE(false, "pool: COMIC_OP_X3 is a bf16-plan layout");
        }
      }
      // row-walking form when there are enough rows to occupy the chip (op->tile: 1 forces it, 2 forces the per-pixel form)
      if (op->kind == 2 && op->KH == 3 && op->KW == 3 && op->SH == 1 && op->SW == 1 && op->PT == 1 && op->PL == 1 &&
          op->Ho == op->H && op->Wo == op->W &&
          (op->tile == 1 || (op->tile == 0 && (long)batch * op->H * (op->Cin / EPC) >= 256L * 256)))
        hipLaunchKernelGGL((maxpool3_rows_kernel<T>), dim3((unsigned)cdiv64((long)batch * op->H * (op->Cin / EPC), 256)), dim3(256),
                           0, st, a);
      else if (op->kind == 2)
        hipLaunchKernelGGL((pool_kernel<T, 0>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
      else
        hipLaunchKernelGGL((pool_kernel<T, 1>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
      break;
    }
    case 4: {
      COMIC_REQUIRE(op->Cin % EPC == 0 && op->src_coff % EPC == 0 && xc % EPC == 0, "global pool: misaligned");
      COMIC_REQUIRE((op->Ho - 1) * op->SH + op->KH <= op->H && (op->Wo - 1) * op->SW + op->KW <= op->W,
                    "global pool: window exceeds input (VALID only)");
      const long total = (long)a.M * (op->Cin / EPC);
      hipLaunchKernelGGL((global_avgpool_kernel<T>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
      break;
    }
    case 7: {
      COMIC_REQUIRE(wt && wt->scale && wt->shift, "pool+bn: missing scale / shift");
      COMIC_REQUIRE(op->KH == 3 && op->KW == 3 && op->SH == 1 && op->SW == 1 && op->PT == 1 && op->PL == 1 &&
                        op->Ho == op->H && op->Wo == op->W && op->Cin == op->Cout,
                    "pool+bn: 3x3 stride-1 SAME only");
      COMIC_REQUIRE(op->Cin % 4 == 0 && op->src_coff % 4 == 0 && xc % 4 == 0 && op->dst_coff % 4 == 0 && yc % 4 == 0,
                    "pool+bn: channel counts/offsets must be multiples of 4");
      COMIC_REQUIRE(op->dst_coff + op->Cin <= (a.x3 ? a.x3 : yc), "pool+bn: destination channel slice out of range");
      const long total = (long)a.M * (op->Cin / 4);
      a.out_f32 = (op->out_f32 || sizeof(T) == 4) ? 1 : 0;
      // row-walking form when there are enough rows to occupy the chip (op->tile: 1 forces it, 2 forces the per-pixel form)
      if (op->tile == 1 || (op->tile == 0 && (long)batch * op->H * (op->Cin / 4) >= 256L * 256))
        hipLaunchKernelGGL(pool_bn_relu_rows_kernel, dim3((unsigned)cdiv64((long)batch * op->H * (op->Cin / 4), 256)), dim3(256),
                           0, st, a);
      else
        hipLaunchKernelGGL(pool_bn_relu_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, a);
      break;
    }
    default:
      COMIC_REQUIRE(false, "unknown cnn op kind %d", op->kind);
  }
  COMIC_LAUNCH_CHECK("cnn op");
  return 0;
}

int validate_grouped_conv(const comic_cnn_op* op, int xc, int yc, const comic_conv_weight* wt, int batch) {
  if (op->kind == 7) {               // pool + BN + ReLU member (same checks as the stand-alone launch)
    COMIC_REQUIRE(wt && wt->scale && wt->shift, "pool+bn: missing scale / shift");
    COMIC_REQUIRE(op->KH == 3 && op->KW == 3 && op->SH == 1 && op->SW == 1 && op->PT == 1 && op->PL == 1 &&
                      op->Ho == op->H && op->Wo == op->W && op->Cin == op->Cout && op->src_f32,
                  "pool+bn: 3x3 stride-1 SAME over an fp32 source only");
    COMIC_REQUIRE(op->Cin % 4 == 0 && op->src_coff % 4 == 0 && xc % 4 == 0 && op->dst_coff % 4 == 0 && yc % 4 == 0,
                  "pool+bn: channel counts/offsets must be multiples of 4");
    COMIC_REQUIRE(op->src_coff + op->Cin <= xc && op->dst_coff + op->Cin <= yc, "pool+bn: channel slice out of range");
    COMIC_REQUIRE((long)batch * op->H * op->W * xc < (1L << 31) && (long)batch * op->Ho * op->Wo * yc < (1L << 31),
                  "pool+bn: tensor too large");
    return 0;
  }
  COMIC_REQUIRE(op->kind == 0, "grouped launch: op kind %d is not a conv", op->kind);
  COMIC_REQUIRE(wt && wt->w && wt->scale && wt->shift, "conv: missing weights");
  COMIC_REQUIRE(op->Cin % 8 == 0 && op->src_coff % 8 == 0 && xc % 8 == 0, "conv: Cin/offset/stride must be multiples of 8");
  COMIC_REQUIRE(op->Cout % 16 == 0 && op->dst_coff % 4 == 0 && yc % 4 == 0, "conv: Cout must be a multiple of 16 (got %d)",
                op->Cout);
  COMIC_REQUIRE(op->src_coff + op->Cin <= xc, "conv: source channel slice out of range");
  COMIC_REQUIRE(op->dst_coff + op->Cout <= yc, "conv: destination channel slice out of range");
  COMIC_REQUIRE((long)batch * op->H * op->W * xc * 2 < (1L << 31) && (long)batch * op->Ho * op->Wo * yc < (1L << 31),
                "conv: activation tensor too large");
  return 0;
}

// All members of a grouped launch are convolutions over the SAME im2col matrix (same source slice, filter window,
// stride and padding -- the 1x1 convs at the head of an Inception block): the launch may then order its
// workgroups by pixel tile (ConvArgs::remap == 2).
bool shared_input_group(const ConvArgs* a, int n) {
  if (n < 2) return false;
  for (int j = 0; j < n; ++j) {
    if (a[j].member_kind != 0) return false;
    if (a[j].x != a[0].x || a[j].x_co != a[0].x_co || a[j].x_cs != a[0].x_cs || a[j].Cin != a[0].Cin ||
        a[j].KH != a[0].KH || a[j].KW != a[0].KW || a[j].SH != a[0].SH || a[j].SW != a[0].SW || a[j].PT != a[0].PT ||
        a[j].PL != a[0].PL || a[j].M != a[0].M || a[j].tiles_m != a[0].tiles_m)
      return false;
  }
  return true;
}

// length of the group run starting at ops[i] (1 when ungrouped)
int group_run(const comic_cnn_op* ops, int n_ops, int i) {
  if (ops[i].group <= 0) return 1;
  int j = i + 1;
  while (j < n_ops && ops[j].group == ops[i].group) ++j;
  return j - i;
}

}  // namespace

extern "C" long comic_cnn_group_args_bytes(const comic_cnn_op* ops, int n_ops) {
  long n = 0;
  for (int i = 0; i < n_ops; ++i) n += ops[i].group > 0;
  return n * (long)sizeof(ConvArgs);
}

extern "C" int comic_cnn_build_group_args(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                                          const int32_t* buf_channels, const comic_conv_weight* weights, int batch,
                                          void* host_out) {
  COMIC_REQUIRE(ops && buffers && buf_channels && weights && host_out, "comic_cnn_build_group_args: null argument");
  ConvArgs* out = (ConvArgs*)host_out;
  for (int i = 0; i < n_ops;) {
    const int n = group_run(ops, n_ops, i);
    if (ops[i].group <= 0) {
      ++i;
      continue;
    }
    if (ws_group_selected(ops + i, n, batch)) {       // conv_ws.hip takes its arguments by value
      memset(out, 0, sizeof(ConvArgs) * n);
      out += n;
      i += n;
      continue;
    }
    COMIC_REQUIRE(ops[i].tile != COMIC_WS_TILE, "conv: group is not eligible for the weight-stationary 1x1 kernel");
    const int tile = group_tile(ops + i, n, batch);
    for (int j = 0; j < n; ++j)
      COMIC_REQUIRE(tile == COMIC_CHAIN_TILE || !(ops[i + j].flags & COMIC_OP_CHAIN_LINK),
                    "conv: COMIC_OP_CHAIN_LINK ops depend on each other -- their group runs on COMIC_CHAIN_TILE only (got tile %d)", tile);
    if (tile == COMIC_IMG_TILE || tile == COMIC_CHAIN_TILE) {      // conv_img.hip takes its arguments by value, too
      memset(out, 0, sizeof(ConvArgs) * n);
      out += n;
      i += n;
      continue;
    }
    int blk = 0;
    for (int j = 0; j < n; ++j) {
      const comic_cnn_op* op = ops + i + j;
      const comic_conv_weight* wt = weights + op->weight;
      if (int rc = validate_grouped_conv(op, buf_channels[op->src], buf_channels[op->dst], wt, batch)) return rc;
      COMIC_REQUIRE(buffers[op->src] && buffers[op->dst], "grouped conv: null buffer");
      ConvArgs& a = out[j];
      fill_args(a, op, buffers[op->src], buf_channels[op->src], buffers[op->dst], buf_channels[op->dst], wt, batch);
      COMIC_REQUIRE(a.zero, "conv: zero page symbol not resolvable");
      a.member_kind = op->kind == 7 ? 1 : 0;
      a.blk0 = blk;
      a.remap = 0;   // members differ in K: contiguous per-XCD ranges would put the heavy ones on few XCDs
      int lds = 0;
      const long nb = member_blocks(tile, a, &lds);
      COMIC_REQUIRE(nb >= 0, "grouped conv: member %d (stride %d, Cin %d) is not eligible for patch tile %d", j, op->SH,
                    op->Cin, tile);
      blk += (int)nb;
    }
    COMIC_REQUIRE(!is_walk_tile(tile) || shared_input_group(out, n),
                  "conv: tile %d walks over members that share their im2col matrix; this launch's do not", tile);
    if (is_im2col_tile(tile) && shared_input_group(out, n)) {
      int nt = 0;
      for (int j = 0; j < n; ++j) {
        out[j].remap = is_walk_tile(tile) ? 3 : 2;     // 3: one workgroup per pixel tile walks all nt out-channel tiles
        out[j].blk0 = nt;
        nt += cdiv(out[j].Cout, tile_bn(tile));
      }
      for (int j = 0; j < n; ++j) out[j].grp_nt = nt;
    }
    out += n;
    i += n;
  }
  return 0;
}

extern "C" int comic_conv2d_bn_relu(const comic_cnn_op* op, const void* x, int x_channels, void* y, int y_channels,
                                    const comic_conv_weight* wt, int batch, int dtype, void* stream) {
  COMIC_REQUIRE(op, "null op");
  hipStream_t st = (hipStream_t)stream;
  if (op->kind == 0 && dtype == COMIC_BF16 &&
      ((op->flags & COMIC_OP_POOLED_SRC) || op->tile == COMIC_WS_TILE || (op->tile == 0 && ws_group_selected(op, 1, batch)))) {
    // a single 1x1 conv on the weight-stationary kernel (conv_ws.hip): one-member group over a two-entry buffer table
    comic_cnn_op o = *op;
    o.src = 0; o.dst = 1; o.weight = 0;
    void* const bufs[2] = {(void*)x, y};
    const int32_t chans[2] = {x_channels, y_channels};
    COMIC_REQUIRE(wt, "conv: missing weights");
    return run_ws_group(&o, 1, bufs, chans, wt, batch, st);
  }
  COMIC_REQUIRE(!(op->kind == 0 && (op->flags & COMIC_OP_POOLED_SRC)), "conv: COMIC_OP_POOLED_SRC needs a bf16 plan");
  if (dtype == COMIC_BF16 && op->src_f32) {
    COMIC_REQUIRE(op->kind == 4 || op->kind == 7, "src_f32 is only supported by the global average pool and pool+bn");
    if (op->kind == 7) return run_op<bf16_t>(op, x, x_channels, y, y_channels, wt, batch, st);
    return run_op<float>(op, x, x_channels, y, y_channels, wt, batch, st);
  }
  if (dtype == COMIC_BF16) return run_op<bf16_t>(op, x, x_channels, y, y_channels, wt, batch, st);
  if (dtype == COMIC_F32) return run_op<float>(op, x, x_channels, y, y_channels, wt, batch, st);
  COMIC_REQUIRE(false, "unknown dtype %d", dtype);
  return 2;
}

namespace {
// Branch lanes: three internal non-blocking streams + fork/join events per device, created on
// first use and kept for the life of the process (the only objects this library owns).
struct Lanes {
  hipStream_t s[3];
  hipEvent_t fork_ev, join_ev[3];
  bool ok = false;
};
Lanes* get_lanes() {
  static Lanes lanes[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  Lanes& l = lanes[dev];
  if (!l.ok) {
    for (int i = 0; i < 3; ++i) {
      if (hipStreamCreateWithFlags(&l.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
      if (hipEventCreateWithFlags(&l.join_ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&l.fork_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
    l.ok = true;
  }
  return &l;
}
}  // namespace

static int cnn_forward_impl(const comic_cnn_op* ops, int n_ops, void* const* buffers, const int32_t* buf_channels,
                            const comic_conv_weight* weights, int batch, int dtype, const void* group_args_dev,
                            void* stream) {
  COMIC_REQUIRE(ops && buffers && buf_channels, "comic_cnn_forward: null table");
  const ConvArgs* gargs = (const ConvArgs*)group_args_dev;
  hipStream_t main_st = (hipStream_t)stream;
  Lanes* lanes = nullptr;
  bool used[3] = {false, false, false};
  for (int i = 0; i < n_ops; ++i) {
    const comic_cnn_op* op = ops + i;
    if (op->kind == 5) {  // fork
      if (!lanes) lanes = get_lanes();
      COMIC_REQUIRE(lanes, "comic_cnn_forward: cannot create branch streams");
      COMIC_REQUIRE(hipEventRecord(lanes->fork_ev, main_st) == hipSuccess, "fork: event record failed");
      for (int l = 0; l < 3; ++l) {
        COMIC_REQUIRE(hipStreamWaitEvent(lanes->s[l], lanes->fork_ev, 0) == hipSuccess, "fork: wait failed");
        used[l] = false;
      }
      continue;
    }
    if (op->kind == 6) {  // join
      COMIC_REQUIRE(lanes, "join without fork");
      for (int l = 0; l < 3; ++l) {
        if (!used[l]) continue;
        COMIC_REQUIRE(hipEventRecord(lanes->join_ev[l], lanes->s[l]) == hipSuccess, "join: event record failed");
        COMIC_REQUIRE(hipStreamWaitEvent(main_st, lanes->join_ev[l], 0) == hipSuccess, "join: wait failed");
      }
      continue;
    }
    if (gargs && op->group > 0) {
      COMIC_REQUIRE(dtype == COMIC_BF16 && (op->kind == 0 || op->kind == 7) && op->lane == 0,
                    "grouped launch needs a bf16 plan and conv / pool+bn ops on the caller's stream");
      const int n = group_run(ops, n_ops, i);
      if (ws_group_selected(op, n, batch)) {
        if (int rc = run_ws_group(op, n, buffers, buf_channels, weights, batch, main_st)) return rc;
        gargs += n;
        i += n - 1;
        continue;
      }
      COMIC_REQUIRE(op->tile != COMIC_WS_TILE, "conv: group is not eligible for the weight-stationary 1x1 kernel");
      const int tile = group_tile(op, n, batch);
      for (int j = 0; j < n; ++j)
        COMIC_REQUIRE(tile == COMIC_CHAIN_TILE || !(op[j].flags & COMIC_OP_CHAIN_LINK),
                      "conv: COMIC_OP_CHAIN_LINK ops depend on each other -- their group runs on COMIC_CHAIN_TILE only (got tile %d)", tile);
      if (tile == COMIC_CHAIN_TILE) {
        if (int rc = launch_img_chains(op, n, buffers, buf_channels, weights, batch, main_st)) return rc;
        COMIC_LAUNCH_CHECK("image-resident conv chains");
        gargs += n;
        i += n - 1;
        continue;
      }
      if (tile == COMIC_IMG_TILE) {
        // members of one shape share a launch; a member of another shape (7x1 128 -> 192 beside 1x7 128 -> 128) gets its own
        COMIC_REQUIRE(n <= kImgMaxMembers, "grouped launch: too many members for the image-resident kernel");
        ConvArgs ma[kImgMaxMembers];
        bool done[kImgMaxMembers] = {false, false, false, false};
        for (int j = 0; j < n; ++j) {
          COMIC_REQUIRE(op[j].kind == 0, "grouped launch: the image-resident kernel takes convolutions only");
          if (int rc = validate_grouped_conv(op + j, buf_channels[op[j].src], buf_channels[op[j].dst], weights + op[j].weight, batch)) return rc;
          fill_args(ma[j], op + j, buffers[op[j].src], buf_channels[op[j].src], buffers[op[j].dst], buf_channels[op[j].dst],
                    weights + op[j].weight, batch);
        }
        for (int j = 0; j < n; ++j) {
          if (done[j]) continue;
          ConvArgs run[kImgMaxMembers];
          int nr = 0;
          for (int q = j; q < n; ++q)
            if (!done[q] && img_same_shape(ma[q], ma[j])) {
              run[nr++] = ma[q];
              done[q] = true;
            }
          if (int rc = launch_img_convs(run, nr, main_st)) return rc;
        }
        COMIC_LAUNCH_CHECK("image-resident conv group");
        gargs += n;
        i += n - 1;
        continue;
      }
      long blocks = 0;
      int lds_max = 0;
      for (int j = 0; j < n; ++j) {
        ConvArgs a;
        fill_args(a, op + j, buffers[op[j].src], buf_channels[op[j].src], buffers[op[j].dst], buf_channels[op[j].dst],
                  weights + op[j].weight, batch);
        a.member_kind = op[j].kind == 7 ? 1 : 0;
        int lds = 0;
        const long nb = member_blocks(tile, a, &lds);
        COMIC_REQUIRE(nb >= 0, "grouped launch: member %d is not eligible for patch tile %d", j, tile);
        blocks += nb;
        lds_max = std::max(lds_max, lds);
      }
      COMIC_REQUIRE(blocks > 0 && blocks < (1L << 31), "grouped launch: bad workgroup count");
      if (is_walk_tile(tile)) blocks = (long)cdiv(batch * op->Ho * op->Wo, tile_bm(tile)) * walk_split(tile);   // one (or two) workgroups per pixel tile
      if (!is_im2col_tile(tile)) {
        if (int rc = launch_patch_grouped_tile(tile, gargs, n, (int)blocks, std::max(lds_max, op->min_lds), main_st)) return rc;
      } else if (int rc = launch_dma_grouped_tile(tile, gargs, n, (int)blocks, op->min_lds, main_st)) {
        return rc;
      }
      COMIC_LAUNCH_CHECK("grouped conv");
      gargs += n;
      i += n - 1;
      continue;
    }
    hipStream_t st = main_st;
    if (op->lane > 0) {
      COMIC_REQUIRE(lanes && op->lane <= 3, "branch lane %d outside a fork/join region", op->lane);
      st = lanes->s[op->lane - 1];
      used[op->lane - 1] = true;
    }
    if (op->kind == 8 || op->kind == 9) {
      // streaming Conv2d_2a -> Conv2d_2b -> MaxPool_3a (conv_stem.hip); weights[op->weight], [op->weight + 1].
      // kind 9: Conv2d_1a_3x3 inside the same pass -- src is the fp32 image (H, W = the image), the weight records are
      // Conv2d_1a (stem layout), Conv2d_2a, Conv2d_2b
      const bool with_1a = op->kind == 9;
      const int H0 = with_1a ? (op->H - 3) / 2 + 1 : op->H, W0 = with_1a ? (op->W - 3) / 2 + 1 : op->W;
      COMIC_REQUIRE(dtype == COMIC_BF16 && op->lane == 0, "stem stream: bf16 plans on the caller's stream only");
      COMIC_REQUIRE(op->Cin == (with_1a ? 3 : 32) && op->Cout == 64 && op->KH == 3 && op->KW == 3 && op->SH == 1 && op->SW == 1,
                    "stem stream: expects the (3 ->) 32 -> 32 -> 64 3x3 stem (got Cin %d, Cout %d)", op->Cin, op->Cout);
      COMIC_REQUIRE(comic_stem_stream_supported(H0, W0) && (!with_1a || comic_stem_stream_1a_supported(op->H, op->W)),
                    "stem stream: map %dx%d is not supported", op->H, op->W);
      const int Hp = (H0 - 2 - 3) / 2 + 1, Wp = (W0 - 2 - 3) / 2 + 1;
      COMIC_REQUIRE(op->Ho == Hp && op->Wo == Wp, "stem stream: Ho/Wo must be the pooled grid %dx%d", Hp, Wp);
      const comic_conv_weight* w0 = with_1a ? weights + op->weight : nullptr;
      const comic_conv_weight* wa = weights + op->weight + (with_1a ? 1 : 0);
      const comic_conv_weight* wb = wa + 1;
      const int xc = buf_channels[op->src], yc = buf_channels[op->dst];
      COMIC_REQUIRE(buffers[op->src] && buffers[op->dst] && wa->w && wb->w && wa->scale && wa->shift && wb->scale && wb->shift,
                    "stem stream: null buffer / weights");
      COMIC_REQUIRE(op->dst_coff + 64 <= yc && yc % 4 == 0 && op->dst_coff % 4 == 0, "stem stream: bad destination slice");
      COMIC_REQUIRE((long)batch * Hp * Wp * yc * 2 < (1L << 31), "stem stream: tensor too large");
      ComicStemArgs sa{};
      if (with_1a) {
        COMIC_REQUIRE(w0->w && w0->scale && w0->shift && xc == 3 && op->src_coff == 0, "stem stream: Conv2d_1a needs the whole fp32 image");
        sa.img = (const float*)buffers[op->src]; sa.Hi = op->H; sa.Wi = op->W;
        sa.w0 = (const float*)w0->w; sa.sc0 = w0->scale; sa.sh0 = w0->shift;
        sa.x = nullptr; sa.x_cs = 32; sa.x_co = 0;
      } else {
        COMIC_REQUIRE(op->src_coff + 32 <= xc && xc % 8 == 0 && op->src_coff % 8 == 0, "stem stream: bad source slice");
        COMIC_REQUIRE((long)batch * op->H * op->W * xc * 2 < (1L << 31), "stem stream: tensor too large");
        sa.x = (const bf16_t*)buffers[op->src]; sa.x_cs = xc; sa.x_co = op->src_coff;
      }
      sa.B = batch; sa.H0 = H0; sa.W0 = W0;
      sa.w1 = (const bf16_t*)wa->w; sa.w2 = (const bf16_t*)wb->w; sa.Kpad = (9 * 32 + 63) / 64 * 64;
      sa.sc1 = wa->scale; sa.sh1 = wa->shift; sa.sc2 = wb->scale; sa.sh2 = wb->shift;
      sa.y = (bf16_t*)buffers[op->dst]; sa.y_cs = yc; sa.y_co = op->dst_coff; sa.Hp = Hp; sa.Wp = Wp;
      sa.n_tasks = 2 * batch;
      if (int rc = comic_stem_stream_launch(sa, main_st)) return rc;
      COMIC_LAUNCH_CHECK("stem stream");
      continue;
    }
    const comic_conv_weight* wt = (op->kind <= 1 || op->kind == 7) ? weights + op->weight : nullptr;
    int rc = comic_conv2d_bn_relu(op, buffers[op->src], buf_channels[op->src], buffers[op->dst],
                                  buf_channels[op->dst], wt, batch, dtype, (void*)st);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int comic_cnn_forward(const comic_cnn_op* ops, int n_ops, void* const* buffers, const int32_t* buf_channels,
                                 const comic_conv_weight* weights, int batch, int dtype, void* stream) {
  return cnn_forward_impl(ops, n_ops, buffers, buf_channels, weights, batch, dtype, nullptr, stream);
}

extern "C" int comic_cnn_forward_grouped(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                                         const int32_t* buf_channels, const comic_conv_weight* weights, int batch,
                                         int dtype, const void* group_args_dev, void* stream) {
  COMIC_REQUIRE(group_args_dev || comic_cnn_group_args_bytes(ops, n_ops) == 0,
                "comic_cnn_forward_grouped: plan has grouped ops but no argument records");
  return cnn_forward_impl(ops, n_ops, buffers, buf_channels, weights, batch, dtype, group_args_dev, stream);
}

extern "C" int comic_pack_conv_weights(const float* w_hwio, void* w_packed, int kh, int kw, int cin, int cout,
                                       int dtype, void* stream) {
  const int K = kh * kw * cin, Kpad = (K + 63) / 64 * 64;
  const long total = (long)cout * Kpad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == COMIC_BF16)
    hipLaunchKernelGGL((pack_conv_weights_kernel<bf16_t>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st,
                       w_hwio, (bf16_t*)w_packed, K, Kpad, cout);
  else
    hipLaunchKernelGGL((pack_conv_weights_kernel<float>), dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st,
                       w_hwio, (float*)w_packed, K, Kpad, cout);
  COMIC_LAUNCH_CHECK("pack_conv_weights");
  return 0;
}

extern "C" int comic_fold_bn(const float* beta, const float* mean, const float* var, float eps, float* scale,
                             float* shift, int c, void* stream) {
  hipLaunchKernelGGL(fold_bn_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, beta, mean, var, eps,
                     scale, shift, c);
  COMIC_LAUNCH_CHECK("fold_bn");
  return 0;
}


// =================================================================================================
// Backward of the plan (train_mode cnn_finetune, src/train.py:241-249): freeze_scopes == '' puts the
// conv weights and BN betas into the trainable set (src/model_base.py:834-849) while the graph stays
// in BN-inference mode (model_base.py:76), so per conv
//     z = conv(x, w) * scale + (beta - mean * scale),  y = relu(z)
//     dz = 1[y > 0] * dy ;  d beta = sum_pixels dz ;  d conv = dz * scale
//     d w = x^T (*) d conv (backward-weight) ;  d x += w^T (*) d conv (backward-data)
// Backward-data runs the FORWARD kernels on `d conv` with the flipped / transposed filter
// (stride-2 convs: `d conv` is written zero-dilated), accumulating into the gradient buffer.
// Backward-weight is an implicit GEMM whose reduction runs over output pixels; operand tiles are
// transposed on their way into LDS so the MFMA fragments are the same 16-byte row reads as in the
// forward kernel; pixel ranges are split over workgroups and combined with fp32 atomics.
// =================================================================================================
namespace {

template <typename T>
__device__ __forceinline__ void load_chunk_f(const void* base, size_t elem_off, bool f32, float* v) {
  constexpr int EPC = Elem<T>::EPC;
  if (sizeof(T) == 4 || f32) {
    const float* p = (const float*)base + elem_off;
#pragma unroll
    for (int i = 0; i < EPC; i += 4) {
      const float4 t = *(const float4*)(p + i);
      v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
  } else {
    load_vec<T>((const T*)base + elem_off, v);
  }
}
template <typename T>
__device__ __forceinline__ void store_chunk_f(void* base, size_t elem_off, bool f32, const float* v) {
  constexpr int EPC = Elem<T>::EPC;
  if (sizeof(T) == 4 || f32) {
    float* p = (float*)base + elem_off;
#pragma unroll
    for (int i = 0; i < EPC; i += 4) *(float4*)(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  } else {
    store_vec<T>((T*)base + elem_off, v);
  }
}

struct ActGradArgs {
  const void* y;    // forward output (post-ReLU), channel slice [yco, yco + C) of rows of ycs channels
  const void* dy;   // its gradient, same geometry
  int ycs, yco, yf32;
  const float* scale;
  void* dz;         // [B][Hd][Wd][C] plan dtype; output pixel (ho, wo) lands at (ho*dil, wo*dil)
  int B, Ho, Wo, C, Hd, Wd, dil;
};

// grid (pixel chunks, ceil(C/64)); 256 threads = (64/EPC channel chunks) x (pixel lanes).  Besides dz
// the workgroup reduces its share of d beta[c] = sum_pixels 1[y>0] dy (unscaled) in LDS and adds
// it to the fp32 accumulator with one atomic per channel.
template <typename T>
__global__ __launch_bounds__(256) void act_grad_kernel(ActGradArgs a, float* __restrict__ dbeta, long px_per_block) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CPB = 64 / EPC;        // channel chunks per workgroup
  constexpr int PL = 256 / CPB;        // pixel lanes
  __shared__ float red[PL][64 + 1];
  const int tid = threadIdx.x;
  const int cc = tid % CPB, pl = tid / CPB;
  const int c0 = blockIdx.y * 64 + cc * EPC;
  const bool cok = c0 < a.C;           // C % EPC == 0
  const long P = (long)a.B * a.Ho * a.Wo;
  const long p0 = (long)blockIdx.x * px_per_block, p1 = min(P, p0 + px_per_block);
  float sc[EPC], sum[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    sc[i] = cok ? a.scale[c0 + i] : 0.f;
    sum[i] = 0.f;
  }
  if (cok) {
    for (long p = p0 + pl; p < p1; p += PL) {
      const int wo = (int)(p % a.Wo);
      const long q = p / a.Wo;
      const int ho = (int)(q % a.Ho), b = (int)(q / a.Ho);
      float yv[EPC], gv[EPC];
      const size_t src = (size_t)p * a.ycs + a.yco + c0;
      load_chunk_f<T>(a.y, src, a.yf32 != 0, yv);
      load_chunk_f<T>(a.dy, src, a.yf32 != 0, gv);
#pragma unroll
      for (int i = 0; i < EPC; ++i) {
        const float g = yv[i] > 0.f ? gv[i] : 0.f;
        sum[i] += g;
        gv[i] = g * sc[i];
      }
      const size_t dst = (((size_t)b * a.Hd + ho * a.dil) * a.Wd + wo * a.dil) * a.C + c0;
      store_chunk_f<T>(a.dz, dst, false, gv);
    }
  }
#pragma unroll
  for (int i = 0; i < EPC; ++i) red[pl][cc * EPC + i] = sum[i];
  __syncthreads();
  if (tid < 64) {
    const int c = blockIdx.y * 64 + tid;
    if (c < a.C) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < PL; ++r) s += red[r][tid];
      atomicAdd(dbeta + c, s);
    }
  }
}

// The same for COMIC_OP_X3 plans (bf16x3 backward): y is hi + lo of two channel regions (or fp32), dy an fp32 gradient
// buffer of the LOGICAL channels, and dz leaves as the three regions [hi | lo | hi] of C channels each -- the operand
// layout of the x3 convolutions, so that the backward-data conv of dz against [Wt_hi | Wt_hi | Wt_lo] is the split product
// hi*hi + lo*hi + hi*lo with fp32 accumulation.
struct ActGradX3Args {
  const void* y;    // forward output: bf16 regions (hi at yco, lo y_lo channels behind) of rows of ycs channels, or fp32 (y_lo = 0)
  const float* dy;  // its gradient: fp32 rows of gcs channels, slice from yco
  int ycs, yco, y_lo, gcs;
  const float* scale;
  bf16_t* dz;       // [B][Hd][Wd][3 C]
  int B, Ho, Wo, C, Hd, Wd, dil;
};
__global__ __launch_bounds__(256) void act_grad_x3_kernel(ActGradX3Args a, float* __restrict__ dbeta, long px_per_block) {
  constexpr int EPC = 8, CPB = 64 / EPC, PL = 256 / CPB;
  __shared__ float red[PL][64 + 1];
  const int tid = threadIdx.x;
  const int cc = tid % CPB, pl = tid / CPB;
  const int c0 = blockIdx.y * 64 + cc * EPC;
  const bool cok = c0 < a.C;           // C % 8 == 0
  const long P = (long)a.B * a.Ho * a.Wo;
  const long p0 = (long)blockIdx.x * px_per_block, p1 = min(P, p0 + px_per_block);
  float sc[EPC], sum[EPC];
#pragma unroll
  for (int i = 0; i < EPC; ++i) {
    sc[i] = cok ? a.scale[c0 + i] : 0.f;
    sum[i] = 0.f;
  }
  if (cok) {
    for (long p = p0 + pl; p < p1; p += PL) {
      const int wo = (int)(p % a.Wo);
      const long q = p / a.Wo;
      const int ho = (int)(q % a.Ho), b = (int)(q / a.Ho);
      float yv[EPC], gv[EPC];
      const size_t src = (size_t)p * a.ycs + a.yco + c0;
      if (a.y_lo) {
        float yl[EPC];
        load_vec<bf16_t>((const bf16_t*)a.y + src, yv);
        load_vec<bf16_t>((const bf16_t*)a.y + src + a.y_lo, yl);
#pragma unroll
        for (int i = 0; i < EPC; ++i) yv[i] += yl[i];
      } else {
        load_chunk_f<bf16_t>(a.y, src, true, yv);
      }
      load_chunk_f<bf16_t>(a.dy, (size_t)p * a.gcs + a.yco + c0, true, gv);
      float hi[EPC], lo[EPC];
#pragma unroll
      for (int i = 0; i < EPC; ++i) {
        const float g = yv[i] > 0.f ? gv[i] : 0.f;
        sum[i] += g;
        const float z = g * sc[i];
        hi[i] = bf16_to_f32(f32_to_bf16(z));
        lo[i] = z - hi[i];
      }
      bf16_t* dst = a.dz + (((size_t)b * a.Hd + ho * a.dil) * a.Wd + wo * a.dil) * 3 * a.C + c0;
      store_vec<bf16_t>(dst, hi);
      store_vec<bf16_t>(dst + a.C, lo);
      store_vec<bf16_t>(dst + 2 * a.C, hi);
    }
  }
#pragma unroll
  for (int i = 0; i < EPC; ++i) red[pl][cc * EPC + i] = sum[i];
  __syncthreads();
  if (tid < 64) {
    const int c = blockIdx.y * 64 + tid;
    if (c < a.C) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < PL; ++r) s += red[r][tid];
      atomicAdd(dbeta + c, s);
    }
  }
}

// master [Cout][Kpad] (fp32, k = (kh*KW + kw)*Cin + ci) -> x3 backward-data filter [Cin][Kpad2], k2 = tap2 * 3 Cout + region * Cout +
// co with the flipped taps of pack_bwd_weights_kernel and the regions [W_hi | W_hi | W_lo]
__device__ __forceinline__ bf16_t pack_bwd_x3_value(const float* __restrict__ master, int k2, int ci, int KH, int KW, int Cin,
                                                    int Cout, int Kpad) {
  if (k2 >= KH * KW * 3 * Cout) return f32_to_bf16(0.f);
  const int tap2 = k2 / (3 * Cout), rr = k2 - tap2 * 3 * Cout;
  const int region = rr / Cout, co = rr - region * Cout;
  const int kh = KH - 1 - tap2 / KW, kw = KW - 1 - tap2 % KW;
  const float v = master[(size_t)co * Kpad + (kh * KW + kw) * Cin + ci];
  const bf16_t hi = f32_to_bf16(v);
  return region < 2 ? hi : f32_to_bf16(v - bf16_to_f32(hi));
}
__global__ void pack_bwd_weights_x3_kernel(const float* __restrict__ master, bf16_t* __restrict__ out, int KH, int KW, int Cin,
                                           int Cout, int Kpad, int Kpad2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cin * Kpad2) return;
  out[idx] = pack_bwd_x3_value(master, (int)(idx % Kpad2), (int)(idx / Kpad2), KH, KW, Cin, Cout, Kpad);
}

// master [Cout][Kpad] (k = (kh*KW + kw)*Cin + ci) -> backward-data filter [Cin][Kpad2],
// k2 = ((KH-1-kh)*KW + (KW-1-kw))*Cout + co
template <typename T>
__global__ void pack_bwd_weights_kernel(const float* __restrict__ master, T* __restrict__ out, int KH, int KW, int Cin,
                                        int Cout, int Kpad, int Kpad2) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Cin * Kpad2) return;
  const int k2 = (int)(idx % Kpad2), ci = (int)(idx / Kpad2);
  float v = 0.f;
  if (k2 < KH * KW * Cout) {
    const int tap2 = k2 / Cout, co = k2 % Cout;
    const int kh = KH - 1 - tap2 / KW, kw = KW - 1 - tap2 % KW;
    v = master[(size_t)co * Kpad + (kh * KW + kw) * Cin + ci];
  }
  if (sizeof(T) == 4)
    ((float*)out)[idx] = v;
  else
    ((bf16_t*)out)[idx] = f32_to_bf16(v);
}

struct WgradArgs {
  const void* dz;   // [B][Hd][Wd][dz_cs] plan dtype, channels [dz_co, dz_co + Cout) (plain plans: dz_cs = Cout, dz_co = 0)
  int Hd, Wd, dil, dz_cs, dz_co;
  int x_part;       // STEM, bf16: 0 the image rounded to bf16, 1 its rounding residue (the lo half of an x3 backward)
  int parts;        // 3: the x3 backward's three products in ONE launch -- blockIdx.z = split * 3 + part, part 0 (x hi, dz hi),
                    // 1 (x lo: x_lo_off channels further / the stem's residue, dz hi), 2 (x hi, dz lo: z_lo_off channels further)
  int x_lo_off, z_lo_off;
  const void* x;    // forward input (plan dtype; STEM: fp32 image)
  int x_cs, x_co;
  float* dw;        // [Cout][Kpad] fp32 (STEM: [K][Cout]), accumulated with atomics
  int B, H, W, Cin, Cout, KH, KW, SH, SW, PT, PL, Ho, Wo, K, Kpad;
  long P, p_per_split;
};

// grid (Kpad/64, ceil(Cout/64), splits): D[co][kk] += sum_{p in slice} dz[p][co] * xcol[p][kk]
template <typename T, bool STEM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 4 * EPC;          // pixels per k-step (64 bytes of k per LDS row)
  constexpr int CPP = 64 / EPC;         // 16-byte chunks per pixel in a 64-channel tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * 64 * kRowBytes];
  unsigned char* Zs = smem;                        // [2][64 co][kRowBytes]
  unsigned char* Xs = smem + 2 * 64 * kRowBytes;   // [2][64 kk][kRowBytes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kk0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  int bz = blockIdx.z;
  if (a.parts == 3) {
    const int part = bz % 3;
    bz /= 3;
    a.x_co += part == 1 ? a.x_lo_off : 0;
    a.x_part = part == 1;
    a.dz_co += part == 2 ? a.z_lo_off : 0;
  }
  const long p_begin = (long)bz * a.p_per_split, p_end = min(a.P, p_begin + a.p_per_split);
  const int px_l = tid / CPP, cg = tid % CPP;
  // this thread's fixed channel chunk of each tile
  const int co_c = co0 + cg * EPC;
  const bool co_ok = co_c < a.Cout;           // Cout % EPC == 0
  const int kk_c = kk0 + cg * EPC;
  int kh = 0, kw = 0, ci = 0;
  bool kk_ok = kk_c < a.K;
  if (!STEM) {
    if (kk_ok) {
      const int tap = kk_c / a.Cin;
      ci = kk_c % a.Cin;
      kh = tap / a.KW;
      kw = tap % a.KW;
    }
  }
  const int hw = a.Ho * a.Wo;
  uint4 zreg, xreg;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  auto load_tile = [&](long pbase) {
    const long p = pbase + px_l;
    zreg = xreg = zero4;
    if (p >= p_end) return;
    const int b = (int)(p / hw);
    const int rem = (int)(p - (long)b * hw);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    if (co_ok)
      zreg = *(const uint4*)((const T*)a.dz + (((size_t)b * a.Hd + ho * a.dil) * a.Wd + wo * a.dil) * a.dz_cs + a.dz_co + co_c);
    if (STEM) {
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int kk = kk_c + e;
        v[e] = 0.f;
        if (kk < a.K) {
          const int tap = kk / a.Cin, c = kk - tap * a.Cin;
          const int hi = ho * a.SH - a.PT + tap / a.KW, wi = wo * a.SW - a.PL + tap % a.KW;
          if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
            v[e] = ((const float*)a.x)[((size_t)(b * a.H + hi) * a.W + wi) * a.x_cs + a.x_co + c];
        }
      }
      if (sizeof(T) == 4) {
        xreg = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
      } else {
        if (a.x_part) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] -= bf16_to_f32(f32_to_bf16(v[e]));
        }
        xreg = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4 % EPC], v[5 % EPC]),
                          pack_bf16x2(v[6 % EPC], v[7 % EPC]));
      }
    } else if (kk_ok) {
      const int hi = ho * a.SH - a.PT + kh, wi = wo * a.SW - a.PL + kw;
      if ((unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
        xreg = *(const uint4*)((const T*)a.x + ((size_t)(b * a.H + hi) * a.W + wi) * a.x_cs + a.x_co + ci);
    }
  };
  // transposed store: element e of the chunk goes to row (cg*EPC + e), k position px_l
  auto store_tile = [&](int buf) {
    unsigned char* zb = Zs + buf * 64 * kRowBytes;
    unsigned char* xb = Xs + buf * 64 * kRowBytes;
    if (sizeof(T) == 4) {
      const uint32_t zu[4] = {zreg.x, zreg.y, zreg.z, zreg.w}, xu[4] = {xreg.x, xreg.y, xreg.z, xreg.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *(uint32_t*)(zb + (cg * 4 + e) * kRowBytes + px_l * 4) = zu[e];
        *(uint32_t*)(xb + (cg * 4 + e) * kRowBytes + px_l * 4) = xu[e];
      }
    } else {
      const uint32_t zu[4] = {zreg.x, zreg.y, zreg.z, zreg.w}, xu[4] = {xreg.x, xreg.y, xreg.z, xreg.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint16_t zv = (uint16_t)(e & 1 ? zu[e >> 1] >> 16 : zu[e >> 1] & 0xFFFFu);
        const uint16_t xv = (uint16_t)(e & 1 ? xu[e >> 1] >> 16 : xu[e >> 1] & 0xFFFFu);
        *(uint16_t*)(zb + (cg * 8 + e) * kRowBytes + px_l * 2) = zv;
        *(uint16_t*)(xb + (cg * 8 + e) * kRowBytes + px_l * 2) = xv;
      }
    }
  };

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchunk = lane >> 4;
  const int zoff = (wn * 32 + frow) * kRowBytes + fchunk * 16;
  const int xoff = (wm * 32 + frow) * kRowBytes + fchunk * 16;

  if (p_begin < p_end) {
    load_tile(p_begin);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (long pb = p_begin; pb < p_end; pb += BKE, buf ^= 1) {
      const bool more = pb + BKE < p_end;
      if (more) load_tile(pb + BKE);
      uint4 zf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) zf[i] = *(const uint4*)(Zs + buf * 64 * kRowBytes + zoff + i * 16 * kRowBytes);
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[j] = *(const uint4*)(Xs + buf * 64 * kRowBytes + xoff + j * 16 * kRowBytes);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[i]),
                                                                __builtin_bit_cast(bf16x8_t, xf[j]), acc[i][j], 0, 0, 0);
          } else {
            const f32x4_t zv = __builtin_bit_cast(f32x4_t, zf[i]);
            const f32x4_t xv = __builtin_bit_cast(f32x4_t, xf[j]);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[0], xv[0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[1], xv[1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[2], xv[2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[3], xv[3], acc[i][j], 0, 0, 0);
          }
        }
      if (more) store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  // D[co][kk]: lane holds co = .. + 4*(lane>>4) + reg, kk = .. + (lane & 15)
  const int kcol = lane & 15, cq = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = kk0 + wm * 32 + j * 16 + kcol;
      if (kk >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wn * 32 + i * 16 + cq + r;
        if (co >= a.Cout) continue;
        float* dst = STEM ? a.dw + (size_t)kk * a.Cout + co : a.dw + (size_t)co * a.Kpad + kk;
        atomicAdd(dst, acc[i][j][r]);
      }
    }
}

// bf16 fast path of the backward-weight product: operand tiles stay pixel-major in LDS (16-byte
// stores, no scalar transposition) and the MFMA fragments are gathered with the CDNA4 transposing
// read ds_read_b64_tr_b16 (4 pixel rows x 16 channels per 16-lane group, delivered channel-major).
// k index <-> pixel mapping of a 32-pixel step: k = 8g + e  <->  pixel 4*(g + 4*(e >> 2)) + (e & 3), the
// same for both operands, so that the two 16-lane groups of a 32-lane half touch 8 consecutive pixel
// rows; with a row stride of 32*odd bytes those are 64 distinct banks (conflict-free).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

template <int BM, int BN>   // BM output channels x BN filter taps*channels per workgroup
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(WgradArgs a) {
  constexpr int ZS = BM * 2 + 32, XS = BN * 2 + 32;       // LDS row strides in bytes (32 * odd)
  constexpr int ZCH = BM / 64, XCH = BN / 64;             // 16-byte chunks per thread per step
  constexpr int TM = BM / 32, TN = BN / 32;               // 16x16 tiles per wave (2 x 2 waves)
  static_assert((ZS / 32) % 2 == 1 && (XS / 32) % 2 == 1, "bank-conflict-free row strides");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 32 * (ZS + XS)];
  unsigned char* Zs = smem;                    // [2][32 px][ZS]
  unsigned char* Xs = smem + 2 * 32 * ZS;      // [2][32 px][XS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kk0 = blockIdx.x * BN, co0 = blockIdx.y * BM;
  int bz = blockIdx.z;
  if (a.parts == 3) {
    const int part = bz % 3;
    bz /= 3;
    a.x_co += part == 1 ? a.x_lo_off : 0;
    a.dz_co += part == 2 ? a.z_lo_off : 0;
  }
  const long p_begin = (long)bz * a.p_per_split, p_end = min(a.P, p_begin + a.p_per_split);
  const int px_l = tid >> 3, cg = tid & 7;     // pixel of the step, 8-channel chunk inside 64 channels
  // fixed channel chunks of this thread
  int co_c[ZCH];
  bool co_ok[ZCH];
#pragma unroll
  for (int i = 0; i < ZCH; ++i) {
    co_c[i] = co0 + i * 64 + cg * 8;
    co_ok[i] = co_c[i] < a.Cout;
  }
  int kh[XCH], kw[XCH], ci[XCH];
  bool kk_ok[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int kk = kk0 + i * 64 + cg * 8;
    kk_ok[i] = kk < a.K;
    const int tap = kk_ok[i] ? kk / a.Cin : 0;
    ci[i] = kk_ok[i] ? kk % a.Cin : 0;
    kh[i] = tap / a.KW;
    kw[i] = tap % a.KW;
  }
  const int hw = a.Ho * a.Wo;
  uint4 zreg[ZCH], xreg[XCH];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  auto load_tile = [&](long pbase) {
    const long p = pbase + px_l;
#pragma unroll
    for (int i = 0; i < ZCH; ++i) zreg[i] = zero4;
#pragma unroll
    for (int i = 0; i < XCH; ++i) xreg[i] = zero4;
    if (p >= p_end) return;
    const int b = (int)(p / hw);
    const int rem = (int)(p - (long)b * hw);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const bf16_t* zp = (const bf16_t*)a.dz + (((size_t)b * a.Hd + ho * a.dil) * a.Wd + wo * a.dil) * a.dz_cs + a.dz_co;
#pragma unroll
    for (int i = 0; i < ZCH; ++i)
      if (co_ok[i]) zreg[i] = *(const uint4*)(zp + co_c[i]);
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int hi = ho * a.SH - a.PT + kh[i], wi = wo * a.SW - a.PL + kw[i];
      if (kk_ok[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
        xreg[i] = *(const uint4*)((const bf16_t*)a.x + ((size_t)(b * a.H + hi) * a.W + wi) * a.x_cs + a.x_co + ci[i]);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ZCH; ++i) *(uint4*)(Zs + (buf * 32 + px_l) * ZS + (i * 64 + cg * 8) * 2) = zreg[i];
#pragma unroll
    for (int i = 0; i < XCH; ++i) *(uint4*)(Xs + (buf * 32 + px_l) * XS + (i * 64 + cg * 8) * 2) = xreg[i];
  };
  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // transposing-read lane addressing: 16-lane group g, lane j = 4q + p supplies row q, columns 4p..4p+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const uint32_t zs0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)Zs);
  const uint32_t xs0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)Xs);
  const uint32_t zlane = (4 * g + q) * ZS + (wn * (BM / 2) + 4 * pp) * 2;   // + 16*ZS for the second half
  const uint32_t xlane = (4 * g + q) * XS + (wm * (BN / 2) + 4 * pp) * 2;
  auto tr_read = [&](uint32_t addr) -> s16x4_t {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)addr);
  };

  if (p_begin < p_end) {
    load_tile(p_begin);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    for (long pb = p_begin; pb < p_end; pb += 32, buf ^= 1) {
      const bool more = pb + 32 < p_end;
      if (more) load_tile(pb + 32);
      const uint32_t zb = zs0 + buf * 32 * ZS + zlane, xb = xs0 + buf * 32 * XS + xlane;
      s16x8_t zf[TM], xf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const s16x4_t lo = tr_read(zb + i * 32), hi = tr_read(zb + i * 32 + 16 * ZS);
        zf[i] = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const s16x4_t lo = tr_read(xb + j * 32), hi = tr_read(xb + j * 32 + 16 * XS);
        xf[j] = (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, zf[i]),
                                                              __builtin_bit_cast(bf16x8_t, xf[j]), acc[i][j], 0, 0, 0);
      if (more) store_tile(buf ^ 1);
      __syncthreads();
    }
  }
  const int kcol = lane & 15, cq = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int kk = kk0 + wm * (BN / 2) + j * 16 + kcol;
      if (kk >= a.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wn * (BM / 2) + i * 16 + cq + r;
        if (co >= a.Cout) continue;
        atomicAdd(a.dw + (size_t)co * a.Kpad + kk, acc[i][j][r]);
      }
    }
}

struct PoolGradArgs {
  const void* x;    // forward input of the pool (max only)
  const void* dy;   // gradient of the pool output, slice [yco, yco + C) of ycs
  void* dx;         // gradient of the pool input, slice [xco, xco + C) of xcs  (accumulated)
  int xcs, xco, ycs, yco, dy_f32, dx_f32;
  int B, H, W, C, KH, KW, SH, SW, PT, PL, Ho, Wo;
  int dxcs, dxco;   // channel stride / offset of dx (plain plans: xcs, xco)
  int x_lo;         // x3 plans: x is hi + lo, the lo region x_lo channels behind the hi region (0: plain)
};

// MODE 0 max (first maximum in window scan order), 1 avg over valid taps, 2 plain mean over the
// KHxKW VALID window (the head's global pool).  Gather form: one thread per input pixel x chunk.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pool_grad_kernel(PoolGradArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  const int cvecs = a.C / EPC;
  const long total = (long)a.B * a.H * a.W * cvecs;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cv = (int)(idx % cvecs);
  const long p = idx / cvecs;
  const int wi = (int)(p % a.W);
  const long q = p / a.W;
  const int hi = (int)(q % a.H), b = (int)(q / a.H);
  float g[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) g[j] = 0.f;
  // windows (ho, wo) that contain (hi, wi): ho*SH - PT <= hi < ho*SH - PT + KH
  const int ho_lo = max(0, (hi + a.PT - a.KH + a.SH) / a.SH), ho_hi = min(a.Ho - 1, (hi + a.PT) / a.SH);
  const int wo_lo = max(0, (wi + a.PL - a.KW + a.SW) / a.SW), wo_hi = min(a.Wo - 1, (wi + a.PL) / a.SW);
  for (int ho = ho_lo; ho <= ho_hi; ++ho)
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      float dyv[EPC];
      load_chunk_f<T>(a.dy, ((size_t)(b * a.Ho + ho) * a.Wo + wo) * a.ycs + a.yco + cv * EPC, a.dy_f32 != 0, dyv);
      if (MODE == 0) {
        float best[EPC];
        int arg[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) { best[j] = -INFINITY; arg[j] = -1; }
        for (int kh = 0; kh < a.KH; ++kh) {
          const int h2 = ho * a.SH - a.PT + kh;
          if ((unsigned)h2 >= (unsigned)a.H) continue;
          for (int kw = 0; kw < a.KW; ++kw) {
            const int w2 = wo * a.SW - a.PL + kw;
            if ((unsigned)w2 >= (unsigned)a.W) continue;
            float v[EPC];
            const T* xp = (const T*)a.x + ((size_t)(b * a.H + h2) * a.W + w2) * a.xcs + a.xco + cv * EPC;
            load_vec<T>(xp, v);
            if (a.x_lo) {                          // (hi + lo is exact in fp32: the order pool_x3_kernel's pairs have)
              float vl[EPC];
              load_vec<T>(xp + a.x_lo, vl);
#pragma unroll
              for (int j = 0; j < EPC; ++j) v[j] += vl[j];
            }
#pragma unroll
            for (int j = 0; j < EPC; ++j)
              if (v[j] > best[j]) { best[j] = v[j]; arg[j] = h2 * a.W + w2; }
          }
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j)
          if (arg[j] == hi * a.W + wi) g[j] += dyv[j];
      } else if (MODE == 1) {
        const int h0 = max(0, ho * a.SH - a.PT), h1 = min(a.H, ho * a.SH - a.PT + a.KH);
        const int w0 = max(0, wo * a.SW - a.PL), w1 = min(a.W, wo * a.SW - a.PL + a.KW);
        const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
#pragma unroll
        for (int j = 0; j < EPC; ++j) g[j] += dyv[j] * inv;
      } else {
        const float inv = 1.0f / (float)(a.KH * a.KW);
#pragma unroll
        for (int j = 0; j < EPC; ++j) g[j] += dyv[j] * inv;
      }
    }
  const size_t off = (size_t)p * a.dxcs + a.dxco + cv * EPC;
  float old[EPC];
  load_chunk_f<T>(a.dx, off, a.dx_f32 != 0, old);
#pragma unroll
  for (int j = 0; j < EPC; ++j) old[j] += g[j];
  store_chunk_f<T>(a.dx, off, a.dx_f32 != 0, old);
}


// MaxPoolGrad of the 3x3 / stride-2 pools (MaxPool_3a / 5a and the pool branches of Mixed_6a / 7a; VALID or padded), tiled
// through the LDS.  The gather kernel above re-derives a window's arg-max once per input pixel it covers: up to four
// windows x nine 16-byte loads per thread -- 136 us per pool at 32 images (profiles/r05_finetune_lanes.txt: 6.5 % of the
// step's chain lane for two element-wise ops).  Here a workgroup owns a 16 x 16 tile of INPUT pixels and CG channel chunks:
//   1. the <= 19 x 19 input pixels its <= 9 x 9 windows read go to the LDS once (taps outside the image: -inf);
//   2. one thread per (window, chunk) finds the arg-max per channel -- FIRST maximum in (kh, kw) scan order, strict >, as
//      the gather kernel and TF's MaxPoolGrad -- and keeps (tap index per channel, the window's dy chunk) in the LDS;
//   3. one thread per (input pixel, chunk) adds the dy of its <= 4 windows whose arg-max it is, in (ho, wo) order, into dx.
// Same comparisons, same sums in the same order as pool_grad_kernel<T, 0>: identical bits.
constexpr int kMpgTile = 16, kMpgWin = kMpgTile / 2 + 1, kMpgX = 2 * kMpgWin + 1, kMpgCG = 4;
// X3 (COMIC_OP_X3 plans): x is hi + lo of two bf16 regions (a.x_lo apart), dy / dx fp32 buffers with their own strides.
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256) void maxpool_grad_s2_kernel(PoolGradArgs a, int tiles_y, int tiles_x, int cgroups) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ uint4 xs[kMpgCG][kMpgX * kMpgX];                           // raw 16-byte chunks of the input pixels
  __shared__ uint4 xs_lo[X3 ? kMpgCG : 1][X3 ? kMpgX * kMpgX : 1];      // ... of their lo region
  __shared__ __attribute__((aligned(16))) float dys[kMpgCG][kMpgWin * kMpgWin][EPC];
  __shared__ __attribute__((aligned(8))) unsigned char args[kMpgCG][kMpgWin * kMpgWin][8];
  int bid = blockIdx.x;
  const int cgi = bid % cgroups; bid /= cgroups;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int tid = threadIdx.x;
  const int cvecs = a.C / EPC;
  const int cv0 = cgi * kMpgCG, ncv = min(kMpgCG, cvecs - cv0);
  const int y0 = ty * kMpgTile, x0 = tx * kMpgTile;
  // windows that touch the tile: ho * 2 - PT <= y <= ho * 2 - PT + 2 for some y in [y0, y0 + 16)
  const int ho_lo = max(0, (y0 + a.PT - 1) >> 1), ho_hi = min(a.Ho - 1, (min(a.H, y0 + kMpgTile) - 1 + a.PT) >> 1);
  const int wo_lo = max(0, (x0 + a.PL - 1) >> 1), wo_hi = min(a.Wo - 1, (min(a.W, x0 + kMpgTile) - 1 + a.PL) >> 1);
  const int nho = ho_hi - ho_lo + 1, nwo = wo_hi - wo_lo + 1;        // <= kMpgWin each (may be <= 0 at a ragged edge)
  const int xr0 = 2 * ho_lo - a.PT, xc0 = 2 * wo_lo - a.PL;          // input pixel of LDS position (0, 0)
  const int nxr = 2 * nho + 1, nxc = 2 * nwo + 1;
  const bool any = nho > 0 && nwo > 0;
  const uint32_t ninf = sizeof(T) == 2 ? 0xFF80FF80u : 0xFF800000u;  // -inf in every element of a chunk
  if (any) {
    for (int i = tid; i < nxr * nxc * ncv; i += 256) {
      const int c = i % ncv, px = i / ncv;
      const int r = px / nxc, q = px - r * nxc;
      const int h = xr0 + r, w = xc0 + q;
      uint4 v = make_uint4(ninf, ninf, ninf, ninf), vl = make_uint4(0, 0, 0, 0);
      if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W) {
        const T* xp = (const T*)a.x + ((size_t)(b * a.H + h) * a.W + w) * a.xcs + a.xco + (cv0 + c) * EPC;
        v = *(const uint4*)xp;
        if constexpr (X3) vl = *(const uint4*)(xp + a.x_lo);
      }
      xs[c][r * kMpgX + q] = v;
      if constexpr (X3) xs_lo[c][r * kMpgX + q] = vl;
    }
  }
  __syncthreads();
  if (any) {
    for (int i = tid; i < nho * nwo * ncv; i += 256) {
      const int c = i % ncv, wdx = i / ncv;
      const int wr = wdx / nwo, wq = wdx - wr * nwo;
      float best[EPC];
      uint32_t arg[EPC];
#pragma unroll
      for (int j = 0; j < EPC; ++j) { best[j] = -INFINITY; arg[j] = 255u; }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float v[EPC];
          load_vec<T>((const T*)&xs[c][(2 * wr + kh) * kMpgX + 2 * wq + kw], v);
          if constexpr (X3) {
            float vl[EPC];
            load_vec<T>((const T*)&xs_lo[c][(2 * wr + kh) * kMpgX + 2 * wq + kw], vl);
#pragma unroll
            for (int j = 0; j < EPC; ++j) v[j] += vl[j];
          }
#pragma unroll
          for (int j = 0; j < EPC; ++j)
            if (v[j] > best[j]) { best[j] = v[j]; arg[j] = kh * 3 + kw; }
        }
      float dyv[EPC];
      load_chunk_f<T>(a.dy, ((size_t)(b * a.Ho + ho_lo + wr) * a.Wo + wo_lo + wq) * a.ycs + a.yco + (cv0 + c) * EPC, a.dy_f32 != 0, dyv);
      const int slot = wr * kMpgWin + wq;
#pragma unroll
      for (int j = 0; j < EPC; j += 4) *(float4*)&dys[c][slot][j] = make_float4(dyv[j], dyv[j + 1], dyv[j + 2], dyv[j + 3]);
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        if (j < 4) lo |= arg[j] << (8 * j); else hi |= arg[j] << (8 * (j - 4));
      }
      *(uint2*)args[c][slot] = make_uint2(lo, hi);
    }
  }
  __syncthreads();
  for (int i = tid; i < kMpgTile * kMpgTile * ncv; i += 256) {
    const int c = i % ncv, px = i / ncv;
    const int hi = y0 + px / kMpgTile, wi = x0 + px % kMpgTile;
    if (hi >= a.H || wi >= a.W) continue;
    float g[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) g[j] = 0.f;
    const int h_lo = max(0, (hi + a.PT - 1) >> 1), h_hi = min(a.Ho - 1, (hi + a.PT) >> 1);
    const int w_lo = max(0, (wi + a.PL - 1) >> 1), w_hi = min(a.Wo - 1, (wi + a.PL) >> 1);
    for (int ho = h_lo; ho <= h_hi; ++ho)
      for (int wo = w_lo; wo <= w_hi; ++wo) {
        const uint32_t mine = (uint32_t)((hi - (2 * ho - a.PT)) * 3 + (wi - (2 * wo - a.PL)));
        const int slot = (ho - ho_lo) * kMpgWin + (wo - wo_lo);
        const uint2 ab = *(const uint2*)args[c][slot];
        float dv[EPC];
#pragma unroll
        for (int j = 0; j < EPC; j += 4) {
          const float4 t = *(const float4*)&dys[c][slot][j];
          dv[j] = t.x; dv[j + 1] = t.y; dv[j + 2] = t.z; dv[j + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const uint32_t aj = j < 4 ? (ab.x >> (8 * j)) & 255u : (ab.y >> (8 * (j - 4))) & 255u;
          if (aj == mine) g[j] += dv[j];
        }
      }
    const size_t off = ((size_t)(b * a.H + hi) * a.W + wi) * a.dxcs + a.dxco + (cv0 + c) * EPC;
    float old[EPC];
    load_chunk_f<T>(a.dx, off, a.dx_f32 != 0, old);
#pragma unroll
    for (int j = 0; j < EPC; ++j) old[j] += g[j];
    store_chunk_f<T>(a.dx, off, a.dx_f32 != 0, old);
  }
}

// workgroups per backward-weight launch that the pixel split aims for (every split adds one fp32
// atomic pass over the filter; fewer splits measured slower)
int wgrad_blocks_target() { return 1024; }

// bytes of a conv's (zero-dilated) d-conv tensor, and of the scratch a backward pass needs: the largest one, or -- when the
// weight gradients run on a lane of their own -- all of them side by side
size_t dz_bytes_of(const comic_cnn_op* op, int batch, size_t es) {
  const int Hd = (op->Ho - 1) * op->SH + 1;
  const int Wd = (op->Wo - 1) * op->SW + 1;
  const int regions = (op->flags & COMIC_OP_X3) ? 3 : 1;      // x3 plans: d conv as [hi | lo | hi]
  return ((size_t)batch * Hd * Wd * op->Cout * regions * es + 255) & ~(size_t)255;
}
// (the scheduled backward keeps kMaskCopies partial d beta rows per conv behind the d-conv slices: fused activation gradients)
size_t dbeta_partial_bytes(const comic_cnn_op* op) { return ((size_t)kMaskCopies * op->Cout * sizeof(float) + 255) & ~(size_t)255; }
int64_t backward_scratch_bytes(const comic_cnn_op* ops, int n_ops, int batch, size_t es, bool all) {
  size_t best = 0, sum = 0;
  for (int i = 0; i < n_ops; ++i) {
    if (ops[i].kind > 1) continue;
    const size_t dz = dz_bytes_of(ops + i, batch, es);
    best = std::max(best, dz);
    sum += dz + dbeta_partial_bytes(ops + i);
  }
  return (int64_t)(all ? sum : best);
}

// d beta[c] += the kMaskCopies partial sums of every conv whose activation gradient ran in a backward-data epilogue; one
// workgroup per conv, the entries as kernel arguments
constexpr int kFoldMax = 120;
struct DbetaFoldTable {
  struct {
    const float* part;
    float* dbeta;
    int C, pad;
  } e[kFoldMax];
  int n;
};
static_assert(sizeof(DbetaFoldTable) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void dbeta_fold_kernel(const DbetaFoldTable tb) {
  const auto& en = tb.e[blockIdx.x];
  for (int c = threadIdx.x; c < en.C; c += blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < kMaskCopies; ++r) s += en.part[(size_t)r * en.C + c];
    en.dbeta[c] += s;
  }
}

// One conv's weight gradient dw += x^T dz (the d-conv tensor dz as act_grad_kernel / a fused backward-data epilogue wrote it).
struct PendingWgrad {
  const comic_cnn_op* op;
  const void* x;
  int xc;
  const void* dz;
  const comic_conv_grad* gr;
};
template <typename T>
int launch_wgrad(const comic_cnn_op* op, const void* x, int xc, const void* dz, const comic_conv_grad* gr, int batch, hipStream_t st) {
  constexpr int EPC = Elem<T>::EPC;
  const bool stem = op->kind == 1;
  const int K = op->KH * op->KW * op->Cin, Kpad = (K + 63) / 64 * 64;
  const int dil = op->SH;
  const int Hd = (op->Ho - 1) * dil + 1, Wd = (op->Wo - 1) * dil + 1;
  {
    WgradArgs a{};
    a.dz_cs = op->Cout; a.dz_co = 0; a.x_part = 0;
    a.dz = dz; a.Hd = Hd; a.Wd = Wd; a.dil = dil; a.x = x; a.x_cs = xc; a.x_co = op->src_coff; a.dw = gr->dw;
    a.B = batch; a.H = op->H; a.W = op->W; a.Cin = op->Cin; a.Cout = op->Cout; a.KH = op->KH; a.KW = op->KW;
    a.SH = op->SH; a.SW = op->SW; a.PT = op->PT; a.PL = op->PL; a.Ho = op->Ho; a.Wo = op->Wo; a.K = K; a.Kpad = Kpad;
    a.P = (long)batch * op->Ho * op->Wo;
    constexpr int BKE = 4 * EPC;
    const int tiles = cdiv(K, 64) * cdiv(op->Cout, 64);
    // (stem: K = 27 taps x 32 channels = 864 sums that EVERY workgroup adds into -- 2048 pixel slices queue up at those
    // addresses: Conv2d_1a's weight gradient, the last launch of the backward, 178 -> ~100 us with a quarter of the slices)
    long S = std::max<long>(1, std::min<long>((stem ? 512 : 2048) / tiles, cdiv64(a.P, (long)BKE * 8)));
    a.p_per_split = cdiv64(cdiv64(a.P, S), BKE) * BKE;
    S = cdiv64(a.P, a.p_per_split);
    dim3 grid(cdiv(K, 64), cdiv(op->Cout, 64), (unsigned)S);
    if (stem) {
      COMIC_REQUIRE(op->Cin <= 4, "stem conv backward: needs Cin <= 4");
      hipLaunchKernelGGL((conv_wgrad_kernel<T, true>), grid, dim3(256), 0, st, a);
    } else {
      COMIC_REQUIRE(op->Cin % EPC == 0 && op->src_coff % EPC == 0 && xc % EPC == 0, "conv backward: misaligned input slice");
      if constexpr (sizeof(T) == 2) {
        const bool big_m = op->Cout % 128 == 0, big_n = Kpad % 128 == 0 && K >= 256;
        const int bm = big_m ? 128 : 64, bn = big_n ? 128 : 64;
        const int tiles2 = cdiv(K, bn) * cdiv(op->Cout, bm);
        long S2 = std::max<long>(1, std::min<long>(wgrad_blocks_target() / tiles2, cdiv64(a.P, 32L * 8)));
        a.p_per_split = cdiv64(cdiv64(a.P, S2), 32) * 32;
        S2 = cdiv64(a.P, a.p_per_split);
        dim3 g2(cdiv(K, bn), cdiv(op->Cout, bm), (unsigned)S2);
        if (big_m && big_n)
          hipLaunchKernelGGL((conv_wgrad_tr_kernel<128, 128>), g2, dim3(256), 0, st, a);
        else if (big_m)
          hipLaunchKernelGGL((conv_wgrad_tr_kernel<128, 64>), g2, dim3(256), 0, st, a);
        else if (big_n)
          hipLaunchKernelGGL((conv_wgrad_tr_kernel<64, 128>), g2, dim3(256), 0, st, a);
        else
          hipLaunchKernelGGL((conv_wgrad_tr_kernel<64, 64>), g2, dim3(256), 0, st, a);
      } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<T, false>), grid, dim3(256), 0, st, a);
      }
    }
  }
  return 0;
}

template <typename T>
int conv_backward(const comic_cnn_op* op, const void* x, int xc, const void* y, const void* gy, int yc, void* gx,
                  const comic_conv_weight* wt, const comic_conv_grad* gr, int batch, void* scratch,
                  int64_t scratch_bytes, hipStream_t st, bool filters_ready, hipStream_t st_w,
                  bool dz_ready = false, const ConvMask* fuse = nullptr, std::vector<PendingWgrad>* defer = nullptr) {
  // dz_ready: the backward-data launch of this conv's only reader has written dz and added d beta (its epilogue applied this
  // conv's activation gradient); fuse: this conv's backward-data result is the gradient at the output of a conv with no
  // other reader -- apply that conv's activation gradient in the epilogue and write its dz instead of accumulating into gx.
  constexpr int EPC = Elem<T>::EPC;
  const bool stem = op->kind == 1;
  COMIC_REQUIRE(wt && wt->scale && gr && gr->w_master && gr->dw && gr->dbeta, "conv backward: missing weight / gradient record");
  COMIC_REQUIRE(op->Cout % EPC == 0 && op->dst_coff % EPC == 0 && yc % EPC == 0, "conv backward: misaligned output slice");
  COMIC_REQUIRE(op->SH == op->SW && (op->SH == 1 || op->SH == 2), "conv backward: stride must be 1 or 2");
  const int K = op->KH * op->KW * op->Cin, Kpad = (K + 63) / 64 * 64;
  const int dil = op->SH;
  // zero-dilated d conv geometry: output pixel (ho, wo) sits at (ho*dil, wo*dil); the backward-data
  // conv pads PT' = KH - 1 - PT rows on top and reads zeros below the buffer (bounds checks)
  const int Hd = (op->Ho - 1) * dil + 1;
  const int Wd = (op->Wo - 1) * dil + 1;
  const size_t dz_bytes = ((size_t)batch * Hd * Wd * op->Cout * sizeof(T) + 255) & ~(size_t)255;
  COMIC_REQUIRE((int64_t)dz_bytes <= scratch_bytes, "conv backward: scratch too small (%zu needed)", dz_bytes);
  T* dz = (T*)scratch;
  COMIC_REQUIRE(!dz_ready || dil == 1, "conv backward: a fused activation gradient writes an undilated d-conv tensor");
  if (dil > 1) {
    COMIC_REQUIRE(hipMemsetAsync(dz, 0, dz_bytes, st) == hipSuccess, "conv backward: memset failed");
  }
  if (!dz_ready) {
    ActGradArgs a{y, gy, yc, op->dst_coff, op->out_f32, wt->scale, dz, batch, op->Ho, op->Wo, op->Cout, Hd, Wd, dil};
    const long P = (long)batch * op->Ho * op->Wo;
    const long ppb = std::max<long>(64, cdiv64(P, 512));
    hipLaunchKernelGGL((act_grad_kernel<T>), dim3((unsigned)cdiv64(P, ppb), cdiv(op->Cout, 64)), dim3(256), 0, st, a,
                       gr->dbeta, ppb);
  }
  if (defer) {            // the caller launches the weight gradient later (launch_wgrad), behind one fork for several convs
    defer->push_back(PendingWgrad{op, x, xc, (const void*)dz, gr});
  } else {
    if (st_w != st) {     // the weight gradient runs on its own lane, beside the backward-data chain (dz is this conv's own)
      hipEvent_t ev;
      COMIC_REQUIRE(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "conv backward: event");
      const bool ok = hipEventRecord(ev, st) == hipSuccess && hipStreamWaitEvent(st_w, ev, 0) == hipSuccess;
      (void)hipEventDestroy(ev);   // released once it has completed
      COMIC_REQUIRE(ok, "conv backward: fork of the weight-gradient lane failed");
    }
    if (int rc = launch_wgrad<T>(op, x, xc, dz, gr, batch, st_w)) return rc;
  }
  COMIC_LAUNCH_CHECK("conv backward (weights)");
  if (stem || (!gx && !fuse)) return 0;
  // backward-data: forward conv of dz with the flipped / transposed filter, accumulated into gx
  COMIC_REQUIRE(gr->w_bwd, "conv backward: missing backward-data filter buffer");
  const int K2 = op->KH * op->KW * op->Cout, Kpad2 = (K2 + 63) / 64 * 64;
  if (!filters_ready) {
    const long n = (long)op->Cin * Kpad2;
    hipLaunchKernelGGL((pack_bwd_weights_kernel<T>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, gr->w_master,
                       (T*)gr->w_bwd, op->KH, op->KW, op->Cin, op->Cout, Kpad, Kpad2);
  }
  comic_cnn_op t = *op;
  t.kind = 0;
  t.H = Hd; t.W = Wd; t.Cin = op->Cout; t.Cout = op->Cin; t.SH = t.SW = 1;
  t.PT = op->KH - 1 - op->PT; t.PL = op->KW - 1 - op->PL;
  t.Ho = op->H; t.Wo = op->W;
  t.src_coff = 0; t.dst_coff = op->src_coff; t.relu = 0; t.out_f32 = 0; t.tile = gr->bwd_tile; t.group = 0; t.lane = 0;
  comic_conv_weight w2{gr->w_bwd, nullptr, nullptr};
  if (fuse) {
    t.dst_coff = 0;
    return run_op<T>(&t, dz, op->Cout, fuse->dz, op->Cin, &w2, batch, st, /*accum=*/0, fuse);
  }
  return run_op<T>(&t, dz, op->Cout, gx, xc, &w2, batch, st, /*accum=*/1);
}

inline bool link_streams(hipStream_t from, hipStream_t to) {     // `to` waits for everything issued on `from` so far
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
  const bool ok = hipEventRecord(ev, from) == hipSuccess && hipStreamWaitEvent(to, ev, 0) == hipSuccess;
  (void)hipEventDestroy(ev);
  return ok;
}

// A conv of a COMIC_OP_X3 plan ("bf16x3": cnn_finetune at fp32-class accuracy on the bf16 matrix cores).  Activations are
// the three bf16 regions [hi | lo | hi] (xc, yc: physical channels = 3 x logical; fp32 outputs as they are), gradient
// buffers fp32 of the LOGICAL channels (gxc, gyc), masters and weight gradients the logical [Cout][Kpad] of the bf16 plan.
//   dz  = 1[y > 0] dy scale as [hi | lo | hi]                                              (act_grad_x3_kernel)
//   dw += x_hi^T dz_hi + x_lo^T dz_hi + x_hi^T dz_lo      one launch of the bf16 backward-weight kernel, three work items per
//                                                         (tile, pixel split) over channel slices of the regions, fp32
//                                                         atomics into the one dw
//   gx += conv(dz, [Wt_hi | Wt_hi | Wt_lo])               the bf16 forward kernels over 3 Cout channels, fp32 accumulate
int conv_backward_x3(const comic_cnn_op* op, const void* x, int xc, const void* y, const float* gy, int yc, int gyc, float* gx,
                     int gxc, const comic_conv_weight* wt, const comic_conv_grad* gr, int batch, void* scratch,
                     int64_t scratch_bytes, hipStream_t st, bool filters_ready, hipStream_t st_w) {
  const bool stem = op->kind == 1;
  COMIC_REQUIRE(wt && wt->scale && gr && gr->w_master && gr->dw && gr->dbeta, "conv backward (x3): missing weight / gradient record");
  COMIC_REQUIRE(op->SH == op->SW && (op->SH == 1 || op->SH == 2), "conv backward (x3): stride must be 1 or 2");
  COMIC_REQUIRE(op->Cout % 8 == 0 && op->dst_coff % 8 == 0 && (stem || (op->Cin % 3 == 0 && xc % 3 == 0)) &&
                    (op->out_f32 || yc % 3 == 0), "conv backward (x3): channel regions do not fit the buffers");
  const int cin = stem ? op->Cin : op->Cin / 3, x_lo = stem ? 0 : xc / 3, y_lo = op->out_f32 ? 0 : yc / 3;
  const int K = op->KH * op->KW * cin, Kpad = (K + 63) / 64 * 64;
  const int dil = op->SH;
  const int Hd = (op->Ho - 1) * dil + 1, Wd = (op->Wo - 1) * dil + 1;
  const size_t dz_bytes = dz_bytes_of(op, batch, 2);
  COMIC_REQUIRE((int64_t)dz_bytes <= scratch_bytes, "conv backward (x3): scratch too small (%zu needed)", dz_bytes);
  bf16_t* dz = (bf16_t*)scratch;
  if (dil > 1) COMIC_REQUIRE(hipMemsetAsync(dz, 0, dz_bytes, st) == hipSuccess, "conv backward (x3): memset failed");
  {
    ActGradX3Args a{y, gy, yc, op->dst_coff, y_lo, gyc, wt->scale, dz, batch, op->Ho, op->Wo, op->Cout, Hd, Wd, dil};
    const long P = (long)batch * op->Ho * op->Wo;
    const long ppb = std::max<long>(64, cdiv64(P, 512));
    hipLaunchKernelGGL(act_grad_x3_kernel, dim3((unsigned)cdiv64(P, ppb), cdiv(op->Cout, 64)), dim3(256), 0, st, a, gr->dbeta, ppb);
  }
  if (st_w != st) COMIC_REQUIRE(link_streams(st, st_w), "conv backward (x3): fork of the weight-gradient lane failed");
  {
    WgradArgs a{};
    a.dz = dz; a.Hd = Hd; a.Wd = Wd; a.dil = dil; a.dz_cs = 3 * op->Cout; a.x = x; a.x_cs = xc; a.dw = gr->dw;
    a.B = batch; a.H = op->H; a.W = op->W; a.Cin = cin; a.Cout = op->Cout; a.KH = op->KH; a.KW = op->KW;
    a.SH = op->SH; a.SW = op->SW; a.PT = op->PT; a.PL = op->PL; a.Ho = op->Ho; a.Wo = op->Wo; a.K = K; a.Kpad = Kpad;
    a.P = (long)batch * op->Ho * op->Wo;
    // the three products (x hi, dz hi), (x lo, dz hi), (x hi, dz lo) in one launch: blockIdx.z = split * 3 + part
    a.parts = 3; a.dz_co = 0; a.x_co = op->src_coff; a.x_part = 0; a.x_lo_off = x_lo; a.z_lo_off = op->Cout;
    if (stem) {
      COMIC_REQUIRE(op->Cin <= 4, "stem conv backward: needs Cin <= 4");
      const int tiles = cdiv(K, 64) * cdiv(op->Cout, 64) * 3;
      long S = std::max<long>(1, std::min<long>(2048 / tiles, cdiv64(a.P, 32L * 8)));
      a.p_per_split = cdiv64(cdiv64(a.P, S), 32) * 32;
      S = cdiv64(a.P, a.p_per_split);
      hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, true>), dim3(cdiv(K, 64), cdiv(op->Cout, 64), (unsigned)(3 * S)), dim3(256), 0,
                         st_w, a);
    } else {
      COMIC_REQUIRE(cin % 8 == 0 && op->src_coff % 8 == 0 && x_lo % 8 == 0, "conv backward (x3): misaligned input slice");
      const bool big_m = op->Cout % 128 == 0, big_n = Kpad % 128 == 0 && K >= 256;
      const int bm = big_m ? 128 : 64, bn = big_n ? 128 : 64;
      const int tiles2 = cdiv(K, bn) * cdiv(op->Cout, bm) * 3;
      long S2 = std::max<long>(1, std::min<long>(wgrad_blocks_target() / tiles2, cdiv64(a.P, 32L * 8)));
      a.p_per_split = cdiv64(cdiv64(a.P, S2), 32) * 32;
      S2 = cdiv64(a.P, a.p_per_split);
      dim3 g2(cdiv(K, bn), cdiv(op->Cout, bm), (unsigned)(3 * S2));
      if (big_m && big_n) hipLaunchKernelGGL((conv_wgrad_tr_kernel<128, 128>), g2, dim3(256), 0, st_w, a);
      else if (big_m) hipLaunchKernelGGL((conv_wgrad_tr_kernel<128, 64>), g2, dim3(256), 0, st_w, a);
      else if (big_n) hipLaunchKernelGGL((conv_wgrad_tr_kernel<64, 128>), g2, dim3(256), 0, st_w, a);
      else hipLaunchKernelGGL((conv_wgrad_tr_kernel<64, 64>), g2, dim3(256), 0, st_w, a);
    }
  }
  COMIC_LAUNCH_CHECK("conv backward (x3, weights)");
  if (stem || !gx) return 0;
  COMIC_REQUIRE(gr->w_bwd, "conv backward (x3): missing backward-data filter buffer");
  const int K2 = op->KH * op->KW * 3 * op->Cout, Kpad2 = (K2 + 63) / 64 * 64;
  if (!filters_ready) {
    const long n = (long)cin * Kpad2;
    hipLaunchKernelGGL(pack_bwd_weights_x3_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, gr->w_master,
                       (bf16_t*)gr->w_bwd, op->KH, op->KW, cin, op->Cout, Kpad, Kpad2);
  }
  comic_cnn_op t = *op;
  t.kind = 0;
  t.flags = COMIC_OP_RAW;
  t.H = Hd; t.W = Wd; t.Cin = 3 * op->Cout; t.Cout = cin; t.SH = t.SW = 1;
  t.PT = op->KH - 1 - op->PT; t.PL = op->KW - 1 - op->PL;
  t.Ho = op->H; t.Wo = op->W;
  t.src_coff = 0; t.dst_coff = op->src_coff; t.relu = 0; t.out_f32 = 1; t.tile = 0; t.group = 0; t.lane = 0;
  comic_conv_weight w2{gr->w_bwd, nullptr, nullptr};
  return run_op<bf16_t>(&t, dz, 3 * op->Cout, gx, gxc, &w2, batch, st, /*accum=*/1);
}

// x3: the pool of a COMIC_OP_X3 plan -- x the bf16 regions (hi + lo), gy / gx fp32 buffers of the logical channels
template <typename T>
int pool_backward(const comic_cnn_op* op, const void* x, int xc, const void* gy, int yc, void* gx, int batch,
                  hipStream_t st, bool x3 = false) {
  constexpr int EPC = Elem<T>::EPC;
  COMIC_REQUIRE(op->Cin % EPC == 0 && op->src_coff % EPC == 0 && op->dst_coff % EPC == 0 && xc % EPC == 0 &&
                    yc % EPC == 0, "pool backward: misaligned channel slices");
  PoolGradArgs a{x, gy, gx, xc, op->src_coff, yc, op->dst_coff, op->kind == 4 ? 1 : 0,
                 (op->kind == 4 && op->src_f32) ? 1 : 0,
                 batch, op->H, op->W, op->Cin, op->KH, op->KW, op->SH, op->SW, op->PT, op->PL, op->Ho, op->Wo};
  a.dxcs = a.xcs; a.dxco = a.xco; a.x_lo = 0;
  if (x3) {
    COMIC_REQUIRE(sizeof(T) == 2 && xc % 3 == 0 && yc % 3 == 0 && (xc / 3) % EPC == 0 && (yc / 3) % EPC == 0,
                  "pool backward (x3): channel regions do not fit the buffers");
    a.x_lo = xc / 3; a.dxcs = xc / 3; a.ycs = yc / 3; a.dy_f32 = 1; a.dx_f32 = 1;
  }
  const long total = (long)batch * op->H * op->W * (op->Cin / EPC);
  dim3 grid((unsigned)cdiv64(total, 256));
  if (op->kind == 2 && op->KH == 3 && op->KW == 3 && op->SH == 2 && op->SW == 2 && op->PT >= 0 && op->PT <= 1 && op->PL >= 0 &&
      op->PL <= 1) {
    const int tiles_y = cdiv(op->H, kMpgTile), tiles_x = cdiv(op->W, kMpgTile), cgroups = cdiv(op->Cin / EPC, kMpgCG);
    const long wgs = (long)batch * tiles_y * tiles_x * cgroups;
    COMIC_REQUIRE(wgs < (1L << 31), "pool backward: too many tiles");
    if constexpr (sizeof(T) == 2) {
      if (x3) hipLaunchKernelGGL((maxpool_grad_s2_kernel<T, true>), dim3((unsigned)wgs), dim3(256), 0, st, a, tiles_y, tiles_x, cgroups);
      else hipLaunchKernelGGL((maxpool_grad_s2_kernel<T>), dim3((unsigned)wgs), dim3(256), 0, st, a, tiles_y, tiles_x, cgroups);
    } else {
      hipLaunchKernelGGL((maxpool_grad_s2_kernel<T>), dim3((unsigned)wgs), dim3(256), 0, st, a, tiles_y, tiles_x, cgroups);
    }
  } else if (op->kind == 2)
    hipLaunchKernelGGL((pool_grad_kernel<T, 0>), grid, dim3(256), 0, st, a);
  else if (op->kind == 3)
    hipLaunchKernelGGL((pool_grad_kernel<T, 1>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((pool_grad_kernel<T, 2>), grid, dim3(256), 0, st, a);
  COMIC_LAUNCH_CHECK("pool backward");
  return 0;
}

// Every conv of a plan in ONE launch (the entries travel as kernel arguments): 93 launches of 5 us each cost the host thread
// 0.63 ms per cnn_finetune step -- the next forward's first launch queued behind them -- for 0.4 ms of device work.
constexpr int kPackTableMax = 96;
struct PackBwdEntry {
  const float* master;
  void* out;
  uint32_t first_block;            // of this entry in the launch (entries ascending)
  uint16_t Cin, Cout;
  uint8_t KH, KW;
  uint8_t x3;                      // 1: [W_hi | W_hi | W_lo] regions per tap (Cin = channels of ONE source region)
  uint8_t pad[5];
};
struct PackBwdTable {
  PackBwdEntry e[kPackTableMax];
  int n;
};
static_assert(sizeof(PackBwdTable) <= 4096, "kernel argument block");

template <typename T>
__global__ __launch_bounds__(256) void pack_bwd_weights_table_kernel(const PackBwdTable tb) {
  int lo = 0, hi = tb.n - 1;                             // the entry this block belongs to
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.e[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackBwdEntry& en = tb.e[lo];
  const int KH = en.KH, KW = en.KW, Cin = en.Cin, Cout = en.Cout;
  const int Kpad = (KH * KW * Cin + 63) / 64 * 64, Kpad2 = (KH * KW * Cout * (en.x3 ? 3 : 1) + 63) / 64 * 64;
  const long idx = (long)(blockIdx.x - en.first_block) * blockDim.x + threadIdx.x;
  if (idx >= (long)Cin * Kpad2) return;
  if (en.x3) {
    if constexpr (sizeof(T) == 2)
      ((bf16_t*)en.out)[idx] = pack_bwd_x3_value(en.master, (int)(idx % Kpad2), (int)(idx / Kpad2), KH, KW, Cin, Cout, Kpad);
    return;
  }
  const int k2 = (int)(idx % Kpad2), ci = (int)(idx / Kpad2);
  float v = 0.f;
  if (k2 < KH * KW * Cout) {
    const int tap2 = k2 / Cout, co = k2 % Cout;
    const int kh = KH - 1 - tap2 / KW, kw = KW - 1 - tap2 % KW;
    v = en.master[(size_t)co * Kpad + (kh * KW + kw) * Cin + ci];
  }
  if (sizeof(T) == 4)
    ((float*)en.out)[idx] = v;
  else
    ((bf16_t*)en.out)[idx] = f32_to_bf16(v);
}

template <typename T>
int pack_bwd_filters_impl(const comic_cnn_op* ops, int n_ops, const comic_conv_grad* grads, hipStream_t st) {
  PackBwdTable tb;
  tb.n = 0;
  uint32_t blocks = 0;
  auto flush = [&]() {
    if (tb.n) hipLaunchKernelGGL((pack_bwd_weights_table_kernel<T>), dim3(blocks), dim3(256), 0, st, tb);
    tb.n = 0;
    blocks = 0;
  };
  for (int i = 0; i < n_ops; ++i) {
    const comic_cnn_op* op = ops + i;
    if (op->kind != 0) continue;
    const comic_conv_grad* gr = grads + op->weight;
    COMIC_REQUIRE(gr->w_master && gr->w_bwd, "pack_bwd_filters: missing buffers for conv %d", i);
    COMIC_REQUIRE(op->KH <= 255 && op->KW <= 255 && op->Cin <= 65535 && op->Cout <= 65535, "pack_bwd_filters: conv %d too large", i);
    const bool x3 = (op->flags & COMIC_OP_X3) != 0;
    COMIC_REQUIRE(!x3 || (sizeof(T) == 2 && op->Cin % 3 == 0), "pack_bwd_filters: COMIC_OP_X3 is a bf16-plan layout");
    const int cin = x3 ? op->Cin / 3 : op->Cin;
    const int K2 = op->KH * op->KW * op->Cout * (x3 ? 3 : 1), Kpad2 = (K2 + 63) / 64 * 64;
    const long n = (long)cin * Kpad2;
    if (tb.n == kPackTableMax) flush();
    PackBwdEntry& en = tb.e[tb.n++];
    en.master = gr->w_master;
    en.out = gr->w_bwd;
    en.first_block = blocks;
    en.x3 = x3 ? 1 : 0;
    en.Cin = (uint16_t)cin;
    en.Cout = (uint16_t)op->Cout;
    en.KH = (uint8_t)op->KH;
    en.KW = (uint8_t)op->KW;
    blocks += (uint32_t)cdiv64(n, 256);
  }
  flush();
  COMIC_LAUNCH_CHECK("pack_bwd_filters");
  return 0;
}

// forward filters of a COMIC_OP_X3 plan from the masters, every conv in one launch (comic_cnn_pack_x3_weights)
struct PackX3Entry {
  const float* master;
  bf16_t* out;
  uint32_t first_block;
  uint16_t cout, cin, taps, pad;
};
struct PackX3Table {
  PackX3Entry e[kPackTableMax];
  int n;
};
static_assert(sizeof(PackX3Table) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void pack_x3_table_kernel(const PackX3Table tb) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.e[mid].first_block <= blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackX3Entry& en = tb.e[lo];
  const int cin = en.cin, K = en.taps * cin, kpad = (K + 63) / 64 * 64, kpad3 = (3 * K + 63) / 64 * 64;
  const long idx = (long)(blockIdx.x - en.first_block) * blockDim.x + threadIdx.x;
  if (idx >= (long)en.cout * kpad3) return;
  const int k3 = (int)(idx % kpad3), row = (int)(idx / kpad3);
  bf16_t o = f32_to_bf16(0.f);
  if (k3 < 3 * K) {
    const int tap = k3 / (3 * cin), rr = k3 - tap * 3 * cin;
    const int region = rr / cin, ci = rr - region * cin;
    const float v = en.master[(size_t)row * kpad + tap * cin + ci];
    const bf16_t h = f32_to_bf16(v);
    o = region < 2 ? h : f32_to_bf16(v - bf16_to_f32(h));
  }
  en.out[idx] = o;
}

template <typename T>
int cnn_backward_impl(const comic_cnn_op* ops, int n_ops, void* const* buffers, void* const* grad_buffers,
                      const int32_t* buf_channels, const comic_conv_weight* weights, const comic_conv_grad* grads,
                      int batch, void* scratch, int64_t scratch_bytes, hipStream_t st, bool filters_ready,
                      hipStream_t st_w) {
  // Two lanes (st_w != st): every conv gets its own d-conv slice of the scratch, so its weight gradient (st_w) only
  // waits for its act_grad launch and runs beside the act_grad / backward-data chain of the earlier layers (st).
  const int64_t all = backward_scratch_bytes(ops, n_ops, batch, sizeof(T), true);
  if (st_w == nullptr || all > scratch_bytes) st_w = st;
  size_t dz_off = 0;
  for (int i = n_ops - 1; i >= 0; --i) {
    const comic_cnn_op* op = ops + i;
    if (op->kind == 5 || op->kind == 6) continue;
    void* gy = grad_buffers[op->dst];
    COMIC_REQUIRE(gy, "cnn_backward: op %d has no output gradient buffer", i);
    void* gx = grad_buffers[op->src];
    const int xc = buf_channels[op->src], yc = buf_channels[op->dst];
    const bool x3 = (op->flags & COMIC_OP_X3) != 0;
    if (x3 && op->kind <= 1) {
      if constexpr (sizeof(T) == 2) {
        const size_t dzb = dz_bytes_of(op, batch, sizeof(T));
        void* dz = st_w != st ? (void*)((char*)scratch + dz_off) : scratch;
        if (int rc = conv_backward_x3(op, buffers[op->src], xc, buffers[op->dst], (const float*)gy, yc, op->out_f32 ? yc : yc / 3,
                                      (float*)gx, op->kind == 1 ? 0 : xc / 3, weights + op->weight, grads + op->weight, batch, dz,
                                      st_w != st ? (int64_t)dzb : scratch_bytes, st, filters_ready, st_w))
          return rc;
        dz_off += dzb;
      } else {
        COMIC_REQUIRE(false, "cnn_backward: COMIC_OP_X3 is a bf16-plan layout");
      }
    } else if (op->kind <= 1) {
      const size_t dzb = dz_bytes_of(op, batch, sizeof(T));
      void* dz = st_w != st ? (void*)((char*)scratch + dz_off) : scratch;
      if (int rc = conv_backward<T>(op, buffers[op->src], xc, buffers[op->dst], gy, yc, gx, weights + op->weight,
                                    grads + op->weight, batch, dz, st_w != st ? (int64_t)dzb : scratch_bytes, st,
                                    filters_ready, st_w))
        return rc;
      dz_off += dzb;
    } else if (op->kind <= 4) {
      if (!gx) continue;
      if (int rc = pool_backward<T>(op, buffers[op->src], xc, gy, yc, gx, batch, st, x3 && op->kind <= 3)) return rc;
    } else {
      COMIC_REQUIRE(false, "cnn_backward: unknown op kind %d", op->kind);
    }
  }
  if (st_w != st) {      // join: what follows on st (all-reduce, optimiser) sees every weight gradient
    hipEvent_t ev;
    COMIC_REQUIRE(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "cnn_backward: event");
    const bool ok = hipEventRecord(ev, st_w) == hipSuccess && hipStreamWaitEvent(st, ev, 0) == hipSuccess;
    (void)hipEventDestroy(ev);
    COMIC_REQUIRE(ok, "cnn_backward: join of the weight-gradient lane failed");
  }
  return 0;
}

// y += x; x = 0 (the join of the two chain lanes of comic_cnn_backward_sched)
template <typename T>
__global__ void add_clear_kernel(T* __restrict__ y, T* __restrict__ x, long n_chunks) {
  constexpr int EPC = Elem<T>::EPC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_chunks) return;
  float a[EPC], b[EPC];
  load_vec<T>(y + i * EPC, a);
  load_vec<T>(x + i * EPC, b);
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    a[j] += b[j];
    b[j] = 0.f;
  }
  store_vec<T>(y + i * EPC, a);
  store_vec<T>(x + i * EPC, b);
}



template <typename T>
int cnn_backward_sched_impl(const comic_cnn_op* ops, int n_ops, const int32_t* sched, int n_sched, void* const* buffers,
                            void* const* grad_buffers, void* const* grad_alt, const int32_t* buf_channels,
                            const comic_conv_weight* weights, const comic_conv_grad* grads, int batch, void* scratch,
                            int64_t scratch_bytes, hipStream_t s0, hipStream_t s1, hipStream_t st_w, bool filters_ready,
                            bool no_act_fusion) {
  constexpr int EPC = Elem<T>::EPC;
  COMIC_REQUIRE(s1 && s1 != s0 && st_w && st_w != s0 && st_w != s1, "cnn_backward_sched: needs three distinct streams");
  COMIC_REQUIRE(backward_scratch_bytes(ops, n_ops, batch, sizeof(T), true) <= scratch_bytes,
                "cnn_backward_sched: scratch too small (every conv needs its own d-conv slice)");
  std::vector<size_t> dz_off((size_t)n_ops, 0), part_off((size_t)n_ops, 0);
  size_t off = 0;
  for (int i = 0; i < n_ops; ++i)
    if (ops[i].kind <= 1) {
      dz_off[i] = off;
      off += dz_bytes_of(ops + i, batch, sizeof(T));
    }
  const size_t part0 = off;
  for (int i = 0; i < n_ops; ++i)
    if (ops[i].kind <= 1) {
      part_off[i] = off;
      off += dbeta_partial_bytes(ops + i);
    }
  COMIC_REQUIRE(hipMemsetAsync((char*)scratch + part0, 0, off - part0, s0) == hipSuccess, "cnn_backward_sched: memset failed");
  // Activation-gradient fusion: conv p whose output slice is read by exactly one op, a conv i on the same lane, gets its d-conv
  // tensor and d beta from the epilogue of i's backward-data launch (the inner convs of the Inception branches: one launch
  // less per conv on the serial chain a block's longest branch is).  fused_by[p] = i, fuses[i] = p.
  std::vector<int> lane_of((size_t)n_ops, -1), fused_by((size_t)n_ops, -1), fuses((size_t)n_ops, -1);
  for (int k = 0; k < n_sched; ++k)
    if (sched[4 * k] == 0 && sched[4 * k + 1] >= 0 && sched[4 * k + 1] < n_ops) lane_of[sched[4 * k + 1]] = sched[4 * k + 2];
  if (!no_act_fusion)
    for (int i = 0; i < n_ops; ++i) {
      const comic_cnn_op& c = ops[i];
      if (c.kind != 0 || c.out_f32 || (c.flags & COMIC_OP_X3)) continue;
      int p = -1, readers = 0;
      for (int j = 0; j < n_ops; ++j) {
        const comic_cnn_op& o = ops[j];
        // every op that touches this channel range of the buffer as a source (pools, concats, convs)
        if (o.src == c.src && !(o.src_coff + o.Cin <= c.src_coff || c.src_coff + c.Cin <= o.src_coff)) ++readers;
        if (o.kind == 0 && o.dst == c.src && o.dst_coff == c.src_coff && o.Cout == c.Cin) p = j;
      }
      if (p < 0 || readers != 1) continue;
      const comic_cnn_op& q = ops[p];
      if (q.SH != 1 || q.SW != 1 || q.out_f32 || (q.flags & (COMIC_OP_X3 | COMIC_OP_RAW)) || lane_of[p] != lane_of[i] || lane_of[i] < 0) continue;
      // nothing else may write into that slice (a pool or a second conv) or read the gradient buffer's slice
      bool only_writer = true;
      for (int j = 0; j < n_ops; ++j)
        if (j != p && ops[j].dst == c.src && ops[j].kind != 5 && ops[j].kind != 6 &&
            !(ops[j].dst_coff + (ops[j].kind == 0 ? ops[j].Cout : ops[j].Cin) <= c.src_coff || c.src_coff + c.Cin <= ops[j].dst_coff))
          only_writer = false;
      if (!only_writer) continue;
      fused_by[p] = i;
      fuses[i] = p;
    }
  std::vector<char> seen((size_t)n_ops, 0);
  bool lane1_open = false;
  // Weight gradients of the convs inside a fork / join region (the branches of an Inception block) are launched at the
  // region's join, behind ONE fork of the weight-gradient lane, instead of one fork per conv between two launches of a
  // chain lane: an event record in front of every backward-data launch kept the chain lanes idle for longer than their
  // kernels ran (10-28 us between dependent launches of 6-25 us, profiles/r05_finetune_lanes.txt).  Every conv owns its
  // d-conv slice of the scratch, so a later launch reads the same bytes.  Outside the regions (the stem's serial chain of
  // large convs) a weight gradient still starts as soon as its d-conv tensor exists.
  std::vector<PendingWgrad> pend;
  auto flush_wgrads = [&]() -> int {
    if (pend.empty()) return 0;
    COMIC_REQUIRE(link_streams(s0, st_w), "cnn_backward_sched: fork of the weight-gradient lane failed");
    for (const PendingWgrad& w : pend)
      if (int rc = launch_wgrad<T>(w.op, w.x, w.xc, w.dz, w.gr, batch, st_w)) return rc;
    pend.clear();
    COMIC_LAUNCH_CHECK("cnn_backward_sched (weight gradients of a block)");
    return 0;
  };
  for (int k = 0; k < n_sched; ++k) {
    const int32_t* r = sched + 4 * k;
    if (r[0] == 1) {                       // FORK
      COMIC_REQUIRE(link_streams(s0, s1), "cnn_backward_sched: fork failed");
      lane1_open = true;
      continue;
    }
    if (r[0] == 2) {                       // JOIN_ADD
      COMIC_REQUIRE(link_streams(s1, s0), "cnn_backward_sched: join failed");
      lane1_open = false;
      if (int rc = flush_wgrads()) return rc;
      if (r[1] >= 0) {
        const comic_cnn_op* any = nullptr;
        for (int i = 0; i < n_ops && !any; ++i)
          if (ops[i].kind != 5 && ops[i].kind != 6 && ops[i].src == r[1]) any = ops + i;
        COMIC_REQUIRE(any && grad_buffers[r[1]] && grad_alt && grad_alt[r[1]], "cnn_backward_sched: no alternate buffer %d", r[1]);
        const long n = (long)batch * any->H * any->W * buf_channels[r[1]];
        COMIC_REQUIRE(n % EPC == 0, "cnn_backward_sched: buffer %d is not a whole number of 16-byte chunks", r[1]);
        if ((any->flags & COMIC_OP_X3) && any->kind <= 3 && any->kind != 1) {      // x3 plans: fp32 gradients of the logical channels
          COMIC_REQUIRE(n % 12 == 0, "cnn_backward_sched: buffer %d is not three regions of whole chunks", r[1]);
          hipLaunchKernelGGL((add_clear_kernel<float>), dim3((unsigned)cdiv64(n / 12, 256)), dim3(256), 0, s0,
                             (float*)grad_buffers[r[1]], (float*)grad_alt[r[1]], n / 12);
        } else
        hipLaunchKernelGGL((add_clear_kernel<T>), dim3((unsigned)cdiv64(n / EPC, 256)), dim3(256), 0, s0,
                           (T*)grad_buffers[r[1]], (T*)grad_alt[r[1]], n / EPC);
      }
      continue;
    }
    COMIC_REQUIRE(r[0] == 0 && r[1] >= 0 && r[1] < n_ops && !seen[r[1]], "cnn_backward_sched: bad schedule row %d", k);
    seen[r[1]] = 1;
    const comic_cnn_op* op = ops + r[1];
    COMIC_REQUIRE(r[2] == 0 || lane1_open, "cnn_backward_sched: lane 1 used outside a fork / join region");
    hipStream_t st = r[2] ? s1 : s0;
    void* gy = grad_buffers[op->dst];
    COMIC_REQUIRE(gy, "cnn_backward_sched: op %d has no output gradient buffer", r[1]);
    void* gx = r[3] ? (grad_alt ? grad_alt[op->src] : nullptr) : grad_buffers[op->src];
    COMIC_REQUIRE(!r[3] || gx, "cnn_backward_sched: op %d has no alternate input-gradient buffer", r[1]);
    const int xc = buf_channels[op->src], yc = buf_channels[op->dst];
    const bool x3 = (op->flags & COMIC_OP_X3) != 0;
    if (x3 && op->kind <= 1) {
      if constexpr (sizeof(T) == 2) {
        if (int rc = conv_backward_x3(op, buffers[op->src], xc, buffers[op->dst], (const float*)gy, yc, op->out_f32 ? yc : yc / 3,
                                      (float*)gx, op->kind == 1 ? 0 : xc / 3, weights + op->weight, grads + op->weight, batch,
                                      (char*)scratch + dz_off[r[1]], (int64_t)dz_bytes_of(op, batch, sizeof(T)), st, filters_ready, st_w))
          return rc;
      } else {
        COMIC_REQUIRE(false, "cnn_backward_sched: COMIC_OP_X3 is a bf16-plan layout");
      }
    } else if (op->kind <= 1) {
      void* dz = (char*)scratch + dz_off[r[1]];
      ConvMask mk{};
      const int p = fuses[r[1]];
      if (p >= 0) {
        COMIC_REQUIRE(!seen[p], "cnn_backward_sched: conv %d runs before its reader %d", p, r[1]);
        mk = ConvMask{buffers[ops[p].dst], buf_channels[ops[p].dst], ops[p].dst_coff, weights[ops[p].weight].scale,
                      (float*)((char*)scratch + part_off[p]), (char*)scratch + dz_off[p]};
        COMIC_REQUIRE(mk.y && mk.scale && grads[ops[p].weight].dbeta, "cnn_backward_sched: conv %d has no forward output / scale / d beta", p);
      }
      if (int rc = conv_backward<T>(op, buffers[op->src], xc, buffers[op->dst], gy, yc, gx, weights + op->weight,
                                    grads + op->weight, batch, dz, (int64_t)dz_bytes_of(op, batch, sizeof(T)), st,
                                    filters_ready, st_w, /*dz_ready=*/fused_by[r[1]] >= 0, p >= 0 ? &mk : nullptr,
                                    lane1_open ? &pend : nullptr))
        return rc;
    } else if (op->kind <= 4) {
      if (!gx) continue;
      if (int rc = pool_backward<T>(op, buffers[op->src], xc, gy, yc, gx, batch, st, x3 && op->kind <= 3)) return rc;
    } else {
      COMIC_REQUIRE(false, "cnn_backward_sched: unknown op kind %d", op->kind);
    }
  }
  COMIC_REQUIRE(!lane1_open, "cnn_backward_sched: the schedule ends inside a fork / join region");
  if (int rc = flush_wgrads()) return rc;
  for (int i = 0; i < n_ops; ++i)
    COMIC_REQUIRE(seen[i] || ops[i].kind == 5 || ops[i].kind == 6, "cnn_backward_sched: op %d is not in the schedule", i);
  {
    // the partial d beta rows of the fused convs -> d beta, on the weight-gradient lane (it waits for both chain lanes' launches
    // so far; the optimiser waits for this lane)
    DbetaFoldTable tb;
    tb.n = 0;
    auto flush = [&]() {
      if (tb.n) hipLaunchKernelGGL(dbeta_fold_kernel, dim3(tb.n), dim3(256), 0, st_w, tb);
      tb.n = 0;
    };
    bool any = false;
    for (int p = 0; p < n_ops; ++p) any = any || fused_by[p] >= 0;
    if (any) COMIC_REQUIRE(link_streams(s0, st_w), "cnn_backward_sched: fold of the fused d beta sums failed");
    for (int p = 0; p < n_ops; ++p) {
      if (fused_by[p] < 0) continue;
      if (tb.n == kFoldMax) flush();
      auto& en = tb.e[tb.n++];
      en.part = (const float*)((char*)scratch + part_off[p]);
      en.dbeta = grads[ops[p].weight].dbeta;
      en.C = ops[p].Cout;
      en.pad = 0;
    }
    flush();
  }
  COMIC_REQUIRE(link_streams(st_w, s0), "cnn_backward_sched: join of the weight-gradient lane failed");
  COMIC_LAUNCH_CHECK("cnn_backward_sched");
  return 0;
}

}  // namespace

extern "C" int comic_cnn_backward_sched(const comic_cnn_op* ops, int n_ops, const int32_t* sched, int n_sched,
                                        void* const* buffers, void* const* grad_buffers, void* const* grad_buffers_alt,
                                        const int32_t* buf_channels, const comic_conv_weight* weights,
                                        const comic_conv_grad* grads, int batch, int dtype, int filters_ready, void* scratch,
                                        int64_t scratch_bytes, void* stream0, void* stream1, void* wgrad_stream) {
  COMIC_REQUIRE(ops && sched && buffers && grad_buffers && buf_channels && weights && grads && scratch,
                "comic_cnn_backward_sched: null argument");
  if (dtype == COMIC_BF16)
    return cnn_backward_sched_impl<bf16_t>(ops, n_ops, sched, n_sched, buffers, grad_buffers, grad_buffers_alt, buf_channels,
                                           weights, grads, batch, scratch, scratch_bytes, (hipStream_t)stream0,
                                           (hipStream_t)stream1, (hipStream_t)wgrad_stream, (filters_ready & 1) != 0,
                                          (filters_ready & COMIC_CNN_BWD_NO_ACT_FUSION) != 0);
  if (dtype == COMIC_F32)
    return cnn_backward_sched_impl<float>(ops, n_ops, sched, n_sched, buffers, grad_buffers, grad_buffers_alt, buf_channels,
                                          weights, grads, batch, scratch, scratch_bytes, (hipStream_t)stream0,
                                          (hipStream_t)stream1, (hipStream_t)wgrad_stream, (filters_ready & 1) != 0,
                                          (filters_ready & COMIC_CNN_BWD_NO_ACT_FUSION) != 0);
  COMIC_REQUIRE(false, "unknown dtype %d", dtype);
  return 2;
}


extern "C" int64_t comic_cnn_backward_scratch_bytes(const comic_cnn_op* ops, int n_ops, int batch, int dtype, int lanes) {
  if (!ops) return -1;
  return backward_scratch_bytes(ops, n_ops, batch, dtype == COMIC_BF16 ? 2 : 4, lanes > 1);
}

extern "C" int comic_cnn_pack_x3_weights(const float* const* masters, void* const* outs, const int32_t* cout,
                                         const int32_t* taps, const int32_t* cin, int n, void* stream) {
  COMIC_REQUIRE(masters && outs && cout && taps && cin && n >= 0, "comic_cnn_pack_x3_weights: null argument");
  hipStream_t st = (hipStream_t)stream;
  PackX3Table tb;
  tb.n = 0;
  uint32_t blocks = 0;
  auto flush = [&]() {
    if (tb.n) hipLaunchKernelGGL(pack_x3_table_kernel, dim3(blocks), dim3(256), 0, st, tb);
    tb.n = 0;
    blocks = 0;
  };
  for (int i = 0; i < n; ++i) {
    COMIC_REQUIRE(masters[i] && outs[i] && cout[i] > 0 && cout[i] <= 65535 && cin[i] > 0 && cin[i] <= 65535 && taps[i] > 0 &&
                      taps[i] <= 65535, "comic_cnn_pack_x3_weights: bad entry %d", i);
    if (tb.n == kPackTableMax) flush();
    PackX3Entry& en = tb.e[tb.n++];
    en.master = masters[i];
    en.out = (bf16_t*)outs[i];
    en.first_block = blocks;
    en.cout = (uint16_t)cout[i]; en.cin = (uint16_t)cin[i]; en.taps = (uint16_t)taps[i]; en.pad = 0;
    const long K3 = 3L * taps[i] * cin[i];
    blocks += (uint32_t)cdiv64((long)cout[i] * ((K3 + 63) / 64 * 64), 256);
  }
  flush();
  COMIC_LAUNCH_CHECK("pack_x3_weights");
  return 0;
}

extern "C" int comic_cnn_pack_bwd_filters(const comic_cnn_op* ops, int n_ops, const comic_conv_grad* grads, int dtype,
                                          void* stream) {
  COMIC_REQUIRE(ops && grads, "comic_cnn_pack_bwd_filters: null argument");
  if (dtype == COMIC_BF16) return pack_bwd_filters_impl<bf16_t>(ops, n_ops, grads, (hipStream_t)stream);
  if (dtype == COMIC_F32) return pack_bwd_filters_impl<float>(ops, n_ops, grads, (hipStream_t)stream);
  COMIC_REQUIRE(false, "unknown dtype %d", dtype);
  return 2;
}

extern "C" int comic_cnn_backward(const comic_cnn_op* ops, int n_ops, void* const* buffers, void* const* grad_buffers,
                                  const int32_t* buf_channels, const comic_conv_weight* weights,
                                  const comic_conv_grad* grads, int batch, int dtype, int filters_ready, void* scratch,
                                  int64_t scratch_bytes, void* stream, void* wgrad_stream) {
  COMIC_REQUIRE(ops && buffers && grad_buffers && buf_channels && weights && grads && scratch,
                "comic_cnn_backward: null argument");
  hipStream_t st = (hipStream_t)stream, st_w = (hipStream_t)wgrad_stream;
  if (dtype == COMIC_BF16)
    return cnn_backward_impl<bf16_t>(ops, n_ops, buffers, grad_buffers, buf_channels, weights, grads, batch, scratch,
                                     scratch_bytes, st, filters_ready != 0, st_w);
  if (dtype == COMIC_F32)
    return cnn_backward_impl<float>(ops, n_ops, buffers, grad_buffers, buf_channels, weights, grads, batch, scratch,
                                    scratch_bytes, st, filters_ready != 0, st_w);
  COMIC_REQUIRE(false, "unknown dtype %d", dtype);
  return 2;
}

// fp32 master -> plan-dtype copy of a flat parameter block (the packed layouts are identical)
namespace {
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f32_to_bf16(in[i]);
}
__global__ void refold_bn_kernel(const float* __restrict__ beta, const float* __restrict__ mean,
                                 const float* __restrict__ scale, float* __restrict__ shift, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) shift[i] = beta[i] - mean[i] * scale[i];
}
}  // namespace

extern "C" int comic_cnn_refresh_weights(const float* master, void* plan_copy, int64_t n, const float* beta,
                                         const float* mean, const float* scale, float* shift, int64_t channels,
                                         void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (plan_copy && n > 0) {
    COMIC_REQUIRE(master, "cnn_refresh_weights: null master");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, master, (bf16_t*)plan_copy,
                       (long)n);
  }
  if (channels > 0) {
    COMIC_REQUIRE(beta && mean && scale && shift, "cnn_refresh_weights: null BN arrays");
    hipLaunchKernelGGL(refold_bn_kernel, dim3((unsigned)cdiv64(channels, 256)), dim3(256), 0, st, beta, mean, scale,
                       shift, (long)channels);
  }
  COMIC_LAUNCH_CHECK("cnn_refresh_weights");
  return 0;
}

#ifdef COMIC_STAMPS
extern "C" int comic_debug_read_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
