// Shared device helpers of the bf16 convolution kernels (conv.hip, conv_ws.hip): argument record, LDS-DMA /
// fragment-read primitives, the BatchNorm + ReLU store epilogue and the XCD-aware tile index.
#pragma once
#include <type_traits>

#include "common.h"

struct ConvArgs {
  const void* x;
  const void* w;
  const float* scale;
  const float* shift;
  void* y;
  int B, H, W, Cin, Cout, KH, KW, SH, SW, PT, PL, Ho, Wo;
  int x_cs, x_co, y_cs, y_co;  // channel stride / offset of the src and dst pixel
  int K, Kpad, M;
  int relu, out_f32;
  const void* zero;  // 16 zero bytes in device memory (source of padding / out-of-range DMA lanes)
  int blk0, tiles_m; // grouped launch: first flat workgroup id of this problem, its pixel-tile count
  int remap;         // 1: XCD-aware workgroup -> tile mapping (see xcd_tile_index); 2: grouped launch whose members
                     // read the SAME im2col matrix (the 1x1 convs of one Inception block): see shared_input_group
  int grp_nt;        // remap 2: out-channel tiles of all members together; blk0 = those of the members before this one
  int accum;         // 1: y += result (backward-data accumulation into a gradient buffer)
  // patch-resident kernel (conv_patch.inc): tile geometry, filled by apply_geometry()
  int p_TC, p_TR, p_ncol, p_PW, p_PXBp, p_CPP, p_CPPp, p_cmagic, p_NR, p_Hp, p_rowB, p_gapB;
  int member_kind;   // grouped launch: 0 conv tile, 1 pool + BN + ReLU (kind 7) work items
  int min_lds;       // host only: lower bound on the dynamic LDS of the launch (comic_cnn_op::min_lds)
  const void* w_frag; // host only: the weights in MFMA-fragment order (comic_conv_weight::w_frag), or null
  int x3;            // COMIC_OP_X3: channel stride between the [hi | lo | hi] regions of the bf16 destination (0: plain store)
  int x3_src;        // pools of a COMIC_OP_X3 plan: the same for the source buffer
  // Backward-data launch fused with the activation gradient of the conv that PRODUCED its input (one reader only): the result
  // g is the gradient at that conv's output y; the epilogue stores g * 1[y > 0] * mask_scale[c] (the gradient at its
  // pre-BatchNorm output, what act_grad_kernel would write) and adds the column sums of g * 1[y > 0] to mask_dbeta.
  const void* mask_y;       // forward output of the producer (plan dtype), rows of mask_cs channels, slice from mask_co; null: off
  const float* mask_scale;
  float* mask_dbeta;        // [kMaskCopies][Cout] partial sums (folded into d beta by dbeta_fold_kernel)
  int mask_cs, mask_co;
};

namespace {

constexpr int kMaskCopies = 32;   // accumulator copies per channel of a fused activation gradient (ConvArgs::mask_dbeta)

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
};
template <>
struct Elem<bf16_t> {
  static constexpr int EPC = 8;
};


#ifdef COMIC_STAMPS
__device__ unsigned long long g_stamps[16384 * 8];
#define STAMP(i) if (tid == 0) g_stamps[(blockIdx.x & 16383) * 8 + (i)] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int OFF>
__device__ __forceinline__ u32x4_t lds_read128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void dma16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

// Epilogue shared by the bf16 conv kernels: y = relu(acc * scale[n] + shift[n]) for a wave's TN x TM
// 16x16 accumulator tiles.  Lane (mcol = lane & 15, nq = (lane >> 4) * 4) holds 4 consecutive output
// channels n0..n0+3 of pixel mrow[j] (< 0: no such pixel).  Every scale / shift vector is loaded up
// front and the arithmetic is branch-free, so the only vector-memory wait in here is the one for those
// loads: with the loads inside the per-tile branches the compiler has to drain vmcnt(0) at the top of
// every tile, i.e. each store waited for the previous store's round trip.
// MASK: the launch is a backward-data conv fused with its producer's activation gradient (ConvArgs::mask_y).  A template
// parameter, not a runtime branch: the extra registers of that path pushed every forward kernel into spills otherwise.
template <int TN, int TM, bool MASK = false>
__device__ __forceinline__ void conv_store_tiles(const ConvArgs& a, f32x4_t (&acc)[TN][TM], const int nbase,
                                                 const int nq, const int (&mrow)[TM]) {
  float4 sc[TN], sh[TN];
  bool nv[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    nv[i] = nbase + i * 16 < a.Cout;                     // wave-uniform: Cout is a multiple of 16
    const int n0 = nv[i] ? nbase + i * 16 + nq : 0;
    sc[i] = a.scale ? *(const float4*)(a.scale + n0) : make_float4(1.f, 1.f, 1.f, 1.f);
    sh[i] = a.scale ? *(const float4*)(a.shift + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float lo = a.relu ? 0.f : -INFINITY;             // relu as one v_max per value
  const int esz = a.out_f32 ? 4 : 2;
  if constexpr (MASK) {        // fused activation gradient of the producer conv (see ConvArgs::mask_y)
    float4 bsc[TN], sum[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      bsc[i] = *(const float4*)(a.mask_scale + (nv[i] ? nbase + i * 16 + nq : 0));
      sum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const bool mok = mrow[j] >= 0;
      const size_t m = (size_t)(mok ? mrow[j] : 0);
      const unsigned char* yrow = (const unsigned char*)a.mask_y + (m * a.mask_cs + a.mask_co + nbase + nq) * esz;
      unsigned char* drow = (unsigned char*)a.y + (m * a.y_cs + a.y_co + nbase + nq) * esz;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const bool ok = nv[i] & mok;
        float yv[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          if (a.out_f32) {
            const float4 t = *(const float4*)(yrow + i * 64);
            yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w;
          } else {
            const uint2 t = *(const uint2*)(yrow + i * 32);
            yv[0] = __uint_as_float(t.x << 16); yv[1] = __uint_as_float(t.x & 0xFFFF0000u);
            yv[2] = __uint_as_float(t.y << 16); yv[3] = __uint_as_float(t.y & 0xFFFF0000u);
          }
        }
        const float g0 = yv[0] > 0.f ? acc[i][j][0] : 0.f, g1 = yv[1] > 0.f ? acc[i][j][1] : 0.f;
        const float g2 = yv[2] > 0.f ? acc[i][j][2] : 0.f, g3 = yv[3] > 0.f ? acc[i][j][3] : 0.f;
        sum[i].x += g0; sum[i].y += g1; sum[i].z += g2; sum[i].w += g3;
        if (ok) {
          if (a.out_f32) *(float4*)(drow + i * 64) = make_float4(g0 * bsc[i].x, g1 * bsc[i].y, g2 * bsc[i].z, g3 * bsc[i].w);
          else *(uint2*)(drow + i * 32) = make_uint2(pack_bf16x2(g0 * bsc[i].x, g1 * bsc[i].y), pack_bf16x2(g2 * bsc[i].z, g3 * bsc[i].w));
        }
      }
    }
    // the 16 lanes of a row group hold the same four channels of 16 different pixels
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        sum[i].x += __shfl_xor(sum[i].x, d);
        sum[i].y += __shfl_xor(sum[i].y, d);
        sum[i].z += __shfl_xor(sum[i].z, d);
        sum[i].w += __shfl_xor(sum[i].w, d);
      }
      if (nv[i] && (threadIdx.x & 15) == 0) {
        // (kMaskCopies accumulators per channel, picked by the workgroup: every workgroup of a launch finishes at about the
        // same time, and a thousand same-address atomics in a row cost 25 us per launch)
        float* db = a.mask_dbeta + (size_t)(blockIdx.x & (kMaskCopies - 1)) * a.Cout + nbase + i * 16 + nq;
        atomicAdd(db + 0, sum[i].x);
        atomicAdd(db + 1, sum[i].y);
        atomicAdd(db + 2, sum[i].z);
        atomicAdd(db + 3, sum[i].w);
      }
    }
    return;
  }
  if (a.x3 && !a.out_f32) {    // COMIC_OP_X3: v -> hi = bf16(v), lo = bf16(v - hi), stored as regions [hi | lo | hi]
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const bool mok = mrow[j] >= 0;
      unsigned char* yrow = (unsigned char*)a.y + ((size_t)(mok ? mrow[j] : 0) * a.y_cs + a.y_co + nbase + nq) * 2;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        float v[4] = {fmaf(acc[i][j][0], sc[i].x, sh[i].x), fmaf(acc[i][j][1], sc[i].y, sh[i].y),
                      fmaf(acc[i][j][2], sc[i].z, sh[i].z), fmaf(acc[i][j][3], sc[i].w, sh[i].w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
        const uint32_t h01 = pack_bf16x2(v[0], v[1]), h23 = pack_bf16x2(v[2], v[3]);
        const uint32_t l01 = pack_bf16x2(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xFFFF0000u));
        const uint32_t l23 = pack_bf16x2(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xFFFF0000u));
        if (nv[i] & mok) {
          unsigned char* yp = yrow + i * 32;
          *(uint2*)yp = make_uint2(h01, h23);
          *(uint2*)(yp + (size_t)a.x3 * 2) = make_uint2(l01, l23);
          *(uint2*)(yp + (size_t)a.x3 * 4) = make_uint2(h01, h23);
        }
      }
    }
    return;
  }
  // bf16, plain store, 16-byte aligned pixel rows: pairs of channel tiles are written as 16 B per lane.  A lane
  // holds channels [4q, 4q+4) of both tiles (q = lane >> 4); v_permlane16_swap exchanges the odd 16-lane rows of
  // tile i with the even rows of tile i+1, after which rows 0 / 2 hold channels [0,8) / [8,16) of tile i and rows
  // 1 / 3 the same of tile i+1: half the store instructions, 64 contiguous bytes per pixel instead of 4 x 8.
  if (!a.out_f32 && !a.accum && (((a.y_cs | a.y_co) & 7) == 0) && TN >= 2) {
    const int q = nq >> 2;
    const int choff = (q >> 1) * 8;                       // channel offset inside the lane's tile after the swap
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const bool mok = mrow[j] >= 0;
      unsigned char* ypix = (unsigned char*)a.y + ((size_t)(mok ? mrow[j] : 0) * a.y_cs + a.y_co + nbase) * 2;
#pragma unroll
      for (int i = 0; i + 1 < TN; i += 2) {
        uint32_t pk[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float v0 = fmaf(acc[i + t][j][0], sc[i + t].x, sh[i + t].x);
          float v1 = fmaf(acc[i + t][j][1], sc[i + t].y, sh[i + t].y);
          float v2 = fmaf(acc[i + t][j][2], sc[i + t].z, sh[i + t].z);
          float v3 = fmaf(acc[i + t][j][3], sc[i + t].w, sh[i + t].w);
          asm("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(v0), "s"(lo));
          asm("v_max_f32 %0, %1, %2" : "=v"(v1) : "v"(v1), "s"(lo));
          asm("v_max_f32 %0, %1, %2" : "=v"(v2) : "v"(v2), "s"(lo));
          asm("v_max_f32 %0, %1, %2" : "=v"(v3) : "v"(v3), "s"(lo));
          pk[t][0] = pack_bf16x2(v0, v1);
          pk[t][1] = pack_bf16x2(v2, v3);
        }
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        const int tsel = i + (q & 1);                     // the tile this lane stores
        const bool ok = mok & ((q & 1) ? nv[i + 1] : nv[i]);
        if (ok) *(uint4*)(ypix + (tsel * 16 + choff) * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
      if constexpr (TN & 1) {                             // odd tile count: the last one 8 bytes per lane
        constexpr int i = TN - 1;
        float v0 = fmaf(acc[i][j][0], sc[i].x, sh[i].x);
        float v1 = fmaf(acc[i][j][1], sc[i].y, sh[i].y);
        float v2 = fmaf(acc[i][j][2], sc[i].z, sh[i].z);
        float v3 = fmaf(acc[i][j][3], sc[i].w, sh[i].w);
        asm("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(v0), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v1) : "v"(v1), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v2) : "v"(v2), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v3) : "v"(v3), "s"(lo));
        if (nv[i] & mok) *(uint2*)(ypix + (i * 16 + nq) * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const bool mok = mrow[j] >= 0;
    // address of this lane's first channel in the pixel row; the tiles along n are 16 channels apart
    unsigned char* yrow = (unsigned char*)a.y + ((size_t)(mok ? mrow[j] : 0) * a.y_cs + a.y_co + nbase + nq) * esz;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      float v0 = fmaf(acc[i][j][0], sc[i].x, sh[i].x);
      float v1 = fmaf(acc[i][j][1], sc[i].y, sh[i].y);
      float v2 = fmaf(acc[i][j][2], sc[i].z, sh[i].z);
      float v3 = fmaf(acc[i][j][3], sc[i].w, sh[i].w);
      asm("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(v0), "s"(lo));
      asm("v_max_f32 %0, %1, %2" : "=v"(v1) : "v"(v1), "s"(lo));
      asm("v_max_f32 %0, %1, %2" : "=v"(v2) : "v"(v2), "s"(lo));
      asm("v_max_f32 %0, %1, %2" : "=v"(v3) : "v"(v3), "s"(lo));
      const bool ok = nv[i] & mok;
      if (a.out_f32) {
        float4* yp = (float4*)(yrow + i * 64);
        if (a.accum) {
          if (ok) {
            const float4 o = *yp;
            v0 += o.x; v1 += o.y; v2 += o.z; v3 += o.w;
          }
        }
        if (ok) *yp = make_float4(v0, v1, v2, v3);
      } else {
        uint2* yp = (uint2*)(yrow + i * 32);
        if (a.accum) {
          if (ok) {
            const uint2 o = *yp;
            v0 += __uint_as_float(o.x << 16); v1 += __uint_as_float(o.x & 0xFFFF0000u);
            v2 += __uint_as_float(o.y << 16); v3 += __uint_as_float(o.y & 0xFFFF0000u);
          }
        }
        if (ok) *yp = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
}

// Workgroups are dealt round-robin to the 8 XCDs (workgroup i -> XCD i % 8), each with a private
// L2.  xcd_tile_index turns the hardware id into a logical tile index such that every XCD owns one
// CONTIGUOUS range of logical tiles; with the out-channel tile as the fastest logical dimension,
// all out-channel tiles of a pixel tile (and its halo neighbours) run on the same XCD, so an
// activation row is pulled from the memory side into exactly one L2 instead of up to 8.
// The grid is padded to a multiple of 8; ids past `total` exit.
__device__ __forceinline__ int xcd_tile_index(int total) {
  const int per = gridDim.x >> 3;
  const int l = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  return l < total ? l : -1;
}

}  // namespace
