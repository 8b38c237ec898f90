// Weight-stationary 1x1 convolution group (bf16) for gfx950, optionally reading its input through a 3x3 / stride-2
// VALID max-pool ("pool-on-load").
//
// Replaces, for the thin 1x1 layers of InceptionV3 (common/nets/inception_v3.py): Conv2d_3b_1x1 behind MaxPool_3a_3x3
// (:111-114), the four Branch_*/Conv2d_0a_1x1 | Conv2d_0b_1x1 convolutions at the head of Mixed_5b/5c/5d (:141-199,
// Mixed_5b behind MaxPool_5a_3x3 :124), each Conv2D (no bias) -> FusedBatchNorm(inference) -> Relu under
// inception_arg_scope (common/nets/inception_utils.py:32-82).
//
// Why a kernel of its own: these layers have K = Cin <= 288 and, over the convolutions that share one input,
// N <= 256 output channels.  The im2col tiles of conv.hip re-fill the activation tile once per member and per
// 64..192-channel tile and spend most of a workgroup's life in prologue / epilogue (3-5 k-tiles): 250-340 TFLOP/s
// and 2-3x the HBM floor at 640 images.  Here
//   * the WHOLE weight matrix of all members lives in the registers of four matrix waves for the life of a
//     persistent workgroup (wave w owns NT 16-channel tiles x all K: KS * NT fragments of 4 VGPRs);
//   * four loader waves stream 64-pixel activation tiles global -> registers -> LDS (double buffered, the same
//     [row][128 B] XOR-swizzled image as conv.hip so fragment reads are conflict-free); with `pooled` the loader
//     forms each 16-byte chunk as the maximum over the 3x3 window of the un-pooled map, so the pooled tensor is
//     never written or re-read;
//   * the matrix waves run KS x (4 ds_read_b128 + 4*NT MFMA) per tile with no barrier inside, then the
//     BatchNorm + ReLU epilogue into each member's destination slice (bf16, or raw fp32 for the projection of a
//     pool branch); one s_barrier per tile hands the buffers over.
// Every activation byte crosses L2 -> CU once per N <= 256 channels (FLOP per fill byte = N instead of
// BM*BN/(BM+BN) ~ 50), so the kernels are bound by HBM: read of the (un-pooled) input + write of the outputs.
// The k order per accumulator is the im2col kernels' (k ascending in steps of 32), so results are bit-identical.
#include <algorithm>

#include "conv_common.h"
#include "conv_ws.h"

namespace {

constexpr int kWsRows = 64;                 // pixels per tile
constexpr int kWsTableBytes = 16 * 16 * 2 * 4;   // scale | shift of up to 256 concatenated channels

// Workgroup barrier without the fence of __syncthreads(): that fence drains vmcnt(0), i.e. the loaders' prefetched
// tile and the matrix waves' epilogue stores.  Only LDS traffic is handed over here.
#define WS_BARRIER()                                      \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

__device__ __forceinline__ uint32_t bf16x2_order(uint32_t v) {
  // sign-magnitude bf16 pair -> two's-complement-ordered int16 pair (an involution): x ^ ((x >> 15) & 0x7fff)
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 s = __builtin_bit_cast(s16x2, v);
  const s16x2 m = (s >> 15) & (short)0x7fff;
  return __builtin_bit_cast(uint32_t, (s16x2)(s ^ m));
}
__device__ __forceinline__ uint32_t max_i16x2(uint32_t a, uint32_t b) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

template <int KS, int NT>
__global__ __launch_bounds__(512) void conv_ws_kernel(ComicWsArgs a) {
  constexpr int KT = (KS + 1) / 2;          // 64-deep k-tiles of the LDS image
  constexpr int ABYTES = KT * kWsRows * 128;
  constexpr int CPR = 4 * KS;               // 16-byte chunks per activation row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* table = (float*)smem;              // [n_tiles*16] scale, then [n_tiles*16] shift
  unsigned char* abuf = smem + kWsTableBytes;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile_w = a.tiles_m;
  // a workgroup owns a CONTIGUOUS range of pixel tiles: consecutive tiles share input rows of their pooling windows
  // (and DRAM pages), which then stay in this CU's L2 instead of being fetched again by another XCD
  const int per_wg = (ntile_w + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile_begin = blockIdx.x * per_wg, tile_end = min(ntile_w, tile_begin + per_wg);

  if (wave >= 4) {
    // ---------------------------------------------------------------- loader waves -------------
    const int ltid = tid - 256;
    int row[KS], ch[KS];
    uint32_t loff[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int c = ltid + 256 * i;
      row[i] = c / CPR;
      ch[i] = c - row[i] * CPR;
      loff[i] = (ch[i] >> 3) * (kWsRows * 128) + row[i] * 128 + ((((ch[i] & 7) ^ ((row[i] >> 1) & 7))) << 4);
    }
    // scale / shift table (identity for raw members)
    for (int c = ltid; c < a.n_tiles * 16; c += 256) {
      int p = 0;
      for (int i = 1; i < a.n_members; ++i)
        if ((c >> 4) >= a.m[i].tile0) p = i;
      const int lc = c - a.m[p].tile0 * 16;
      table[c] = a.m[p].scale ? a.m[p].scale[lc] : 1.f;
      table[a.n_tiles * 16 + c] = a.m[p].scale ? a.m[p].shift[lc] : 0.f;
    }
    const int HoWo = a.Ho * a.Wo;
    auto load_plain = [&](int tile, uint4 (&v)[KS]) {
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int m = tile * kWsRows + row[i];
        v[i] = make_uint4(0, 0, 0, 0);
        if (m < a.M) v[i] = *(const uint4*)(a.x + ((size_t)m * a.x_cs + a.x_co + ch[i] * 8));
      }
    };
    auto store_tile = [&](int buf, const uint4 (&v)[KS]) {
#pragma unroll
      for (int i = 0; i < KS; ++i) *(uint4*)(abuf + buf * ABYTES + loff[i]) = v[i];
    };
    // pooled: chunk i of the tile = max over the 3x3 window (9 loads); two chunks' loads are kept in flight
    auto window_base = [&](int tile, int i, bool& ok) -> const bf16_t* {
      const int m = tile * kWsRows + row[i];
      ok = m < a.M;
      const int mm = ok ? m : 0;
      const int b = mm / HoWo;
      const int r = mm - b * HoWo;
      const int ho = r / a.Wo;
      const int wo = r - ho * a.Wo;
      return a.x + ((size_t)((b * a.H + 2 * ho) * a.W + 2 * wo) * a.x_cs + a.x_co + ch[i] * 8);
    };
    auto load_window = [&](const bf16_t* base, uint4 (&t)[9]) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) t[dy * 3 + dx] = *(const uint4*)(base + (size_t)(dy * a.W + dx) * a.x_cs);
    };
    auto reduce_window = [&](const uint4 (&t)[9]) {
      uint32_t r[4] = {bf16x2_order(t[0].x), bf16x2_order(t[0].y), bf16x2_order(t[0].z), bf16x2_order(t[0].w)};
#pragma unroll
      for (int k = 1; k < 9; ++k) {
        r[0] = max_i16x2(r[0], bf16x2_order(t[k].x));
        r[1] = max_i16x2(r[1], bf16x2_order(t[k].y));
        r[2] = max_i16x2(r[2], bf16x2_order(t[k].z));
        r[3] = max_i16x2(r[3], bf16x2_order(t[k].w));
      }
      return make_uint4(bf16x2_order(r[0]), bf16x2_order(r[1]), bf16x2_order(r[2]), bf16x2_order(r[3]));
    };
    // three windows' loads (27 x 16 bytes a thread) stay in flight: with two, the launch behind MaxPool_5a moved 1.33 GB in
    // 437 us, request-latency bound at 73 KB in flight per CU (the loader waves own registers the matrix waves' weight
    // fragments do not need)
    // (the pooled sources have Cin 64 / 192: KS 2 / 6; wider instantiations keep two windows -- their matrix waves hold more
    // weight fragments and the allocation is per kernel)
    constexpr int NWIN = KS <= 6 ? 3 : 2;
    auto pooled_tile = [&](int tile, int buf) {
      uint4 tw[NWIN][9];
      bool okw[NWIN];
#pragma unroll
      for (int i = 0; i < NWIN; ++i)
        if (i < KS) load_window(window_base(tile, i, okw[i]), tw[i]);
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const uint4 r = reduce_window(tw[i % NWIN]);
        *(uint4*)(abuf + buf * ABYTES + loff[i]) = okw[i % NWIN] ? r : make_uint4(0, 0, 0, 0);
        if (i + NWIN < KS) load_window(window_base(tile, i + NWIN, okw[i % NWIN]), tw[i % NWIN]);
      }
    };

    int tile = tile_begin;
    if (a.pooled) {
      if (tile < tile_end) pooled_tile(tile, 0);
      WS_BARRIER();
      int it = 0;
      for (; tile < tile_end; ++tile, ++it) {
        const int nxt = tile + 1;
        if (nxt < tile_end) pooled_tile(nxt, (it + 1) & 1);
        WS_BARRIER();
      }
    } else {
      uint4 cur[KS], nx[KS];
#pragma unroll
      for (int i = 0; i < KS; ++i) cur[i] = nx[i] = make_uint4(0, 0, 0, 0);
      if (tile < tile_end) {
        load_plain(tile, cur);
        store_tile(0, cur);
      }
      if (tile + 1 < tile_end) load_plain(tile + 1, cur);
      WS_BARRIER();
      int it = 0;
      for (; tile < tile_end; ++tile, ++it) {
        const int nxt = tile + 1;                  // its loads are in `cur`
        const int nxt2 = nxt + 1;
        if (nxt2 < tile_end) load_plain(nxt2, nx);
        if (nxt < tile_end) store_tile((it + 1) & 1, cur);
#pragma unroll
        for (int i = 0; i < KS; ++i) cur[i] = nx[i];
        WS_BARRIER();
      }
    }
    return;
  }

  // ------------------------------------------------------------------ matrix waves ---------------
  const int fr = lane & 15, fg = lane >> 4;
  // this wave's 16-channel tiles: NT consecutive tiles of the concatenated N
  bf16x8_t wreg[KS][NT];
  int mem_of[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = wave * NT + i;
    int p = -1;
    if (nt < a.n_tiles) {
      p = 0;
      for (int q = 1; q < a.n_members; ++q)
        if (nt >= a.m[q].tile0) p = q;
    }
    mem_of[i] = p;
    const bf16_t* wp = p >= 0 ? a.m[p].w + (size_t)((nt - a.m[p].tile0) * 16 + fr) * a.Kpad + fg * 8 : nullptr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (p >= 0) v = *(const uint4*)(wp + ks * 32);
      wreg[ks][i] = __builtin_bit_cast(bf16x8_t, v);
    }
  }
  const uint32_t sw = (fr >> 1) & 7;
  uint32_t xoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) xoff[h] = fr * 128 + ((((h * 4 + fg) ^ sw) & 7) << 4);

  WS_BARRIER();      // tile 0 + the scale / shift table are in LDS
  int it = 0;
  for (int tile = tile_begin; tile < tile_end; ++tile, ++it) {
    const unsigned char* buf = abuf + (it & 1) * ABYTES;
    f32x4_t acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xf[j] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(buf + (ks >> 1) * (kWsRows * 128) + j * 2048 + xoff[ks & 1]));
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][i], xf[j], acc[i][j], 0, 0, 0);
    }
    // epilogue: lane holds channels fg*4 .. +3 of pixel (tile*64 + j*16 + fr) for each of its tiles
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int p = mem_of[i];
      if (p < 0) continue;
      const int nt = wave * NT + i;
      const float4 sc = *(const float4*)(table + nt * 16 + fg * 4);
      const float4 sh = *(const float4*)(table + a.n_tiles * 16 + nt * 16 + fg * 4);
      const float lo = a.m[p].relu ? 0.f : -INFINITY;
      const int cho = a.m[p].y_co + (nt - a.m[p].tile0) * 16 + fg * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = tile * kWsRows + j * 16 + fr;
        const float v0 = fmaxf(fmaf(acc[i][j][0], sc.x, sh.x), lo), v1 = fmaxf(fmaf(acc[i][j][1], sc.y, sh.y), lo);
        const float v2 = fmaxf(fmaf(acc[i][j][2], sc.z, sh.z), lo), v3 = fmaxf(fmaf(acc[i][j][3], sc.w, sh.w), lo);
        if (m < a.M) {
          const size_t off = (size_t)m * a.m[p].y_cs + cho;
          if (a.m[p].out_f32)
            *(float4*)((float*)a.m[p].y + off) = make_float4(v0, v1, v2, v3);
          else
            *(uint2*)((bf16_t*)a.m[p].y + off) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      }
    }
    WS_BARRIER();
  }
}

template <int KS, int NT>
int launch_ws(const ComicWsArgs& a, hipStream_t st) {
  constexpr int lds = kWsTableBytes + 2 * ((KS + 1) / 2) * kWsRows * 128;
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_ws_kernel<KS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      comic_set_error("conv_ws: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = std::min(a.tiles_m, cus);
  hipLaunchKernelGGL((conv_ws_kernel<KS, NT>), dim3(grid), dim3(512), lds, st, a);
  return 0;
}

}  // namespace

bool comic_ws_supported(int Cin, int n_tiles) {
  const int ks = Cin / 32, nt = (n_tiles + 3) / 4;
  if (Cin % 32 != 0 || n_tiles < 1 || n_tiles > 16) return false;
  return (ks == 2 && nt <= 2) || ((ks == 6 || ks == 8 || ks == 9) && nt <= 4);
}

int comic_ws_launch(const ComicWsArgs& a, hipStream_t st) {
  const int ks = a.Cin / 32, nt = (a.n_tiles + 3) / 4;
  if (!comic_ws_supported(a.Cin, a.n_tiles)) {
    comic_set_error("conv_ws: unsupported shape (Cin %d, %d channel tiles)", a.Cin, a.n_tiles);
    return 2;
  }
  if (ks == 2) return launch_ws<2, 2>(a, st);
  if (ks == 6) return launch_ws<6, 4>(a, st);
  if (ks == 8) return launch_ws<8, 4>(a, st);
  return launch_ws<9, 4>(a, st);
}
