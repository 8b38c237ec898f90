// Image-resident implicit GEMM for the stride-1 SAME convolutions on SMALL feature maps (bf16): the 1x7 / 7x1 chains
// of Mixed_6 (12x12), the 3x3 convs of Mixed_5 (25x25) and the 3x3 / 1x3 / 3x1 convs of Mixed_7 (5x5)
// (common/nets/inception_v3.py:124-413).
//
// What bounds the im2col kernels (conv.hip) on these layers is the L2 -> LDS fill: every k-tile moves (BM + BN) * 128
// bytes for BM * BN / 32 MFMA cycles and the fill path delivers about 30 B/clk/CU (the committed
// counters show the matrix pipe 29-45 % busy).  Here a workgroup keeps G WHOLE images resident in the LDS:
//   * pixels    [P = G*H*W][Cin] bf16, pixel stride padded to an odd multiple of 32 B (the 16 pixels x 2 chunks of a
//               ds_read_b128 lane group fall into 16 distinct 16-byte bank groups), NO halo: a filter tap that leaves
//               the image reads a zero page instead (one validity bit per pixel and tap, computed once);
//   * weights   never touch the LDS: they are pre-packed in MFMA-fragment order ([16-channel tile][k32-step][lane][8])
//               so that a wave's A operand of one k-step is ONE contiguous KiB, and stream global -> VGPR one step ahead
//               (buffer loads with a scalar offset: no address arithmetic on the vector unit);
//   * no barrier in the main loop: after the patch fill the WM x WN waves are independent; two waves per SIMD cover
//     each other's LDS latency.
// Per 32-deep k-step a wave reads TM pixel fragments for TM * TN MFMAs; a k-step lies inside one filter tap
// (Cin % 32 == 0), so the tap's per-lane addresses are formed once per tap and the channel steps are ds_read immediates.
// Fill bytes per MFMA cycle fall from (BM + BN) * 128 / (BM * BN / 32) to the weight stream alone, shared through L1/L2
// by the waves of a CU: 14 B/clk/CU at 288 pixels x 192 channels.
//
// Same operands and the same k order per accumulator as conv_igemm_dma_body: results are bit-identical.
#include <algorithm>

#include "conv_img.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned img_u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t img_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

// PD: k-steps the weight fragments are requested ahead of their use.  Measured (round 5, 1280 images): PD 4 / 6 and a second
// pixel-fragment set read one step ahead are no faster where they fit the registers and slower where they spill; with ALL
// weight loads or ALL pixel reads removed the kernel gains 4-10 %: the k loop was never the problem.  What was: one
// workgroup of two images per CU (the LDS a pair needs) leaves the patch fill and the epilogue stores of every workgroup
// exposed -- 27 us per workgroup for 11 us of MFMA work at 12x12 128 -> 192.  WM = 1 with ONE image per workgroup (four
// waves, half the LDS) puts two workgroups on a CU: one's fill and stores run under the other's MFMAs, +15-25 %.
// 160 output channels run on the 192-channel geometry (the tiles beyond Cout load zeros -- the buffer descriptor ends at Cout --
// and are not stored): 17 % of the MFMAs wasted, still +16 % over 10 waves x 2 images; skipping the dead tiles' loads and
// MFMAs with wave-uniform branches measured 15 % SLOWER than issuing them.
// zeros behind the pixels: a masked lane reads at (its live address mod 256) + the channel step's immediate offset
constexpr int img_zero_bytes(int cs) { return (256 + cs * 64 + 1023) / 1024 * 1024; }

// Per-lane state of a wave's TM pixel tiles under one filter shape: LDS address of the lane's pixel (+ its 16-byte k group),
// the taps (kh, kw) that stay inside the image as a bit mask (bit kh*KW + kw), and the pixel's row in the NHWC destination.
template <int TM>
__device__ __forceinline__ void img_pixel_state(int lane, int wm, int P, int H, int W, int PXBp, int KH, int KW, int PT, int PL,
                                                int img0, uint32_t (&pixaddr)[TM], uint32_t (&mask)[TM], int (&mrow)[TM]) {
  const int fr = lane & 15, fg = lane >> 4;
  const int HW = H * W;
  const float rHW = 1.0f / (float)HW, rW = 1.0f / (float)W;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = (wm * TM + j) * 16 + fr;
    const bool pv = p < P;
    const int pp = pv ? p : 0;
    const int img = (int)(((float)pp + 0.5f) * rHW);        // exact for these sizes (pp < 2^16)
    const int r = pp - img * HW;
    const int h = (int)(((float)r + 0.5f) * rW), w = r - h * W;
    pixaddr[j] = (uint32_t)(pp * PXBp + fg * 16);
    // taps (kh, kw) inside the image: kh in [klo, khi], kw in [wlo, whi]; bit kh*KW + kw
    const int klo = max(0, PT - h), khi = min(KH - 1, H - 1 - h + PT);
    const int wlo = max(0, PL - w), whi = min(KW - 1, W - 1 - w + PL);
    const uint32_t rowbits = ((2u << whi) - 1u) & ~((1u << wlo) - 1u);
    uint32_t mk = 0;
    for (int kh = klo; kh <= khi; ++kh) mk |= rowbits << (kh * KW);
    mask[j] = pv ? mk : 0u;
    mrow[j] = pv ? img0 * HW + pp : -1;
  }
}

// The k loop of one convolution over the resident pixels: acc[i][j] = sum over (tap, channel step) of W-fragment x pixel
// fragment, k order (kh, kw, c) as in conv_igemm_dma_body.  SYNC: a workgroup barrier between the first weight requests and
// the first LDS read (the patch fill of conv_img_kernel ends there).
template <int TM, int TN, int CS, int PD, bool SYNC>
__device__ __forceinline__ void img_kloop(const unsigned char* smem, uint32_t zoff, const uint32_t (&pixaddr)[TM],
                                          const uint32_t (&mask)[TM], const bf16_t* wf, int Cout, int KS32, int KH, int KW,
                                          int PT, int PL, int W, int PXBp, int wn, int lane, f32x4_t (&acc)[TN][TM]) {
  const int taps = KH * KW;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- weight stream: tile (wn*TN + i), step s at byte ((tile * KS32 + s) * 64 + lane) * 16 ---------------------------
  const __amdgpu_buffer_rsrc_t wr = img_rsrc(wf, (uint32_t)(Cout / 16) * (uint32_t)KS32 * 1024u);
  const int wv = lane * 16;
  const int tile_stride = KS32 * 1024;
  const int wbase = wn * TN * tile_stride;
  const int nsteps = taps * CS;
  // software pipeline of depth PD: wq[0] = this step's fragments, wq[d] = those of step s + d, wq[PD] = the loads issued now.
  // The sched_barrier pins the loads at the top of the step (left alone, the scheduler sinks them behind the last MFMA
  // that reads the register they overwrite and their latency is exposed).
  img_u32x4 wq[PD + 1][TN];
#pragma unroll
  for (int d = 0; d < PD; ++d)
#pragma unroll
    for (int i = 0; i < TN; ++i)
      wq[d][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, wbase + min(d, nsteps - 1) * 1024 + i * tile_stride, 0);

  if constexpr (SYNC) __syncthreads();   // the patch is complete (the only barrier of conv_img_kernel)

  auto tap_addrs = [&](int t, int kh, int kw, uint32_t (&ad)[TM]) {
    const int tapoff = ((kh - PT) * W + (kw - PL)) * PXBp;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      // a tap outside the image reads zeros AT THE BANKS ITS PIXEL WOULD HAVE USED (the zero page is a multiple of 256 bytes
      // from the base): the lane groups of a ds_read_b128 stay conflict-free whatever the mix of inside / outside lanes.
      // (One fixed zero address per quarter wave collided with a live lane's bank in most groups of the 1x7 / 7x1 edge taps:
      // SQ_LDS_BANK_CONFLICT 27-39 % of the LDS cycles in profiles/r04_cnn_mfma_counters_1280.json.)
      const uint32_t live = pixaddr[j] + (uint32_t)tapoff;
      ad[j] = ((mask[j] >> t) & 1u) ? live : zoff + (live & 255u);
    }
  };
  int kh = 0, kw = 0, s = 0;
  uint32_t addr[TM];
  uint4 xf[TM];
  tap_addrs(0, 0, 0, addr);
  for (int t = 0; t < taps; ++t) {
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
      const int so = wbase + min(s + PD, nsteps - 1) * 1024;      // the last prefetches re-read the last step
      ++s;
#pragma unroll
      for (int i = 0; i < TN; ++i)
        wq[PD][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wv, so + i * tile_stride, 0);
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const uint4*)(smem + addr[j] + cs * 64);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wq[0][i]),
                                                              __builtin_bit_cast(bf16x8_t, xf[j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int i = 0; i < TN; ++i) wq[d][i] = wq[d + 1][i];
    }
    if (++kw == KW) {
      kw = 0;
      ++kh;
    }
    if (t + 1 < taps) tap_addrs(t + 1, kh, kw, addr);
  }
}

template <int TM, int TN, int WM, int WN, int CS, int PD = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 ? 2 : WM * WN <= 5 ? 3 : 1)) void conv_img_kernel(const ComicImgArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int CPP = CS * 4;            // 16-byte chunks per pixel (Cin = 32 * CS)
  constexpr int PMAX = TM * WM * 16;     // pixel slots of the workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mi = blockIdx.x % a.n_members, grp = blockIdx.x / a.n_members;
  const ComicImgMember& m = a.m[mi];
  const int HW = a.H * a.W, W = a.W, H = a.H;
  const int img0 = grp * a.G;
  const int P = min(a.G, a.B - img0) * HW;            // resident pixels of this workgroup
  const int PXBp = a.PXBp;
  const uint32_t zoff = (PMAX * PXBp + 255) & ~255;    // zeros behind the pixels, 256-byte aligned

  // ---- patch fill: the G images are consecutive pixels of the NHWC source; every load of a thread is in flight before
  // its first LDS write (one memory latency for the whole patch, not one per pass) ----------------------------------------
  {
    constexpr int U = (PMAX * CPP + NT - 1) / NT;
    const bf16_t* __restrict__ xg = m.x + (size_t)img0 * HW * m.x_cs + m.x_co;
    const int x_cs = m.x_cs;
    const int total = P * CPP;
    uint4 v[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * NT + tid;
      const bool ok = idx < total;
      const int p = ok ? idx / CPP : 0;
      const int c = ok ? idx - p * CPP : 0;
      dst[u] = ok ? p * PXBp + c * 16 : -1;
      v[u] = *(const uint4*)(xg + (size_t)p * x_cs + c * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) *(uint4*)(smem + dst[u]) = v[u];
    for (int z = tid; z < img_zero_bytes(CS) / 16; z += NT) *(uint4*)(smem + zoff + z * 16) = make_uint4(0u, 0u, 0u, 0u);
  }

  // ---- per-lane pixel state, k loop (img_pixel_state / img_kloop: shared with conv_img_chain_kernel) --------------------
  const int fg = lane >> 4;
  uint32_t pixaddr[TM], mask[TM];
  int mrow[TM];
  img_pixel_state<TM>(lane, wm, P, a.H, a.W, PXBp, m.KH, m.KW, m.PT, m.PL, img0, pixaddr, mask, mrow);
  f32x4_t acc[TN][TM];
  img_kloop<TM, TN, CS, PD, true>(smem, zoff, pixaddr, mask, m.wf, a.Cout, a.KS32, m.KH, m.KW, m.PT, m.PL, a.W, PXBp, wn, lane, acc);

  // ---- epilogue: BatchNorm + ReLU + store, shared with the other bf16 conv kernels ----------------------------------
  ConvArgs ca;
  ca.scale = m.scale; ca.shift = m.shift; ca.y = m.y; ca.y_cs = m.y_cs; ca.y_co = m.y_co; ca.Cout = a.Cout;
  ca.relu = m.relu; ca.out_f32 = m.out_f32; ca.accum = 0; ca.x3 = 0; ca.x3_src = 0;   // (x3 plans carry no fragment-order weights: never here)
  ca.mask_y = nullptr;     // (no fused activation gradient: a forward kernel)
  conv_store_tiles<TN, TM>(ca, acc, wn * TN * 16, fg * 4, mrow);
}

// ---- Branch chains of Mixed_6b-e (common/nets/inception_v3.py:262-345): 1x7 -> 7x1 and 7x1 -> 1x7 -> 7x1 -> 1x7 ------------
// conv_img_kernel spends 11 of a workgroup's 27 us in MFMAs at 12x12 128 -> 192; the rest is the patch fill in front of
// the loop and the epilogue stores behind it, and between two convs of a branch those are a write of the 12x12xC map to
// HBM and a read of the same map.  Here ONE workgroup (one image, four waves) runs every 7-tap conv of a branch: after a
// conv's k loop the complete output map sits in the accumulators of the four waves (wave wn: channels [wn*TN*16, +TN*16) of
// all 144 pixels), so BatchNorm + ReLU + the bf16 rounding of the unfused store are applied in registers and the map is
// written IN PLACE over the patch it was computed from (barrier - write - barrier): one patch, 41-61 KB, still two
// workgroups per CU.  Only the first conv's input is filled from HBM and only the last conv's output is stored; weights
// stream global -> VGPR as before.  Same operands, same k order, same epilogue arithmetic per value: bit-identical to the
// chain of single launches.
// TNI: 16-channel tiles per wave of the INNER convs (Cout = Cin: 2 at 128 channels, 3 at 160 -- ragged, as cfg 4 -- and
// 192); the last conv of a chain has 192 output channels (three tiles per wave).
template <int TM, int TNI, int CS, int PD = 2>
__global__ __launch_bounds__(256, 2) void conv_img_chain_kernel(const ComicChainArgs a) {
  constexpr int NT = 256, WN = 4;
  constexpr int CPP = CS * 4;
  constexpr int PMAX = TM * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  // member 0 holds the longer chain: all of its workgroups are dispatched first
  const int mi = (int)blockIdx.x >= a.B ? 1 : 0;
  const int img0 = (int)blockIdx.x - mi * a.B;
  const ComicChainMember& m = a.m[mi];
  const int HW = a.H * a.W;
  const int P = HW;
  const int PXBp = a.PXBp;
  const uint32_t zoff = (PMAX * PXBp + 255) & ~255;

  {
    constexpr int U = (PMAX * CPP + NT - 1) / NT;
    const bf16_t* __restrict__ xg = m.x + (size_t)img0 * HW * m.x_cs + m.x_co;
    const int x_cs = m.x_cs;
    const int total = P * CPP;
    uint4 v[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * NT + tid;
      const bool ok = idx < total;
      const int p = ok ? idx / CPP : 0;
      const int c = ok ? idx - p * CPP : 0;
      dst[u] = ok ? p * PXBp + c * 16 : -1;
      v[u] = *(const uint4*)(xg + (size_t)p * x_cs + c * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) *(uint4*)(smem + dst[u]) = v[u];
    for (int z = tid; z < img_zero_bytes(CS) / 16; z += NT) *(uint4*)(smem + zoff + z * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  uint32_t pixaddr[TM], mask[TM];
  int mrow[TM];
  const int nc = m.n_convs;
  for (int ci = 0; ci + 1 < nc; ++ci) {
    const ComicChainConv& c = m.c[ci];
    img_pixel_state<TM>(lane, 0, P, a.H, a.W, PXBp, c.KH, c.KW, c.PT, c.PL, img0, pixaddr, mask, mrow);
    f32x4_t acc[TNI][TM];
    img_kloop<TM, TNI, CS, PD, false>(smem, zoff, pixaddr, mask, c.wf, a.Cin, c.KS32, c.KH, c.KW, c.PT, c.PL, a.W, PXBp, wn, lane, acc);
    // BatchNorm + ReLU + bf16 as conv_store_tiles computes them, written over the patch: lane (fr, fg) holds channels
    // [n0 + 4 fg, + 4) of pixel 16 j + fr
    float4 sc[TNI], sh[TNI];
    bool nv[TNI];
#pragma unroll
    for (int i = 0; i < TNI; ++i) {
      nv[i] = (wn * TNI + i) * 16 < a.Cin;                    // wave-uniform
      const int n0 = nv[i] ? (wn * TNI + i) * 16 + fg * 4 : 0;
      sc[i] = *(const float4*)(c.scale + n0);
      sh[i] = *(const float4*)(c.shift + n0);
    }
    const float lo = c.relu ? 0.f : -INFINITY;
    __syncthreads();                    // every wave has read its last pixel fragment of this conv
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      unsigned char* prow = smem + (j * 16 + fr) * PXBp + (wn * TNI * 16 + fg * 4) * 2;
#pragma unroll
      for (int i = 0; i < TNI; ++i) {
        float v0 = fmaf(acc[i][j][0], sc[i].x, sh[i].x);
        float v1 = fmaf(acc[i][j][1], sc[i].y, sh[i].y);
        float v2 = fmaf(acc[i][j][2], sc[i].z, sh[i].z);
        float v3 = fmaf(acc[i][j][3], sc[i].w, sh[i].w);
        asm("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(v0), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v1) : "v"(v1), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v2) : "v"(v2), "s"(lo));
        asm("v_max_f32 %0, %1, %2" : "=v"(v3) : "v"(v3), "s"(lo));
        const uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        if (nv[i]) *(uint2*)(prow + i * 32) = pk;
        // COMIC_OP_CHAIN_KEEP (trainable plans: the backward reads every conv's output): the same bits go to the conv's own
        // destination as well -- a store that nothing of this launch waits for
        if (c.keep && nv[i])
          *(uint2*)((unsigned char*)c.keep + ((size_t)(img0 * HW + j * 16 + fr) * c.keep_cs + c.keep_co + (wn * TNI + i) * 16 + fg * 4) * 2) = pk;
      }
    }
    __syncthreads();                    // the next conv's input is complete
  }
  {
    const ComicChainConv& c = m.c[nc - 1];
    img_pixel_state<TM>(lane, 0, P, a.H, a.W, PXBp, c.KH, c.KW, c.PT, c.PL, img0, pixaddr, mask, mrow);
    f32x4_t acc[3][TM];
    img_kloop<TM, 3, CS, PD, false>(smem, zoff, pixaddr, mask, c.wf, c.Cout, c.KS32, c.KH, c.KW, c.PT, c.PL, a.W, PXBp, wn, lane, acc);
    ConvArgs ca;
    ca.scale = c.scale; ca.shift = c.shift; ca.y = m.y; ca.y_cs = m.y_cs; ca.y_co = m.y_co; ca.Cout = c.Cout;
    ca.relu = c.relu; ca.out_f32 = m.out_f32; ca.accum = 0; ca.x3 = 0; ca.x3_src = 0;
    ca.mask_y = nullptr;
    conv_store_tiles<3, TM>(ca, acc, wn * 3 * 16, fg * 4, mrow);
  }
}

// [Cout][Kpad] bf16 -> fragment order, for every conv weight of a flat plan buffer in one launch.  table: per weight
// {element offset, Cout, Kpad} (int64 x 3), sorted by offset; stem / ineligible records carry Kpad 0 and are copied.
__global__ void pack_frag_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const long* __restrict__ table,
                                 int n, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte chunk per thread
  if (idx * 8 >= total) return;
  const long e0 = idx * 8;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[3 * mid] <= e0) lo = mid; else hi = mid - 1;
  }
  const long off = table[3 * lo], cout = table[3 * lo + 1], kpad = table[3 * lo + 2];
  const long local = e0 - off;
  long s_el = e0;                                                   // source element of this chunk (alignment gaps: copied)
  if (kpad > 0 && local < cout * kpad) {
    const long ks32 = kpad / 32;
    const long lane = (local / 8) % 64, s = (local / 512) % ks32, nt = local / (512 * ks32);
    s_el = off + (nt * 16 + (lane & 15)) * kpad + s * 32 + (lane >> 4) * 8;
  }
  *(uint4*)(dst + e0) = *(const uint4*)(src + s_el);
}

struct ImgCfg { int HW, Cin, Cout, G; };
// instantiations: (TM, TN, WM, WN, CS)
constexpr ImgCfg kCfg[] = {
    {144, 128, 192, 1}, {144, 160, 192, 1}, {144, 192, 192, 1},      // 0-2: 9,3,1,4,{4,5,6}
    {144, 128, 128, 1},                                                // 3:   9,2,1,4,4
    {144, 160, 160, 1},                                                // 4:   9,3,1,4,5 on the 192-channel geometry (ragged)
    {625, 64, 96, 1},  {625, 96, 96, 1},                               // 5-6: 10,3,4,2,{2,3}
    {25, 384, 384, 6}, {25, 448, 384, 6},                              // 7-8: 5,6,2,4,{12,14}
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <int TM, int TN, int WM, int WN, int CS, int PD = 2>
int launch_img(const ComicImgArgs& a, hipStream_t st) {
  const int lds = ((TM * WM * 16 * a.PXBp + 255) & ~255) + img_zero_bytes(CS);
  if (lds > 160 * 1024) {
    comic_set_error("conv_img: %d bytes of LDS", lds);
    return 2;
  }
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_img_kernel<TM, TN, WM, WN, CS, PD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      comic_set_error("conv_img: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_img_kernel<TM, TN, WM, WN, CS, PD>), dim3(a.groups * a.n_members), dim3(64 * WM * WN), lds, st, a);
  return 0;
}

template <int TNI, int CS>
int launch_chain(const ComicChainArgs& a, hipStream_t st) {
  constexpr int TM = 9;
  const int lds = ((TM * 16 * a.PXBp + 255) & ~255) + img_zero_bytes(CS);
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_img_chain_kernel<TM, TNI, CS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      comic_set_error("conv_img_chain: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_img_chain_kernel<TM, TNI, CS>), dim3(a.B * a.n_members), dim3(256), lds, st, a);
  return 0;
}

}  // namespace

int comic_img_config(int H, int W, int Cin, int Cout, int KH, int KW, int SH, int SW, int Ho, int Wo) {
  if (SH != 1 || SW != 1 || Ho != H || Wo != W || KH * KW > 32 || KH * KW < 2) return -1;
  for (int c = 0; c < kNumCfg; ++c)
    if (kCfg[c].HW == H * W && kCfg[c].Cin == Cin && kCfg[c].Cout == Cout) return c;
  return -1;
}

int comic_img_images_per_group(int cfg) {
  if (cfg < 0 || cfg >= kNumCfg) return 0;
  return kCfg[cfg].G;
}


int comic_img_launch(int cfg, const ComicImgArgs& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_img<9, 3, 1, 4, 4>(a, st);
    case 1: return launch_img<9, 3, 1, 4, 5>(a, st);
    case 2: return launch_img<9, 3, 1, 4, 6>(a, st);
    case 3: return launch_img<9, 2, 1, 4, 4>(a, st);
    case 4: return launch_img<9, 3, 1, 4, 5>(a, st);      // 160 channels on the 192-channel geometry
    case 5: return launch_img<10, 3, 4, 2, 2>(a, st);
    case 6: return launch_img<10, 3, 4, 2, 3>(a, st);
    case 7: return launch_img<5, 6, 2, 4, 12>(a, st);
    case 8: return launch_img<5, 6, 2, 4, 14>(a, st);
    default:
      comic_set_error("conv_img: unknown configuration %d", cfg);
      return 2;
  }
}

// Chains of stride-1 SAME 7-tap convs over 12x12 maps with Cin = every inner Cout in {128, 160, 192} and 192 channels out of
// the last conv (the branches of Mixed_6b-e): 1 where conv_img_chain_kernel serves the shape.
int comic_img_chain_supported(int H, int W, int Cin) {
  return H * W == 144 && (Cin == 128 || Cin == 160 || Cin == 192);
}

int comic_img_chain_launch(const ComicChainArgs& a, hipStream_t st) {
  if (!comic_img_chain_supported(a.H, a.W, a.Cin) || a.n_members < 1 || a.n_members > kChainMaxMembers) {
    comic_set_error("conv_img_chain: unsupported shape (%dx%d, Cin %d, %d members)", a.H, a.W, a.Cin, a.n_members);
    return 2;
  }
  switch (a.Cin) {
    case 128: return launch_chain<2, 4>(a, st);
    case 160: return launch_chain<3, 5>(a, st);
    default: return launch_chain<3, 6>(a, st);
  }
}

extern "C" int comic_cnn_pack_frag_weights(const void* w_plan, void* w_frag, const int64_t* table_dev, int n_weights,
                                           int64_t total_elems, void* stream) {
  COMIC_REQUIRE(w_plan && w_frag && table_dev && n_weights > 0 && total_elems % 8 == 0, "pack_frag_weights: bad arguments");
  const long chunks = total_elems / 8;
  hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)cdiv64(chunks, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w_plan, (bf16_t*)w_frag, (const long*)table_dev, n_weights, (long)total_elems);
  COMIC_LAUNCH_CHECK("pack_frag_weights");
  return 0;
}
