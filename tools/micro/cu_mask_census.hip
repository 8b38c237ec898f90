// Which CUs does a CU-masked stream reach?  For a few 256-bit masks: launch 512 one-per-CU workgroups on a stream made by
// hipExtStreamCreateWithCUMask and list, per XCC, the (se, sh, cu) triples the workgroups reported.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask_census.hip -o /tmp/cu_mask_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <set>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void census(uint32_t* out, int spin) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    lds[0] = (char)hw;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
}

int main() {
  const int NWG = 512;
  uint32_t* d;
  CK(hipMalloc(&d, NWG * 8));
  uint32_t h[NWG * 2];
  struct { const char* name; uint32_t w[8]; } masks[] = {
      {"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
      {"lo128", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
      {"hi128", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
      {"even", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
      {"byte0", {0x000000ffu, 0, 0, 0, 0, 0, 0, 0}},
      {"bit0-15", {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0}},
      {"word0", {~0u, 0, 0, 0, 0, 0, 0, 0}},
      {"lo16of32x8", {0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu, 0x0000ffffu}},
  };
  for (auto& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { printf("mask %s: create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0xff, NWG * 8, st));
    hipLaunchKernelGGL(census, dim3(NWG), dim3(256), 100 * 1024, st, d, 2000 /* 20 us at 100 MHz */);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, d, NWG * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<int>> per_xcc;
    std::map<int, std::set<int>> first64;
    for (int b = 0; b < NWG; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    int total = 0;
    printf("mask %-12s:", m.name);
    for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total CUs %d\n", total);
    if (!strcmp(m.name, "byte0") || !strcmp(m.name, "bit0-15") || !strcmp(m.name, "word0")) {
      for (auto& kv : per_xcc) { printf("    xcc%d se/sh/cu:", kv.first); for (int v : kv.second) printf(" %d/%d/%d", v / 32, (v / 16) & 1, v & 15); printf("\n"); }
    }
    printf("    blocks 0..15 -> xcc:"); for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b + 1] & 0xf); printf("\n");
    CK(hipStreamDestroy(st));
  }
  return 0;
}
