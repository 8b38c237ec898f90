// Micro-benchmark (diagnostic, not part of the library): cost of an in-launch all-gather on MI355X.
// 256 workgroups in 4 groups of 64; per round every workgroup sc1-stores its 1/64 slice of its group's buffer
// (BYTES per group), then validates-by-sentinel and reads the WHOLE group buffer with sc1 loads.
// Prints the mean time from "own slice stored" to "whole buffer read" for BYTES = 32, 64, 128 KB.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
constexpr unsigned kSent = 0xFFFFDEADu;
constexpr int kRounds = 32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}

template <int NLOAD, bool SC1_LOADS>   // NLOAD 16-byte loads per lane (512 threads): BYTES = NLOAD * 8 KB
__global__ __launch_bounds__(512) void gather_kernel(unsigned* buf, unsigned long long* stamps, unsigned* sink) {
  const int grp = blockIdx.x / 64, wi = blockIdx.x % 64, tid = threadIdx.x;
  constexpr long BYTES = (long)NLOAD * 8192;
  unsigned acc = 0;
  unsigned long long t_sum = 0;
  for (int r = 0; r < kRounds; ++r) {
    unsigned char* base = (unsigned char*)buf + ((long)r * 4 + grp) * BYTES;
    const __amdgpu_buffer_rsrc_t rs = rsrc(base, BYTES);
    // own slice: BYTES / 64 bytes, 16 B per thread for the first BYTES / 1024 threads
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (tid < BYTES / 1024) {
      const u32x4_t v = {(unsigned)r + 1, (unsigned)wi, (unsigned)tid, 7u};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(wi * (BYTES / 64) + tid * 16), 0, 16);
    }
    __syncthreads();
    u32x4_t x[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 512 + tid) * 16, 0, SC1_LOADS ? 16 : 0);
    for (unsigned spins = 0; spins < (1u << 18); ++spins) {
      unsigned bad = 0;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i)
        if (__any(x[i].x == kSent || x[i].y == kSent || x[i].z == kSent || x[i].w == kSent)) bad |= 1u << i;
      if (!bad) break;
      __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NLOAD; ++i)
        if ((bad >> i) & 1u) x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 512 + tid) * 16, 0, 16);
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) acc += x[i].x + x[i].w;
    __syncthreads();
    t_sum += __builtin_amdgcn_s_memrealtime() - t0;
  }
  if (tid == 0) stamps[blockIdx.x] = t_sum;
  sink[blockIdx.x * 512 + tid] = acc;
}

template <int NLOAD, bool SC1>
void run(const char* name) {
  constexpr long BYTES = (long)NLOAD * 8192;
  const long total = BYTES * 4 * kRounds;
  unsigned* buf; unsigned long long* st; unsigned* sink;
  hipMalloc((void**)&buf, total); hipMalloc((void**)&st, 256 * 8); hipMalloc((void**)&sink, 256 * 512 * 4);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(buf, 0, total);
    std::vector<unsigned> h(total / 4, kSent);
    hipMemcpy(buf, h.data(), total, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((gather_kernel<NLOAD, SC1>), dim3(256), dim3(512), 0, 0, buf, st, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    unsigned long long hs[256];
    hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)hs[i];
    mean = mean / 256 / kRounds / 100.0;
    if (mean < best) best = mean;
  }
  printf("%-28s %4ld KB per group: %.2f us per round (store own slice -> whole buffer validated)\n", name, BYTES / 1024, best);
  hipFree(buf); hipFree(st); hipFree(sink);
}

int main() {
  run<1, true>("sc1 loads");
  run<4, true>("sc1 loads");
  run<8, true>("sc1 loads");
  run<16, true>("sc1 loads");
  run<16, false>("plain first load, sc1 polls");
  return 0;
}
