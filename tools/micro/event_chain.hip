// What a cross-stream hand-off costs the stream that RECORDS it: a chain of N small dependent kernels on one stream, (a) bare,
// (b) an event recorded after every kernel, (c) the event also awaited by a second stream that launches a kernel of its own
// behind it (the cnn_finetune backward: every conv forks its weight gradient to another lane), (d) as (c) with one reusable
// event per slot instead of create / destroy, (e) as (c) with hipEventReleaseToDevice events.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/event_chain tools/micro/event_chain.hip && /tmp/event_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  const int N = 200, iters = 2000;
  float *a, *b;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096);
  hipMemset(a, 0, 4096); hipMemset(b, 0, 4096);
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  std::vector<hipEvent_t> pool(N);
  for (auto& e : pool) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      hipEvent_t t0, t1;
      hipEventCreate(&t0); hipEventCreate(&t1);
      auto h0 = std::chrono::steady_clock::now();
      hipEventRecord(t0, s0);
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s0, a, iters);
        if (mode == 0) continue;
        hipEvent_t ev;
        if (mode == 3) ev = pool[i];
        else hipEventCreateWithFlags(&ev, hipEventDisableTiming | (mode == 4 ? hipEventReleaseToDevice : 0));
        hipEventRecord(ev, s0);
        if (mode >= 2) {
          hipStreamWaitEvent(s1, ev, 0);
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s1, b, iters);
        }
        if (mode != 3) hipEventDestroy(ev);
      }
      hipEventRecord(t1, s0);
      auto h1 = std::chrono::steady_clock::now();
      hipStreamSynchronize(s0); hipStreamSynchronize(s1);
      float ms = 0;
      hipEventElapsedTime(&ms, t0, t1);
      if (rep == 2)
        printf("mode %d: %.1f us per link on the device, %.1f us per link to issue\n", mode, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
      hipEventDestroy(t0); hipEventDestroy(t1);
    }
  }
  return 0;
}
