#!/bin/bash
# builds lib/libcomic_hip_STAMPS.so: beam_logits.hip with phase clocks (scratch, not shipped)
set -e
cd /root/repo/comic-compact-image-captioning-with-attention_amd/csrc
python - <<'PY'
s=open('beam_logits.hip').read()
s=s.replace("namespace {\n\nconstexpr int kChunkCols","#undef STAMP\n__device__ long long g_bl_stamps[256 * 8 * 8];\n#define STAMP(i) do { if (lane == 0) g_bl_stamps[(blockIdx.x * 8 + wave) * 8 + (i)] = wall_clock64(); } while (0)\nextern \"C\" int comic_debug_bl_stamps(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bl_stamps), sizeof(long long) * 256 * 8 * 8); }\nnamespace {\n\nconstexpr int kChunkCols",1)
s=s.replace("  if (wave == 0 && lane < 32) bl_dma16(a.bias_pad","  STAMP(0);\n  if (wave == 0 && lane < 32) bl_dma16(a.bias_pad")
s=s.replace("    __builtin_amdgcn_s_barrier();                          // ... everybody's; the other buffer is no longer read\n","    __builtin_amdgcn_s_barrier();                          // ... everybody's; the other buffer is no longer read\n    if (q < 4) STAMP(1 + q);\n")
s=s.replace("  if constexpr (NT == 0) return;","  STAMP(5);\n  if constexpr (NT == 0) return;")
s=s.replace("        a.cand_i[ro * a.W + k] = bi == 0x7fffffff ? -1 : bi;\n      }\n    }\n  }\n}","        a.cand_i[ro * a.W + k] = bi == 0x7fffffff ? -1 : bi;\n      }\n    }\n  }\n  STAMP(6);\n}")
s=s.replace('extern "C" int comic_debug_bl_stamps','__device__ long long g_bm_stamps[256 * 8 * 8];\n#define MSTAMP(i) do { if (lane == 0) g_bm_stamps[(blockIdx.x * 8 + wave) * 8 + (i)] = wall_clock64(); } while (0)\nextern "C" int comic_debug_bm_stamps(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bm_stamps), sizeof(long long) * 256 * 8 * 8); }\nextern "C" int comic_debug_bl_stamps',1)
s=s.replace("  constexpr int CP = 8, CQ = 8;\n  const bool pre","  MSTAMP(0);\n  constexpr int CP = 8, CQ = 8;\n  const bool pre")
s=s.replace("  __syncthreads();\n  // candidate slots: beam w, slot k < chunks * W.","  __syncthreads();\n  MSTAMP(1);\n  // candidate slots: beam w, slot k < chunks * W.")
s=s.replace("  __syncthreads();\n  // the W best of every beam (a wave per beam)","  __syncthreads();\n  MSTAMP(2);\n  // the W best of every beam (a wave per beam)")
s=s.replace("  __syncthreads();\n  // the W best of the finalists","  __syncthreads();\n  MSTAMP(3);\n  // the W best of the finalists")
s=s.replace("  __syncthreads();\n  if (tid < W) {\n    const int f = s_sel[tid];","  __syncthreads();\n  MSTAMP(4);\n  if (tid < W) {\n    const int f = s_sel[tid];")
s=s.replace("      if (done == gridDim.x && steps_executed[0] == max_steps) steps_executed[0] = t + 1;\n    }\n  }\n}","      if (done == gridDim.x && steps_executed[0] == max_steps) steps_executed[0] = t + 1;\n    }\n  }\n  MSTAMP(5);\n}")
assert s.count('MSTAMP(') == 7, s.count('MSTAMP(')
open('beam_logits_var.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c beam_logits_var.hip -o /tmp/bl_STAMPS.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcomic_hip_STAMPS.so conv.o conv_ws.o conv_stem.o conv_img.o gemm.o decoder.o decode.o /tmp/bl_STAMPS.o lstm_stream.o decoder_exec.o decoder_fused.o decoder_persist.o decoder_persist_bwd.o preprocess.o abi.o -lpthread
rm beam_logits_var.hip
