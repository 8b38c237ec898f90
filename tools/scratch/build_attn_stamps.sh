#!/bin/bash
set -e
cd /root/repo/comic-compact-image-captioning-with-attention_amd/csrc
python - <<'PY'
s=open('decoder.hip').read()
s=s.replace('constexpr int kAttnWaves = 16;','__device__ long long g_at_stamps[256 * 16 * 8];\n#define ASTAMP(i) do { if (lane == 0) g_at_stamps[(blockIdx.x * 16 + wave) * 8 + (i)] = wall_clock64(); } while (0)\nextern "C" int comic_debug_at_stamps(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_at_stamps), sizeof(long long) * 256 * 16 * 8); }\nconstexpr int kAttnWaves = 16;',1)
i=s.index('void attn_fwd_kernel(AttnArgs a) {')
j=s.index('// ---- large memories: gridDim.y workgroups per batch row')
k=s[i:j]
k=k.replace("  int b = blockIdx.x;\n","  int b = blockIdx.x;\n  ASTAMP(0);\n",1)
k=k.replace("  sum_q_parts<EPL>(a.q + (size_t)b * D + k0, a.q_parts, (size_t)a.d.B * D, qv);\n","  sum_q_parts<EPL>(a.q + (size_t)b * D + k0, a.q_parts, (size_t)a.d.B * D, qv);\n  ASTAMP(1);\n",1)
k=k.replace("  float2 vpre[kPre];\n","  ASTAMP(2);\n  float2 vpre[kPre];\n",1)
k=k.replace("  __syncthreads();\n  // probability fn per head","  __syncthreads();\n  ASTAMP(3);\n  // probability fn per head",1)
k=k.replace("  __syncthreads();\n  // context: ctx[c]","  __syncthreads();\n  ASTAMP(4);\n  // context: ctx[c]",1)
k=k.replace("    finish(2 * tid + 1, a1);\n    return;","    finish(2 * tid + 1, a1);\n    ASTAMP(5);\n    return;",1)
assert k.count('ASTAMP(')==6, k.count('ASTAMP(')
s=s[:i]+k+s[j:]
open('decoder_var.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c decoder_var.hip -o /tmp/dec_STAMPS.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcomic_hip_ASTAMPS.so conv.o conv_ws.o conv_stem.o conv_img.o gemm.o /tmp/dec_STAMPS.o decode.o beam_logits.o lstm_stream.o decoder_exec.o decoder_fused.o decoder_persist.o decoder_persist_bwd.o preprocess.o abi.o -lpthread
rm decoder_var.hip
