// Developer microbenchmark (not part of libdcamd): issue cost of VALU / transcendental / packed-fp32 / MFMA instructions on gfx950 and whether
// VALU or transcendental work of OTHER waves on a SIMD slows a wave's MFMA stream.  Cycles are s_memtime (shader clock) per instruction per wave.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dev/_build/ubench_valu tools/dev/ubench_valu.hip && tools/dev/_build/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { OP_FMA, OP_PKFMA, OP_EXP, OP_RCP, OP_MFMA, OP_NONE };

template <int OP> __device__ __forceinline__ void body(float (&r)[16], f32x2 (&p)[8], f32x4 (&acc)[8], bf16x8 a, bf16x8 b) {
  if constexpr (OP == OP_FMA) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
  } else if constexpr (OP == OP_PKFMA) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
  } else if constexpr (OP == OP_EXP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
  } else if constexpr (OP == OP_RCP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
  } else if constexpr (OP == OP_MFMA) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
}

// waves [0, split) run OPA, waves [split, nwaves) run OPB; out[block][wave] = cycles per instruction
template <int OPA, int OPB> __global__ void k(int iters, int split, float* out, float seed) {
  float r[16]; f32x2 p[8]; f32x4 acc[8]; bf16x8 a, b;
  for (int i = 0; i < 16; ++i) r[i] = seed + i * 1e-3f + threadIdx.x * 1e-6f;
  for (int i = 0; i < 8; ++i) { p[i] = f32x2{r[i], r[i + 8]}; acc[i] = f32x4{0, 0, 0, 0}; }
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)seed; b[i] = (__bf16)(seed * 0.5f); }
  extern __shared__ char lds[];          // 100 KiB requested at launch: ONE workgroup per CU, so waves/SIMD is what the launch says
  if (seed == 77.f) lds[threadIdx.x] = 1;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (w < split) for (int it = 0; it < iters; ++it) body<OPA>(r, p, acc, a, b);
  else for (int it = 0; it < iters; ++it) body<OPB>(r, p, acc, a, b);
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + acc[i][0] + acc[i][3];
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + w) * 2] = (float)(t1 - t0) / (iters * 16.0f);
  if (s == 12345.678f) out[1] = s;
}

template <int OPA, int OPB> static void run(const char* name, int nwaves, int split) {
  float* d; const int blocks = 256, iters = 2000;
  hipMalloc(&d, blocks * nwaves * 2 * sizeof(float));
  hipMemset(d, 0, blocks * nwaves * 2 * sizeof(float));
  hipFuncSetAttribute((const void*)k<OPA, OPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<OPA, OPB>), dim3(blocks), dim3(nwaves * 64), 100 * 1024, 0, iters, split, d, 1.0f);
  hipDeviceSynchronize();
  float* h = (float*)malloc(blocks * nwaves * 2 * sizeof(float));
  hipMemcpy(h, d, blocks * nwaves * 2 * sizeof(float), hipMemcpyDeviceToHost);
  double sa = 0, sb = 0; int na = 0, nb = 0;
  for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < nwaves; ++w) { const float v = h[(bl * nwaves + w) * 2]; if (w < split) { sa += v; ++na; } else { sb += v; ++nb; } }
  printf("%-44s waves/SIMD %d : A %.2f cycles/instr", name, nwaves / 4, na ? sa / na : 0.0);
  if (nb) printf("   B %.2f cycles/instr", sb / nb);
  printf("\n");
  free(h); hipFree(d);
}

int main() {
  run<OP_FMA, OP_NONE>("v_fma_f32", 4, 4);        run<OP_FMA, OP_NONE>("v_fma_f32", 8, 8);
  run<OP_PKFMA, OP_NONE>("v_pk_fma_f32", 4, 4);   run<OP_PKFMA, OP_NONE>("v_pk_fma_f32", 8, 8);
  run<OP_EXP, OP_NONE>("v_exp_f32", 4, 4);        run<OP_EXP, OP_NONE>("v_exp_f32", 8, 8);
  run<OP_RCP, OP_NONE>("v_rcp_f32", 4, 4);        run<OP_RCP, OP_NONE>("v_rcp_f32", 8, 8);
  run<OP_MFMA, OP_NONE>("mfma_16x16x32_bf16", 4, 4); run<OP_MFMA, OP_NONE>("mfma_16x16x32_bf16", 8, 8);
  // waves 0-3 (one per SIMD) MFMA, waves 4-7 (one per SIMD) the other op
  run<OP_MFMA, OP_FMA>("A mfma | B v_fma_f32", 8, 4);
  run<OP_MFMA, OP_PKFMA>("A mfma | B v_pk_fma_f32", 8, 4);
  run<OP_MFMA, OP_EXP>("A mfma | B v_exp_f32", 8, 4);
  run<OP_MFMA, OP_RCP>("A mfma | B v_rcp_f32", 8, 4);
  run<OP_EXP, OP_FMA>("A v_exp_f32 | B v_fma_f32", 8, 4);
  run<OP_EXP, OP_PKFMA>("A v_exp_f32 | B v_pk_fma_f32", 8, 4);
  run<OP_EXP, OP_RCP>("A v_exp_f32 | B v_rcp_f32", 8, 4);
  // 2 MFMA waves + 1 VALU wave per SIMD
  run<OP_MFMA, OP_EXP>("A mfma x2 | B v_exp_f32", 12, 8);
  run<OP_MFMA, OP_FMA>("A mfma x2 | B v_fma_f32", 12, 8);
  return 0;
}
