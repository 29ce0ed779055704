// mfma_peak.hip — developer micro-benchmark: what the matrix pipe of THIS box sustains (the practical
// ceiling the conv / GEMM kernels are compared with), in three loop shapes of the halo kernel's tap step:
//   mode 0: 32 MFMA 16x16x32 bf16 per iteration, operands in registers (pure matrix-pipe rate)
//   mode 1: + the 12 ds_read_b128 of a tap (4 W + 8 X fragments), asynchronous with counted waits
//   mode 2: + one s_barrier per iteration (the per-tap workgroup barrier)
//   mode 3: + the LDS-DMA fills of the halo kernel (8 KiB W tile per step, 3-stage ring, counted vmcnt; 24 KiB X halo
//           every 9th step) from an L2-resident buffer
//   mode 4: as 3 but the W ring only (no X halo traffic)
//   mode 5: as 4, but the W tile gathered the way conv3_halo does it: 128 rows of 64 B, 4608 B apart (K = 2304)
//   mode 6: as 5 with 128-byte rows (two 64-channel... i.e. 16 KiB every second step): full cache lines
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t chunk16;

__device__ __forceinline__ chunk16 ds_read16_async(uint32_t addr) {
  chunk16 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
template <int N> __device__ __forceinline__ void lgkm_wait(chunk16& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ int g_xpat;   // X fragment address pattern under test (mode >= 1)
template <int MODE>
__global__ __launch_bounds__(256, 2) void peak_kernel(float* out, int iters, const char* src) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 72 * 1024 / 16; i += 256) reinterpret_cast<chunk16*>(smem)[i] = chunk16{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  chunk16 wf[4], xf[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = chunk16{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
  for (int j = 0; j < 8; ++j) xf[j] = chunk16{0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const uint32_t woff = base + 49152 + (lane & 15) * 64 + (lane >> 4) * 16;
  // X fragment read patterns: 0 row-major 64-B rows (conv3_halo today); 1 chunk-major planes 128 B mod 256 apart; 2 chunk-major
  // planes 0 mod 256 apart; 3 row-major with the W tile's XOR swizzle; 4 row-major, chunk rotated by row
  const int xp = g_xpat, lr_ = lane & 15, lq_ = lane >> 4;
  uint32_t xlane = lr_ * 64 + lq_ * 16;
  if (xp == 1) xlane = lq_ * (344 * 16) + lr_ * 16;
  if (xp == 2) xlane = lq_ * (352 * 16) + lr_ * 16;
  if (xp == 3) xlane = lr_ * 64 + ((lq_ ^ ((0x78 >> (((lr_ >> 2) & 3) << 1)) & 3)) << 4);
  if (xp == 4) xlane = lr_ * 64 + (((lq_ + (lr_ >> 2)) & 3) << 4);
  const uint32_t xoff = base + ((t >> 6) >> 1) * (xp == 1 || xp == 2 ? 2048 : 8192) + xlane;
  const uint32_t xj = (xp == 1 || xp == 2) ? 256 : 1024;          // fragment j: 16 rows further
  const uint32_t xtap = (xp == 1 || xp == 2) ? 16 : 64;           // tap: one row further
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const char* wsrc = src + t * 16;                               // 1 MiB of "weights", L2 resident
  const char* xsrc = src + (1 << 20) + (size_t)(blockIdx.x & 1023) * 24576 + t * 16;
  if (MODE >= 3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + i * 4096), (lptr_t)(smem + 49152 + i * 4096 + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 8192 + i * 4096), (lptr_t)(smem + 49152 + 8192 + i * 4096 + wave * 1024), 16, 0, 0);
    }
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      const bool xfly = MODE == 3 && (it % 9) >= 1 && (it % 9) <= 2;
      if (xfly) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
    if (MODE >= 3) {
      const int s = it + 2;
      if (MODE == 5) {
        const int s72 = s % 72, cc = s72 / 9, tp = s72 - cc * 9;
        const char* wp = src + (size_t)(t >> 2) * 4608 + (t & 3) * 16 + tp * 512 + cc * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(wp + (size_t)i * 64 * 4608), (lptr_t)(smem + 49152 + (s % 3) * 8192 + i * 4096 + wave * 1024), 16, 0, 0);
      } else if (MODE == 6) {
        // same bytes per step on average, fetched as whole 128-B lines: rows of 128 B (t>>3), 64 rows per instruction pair
        const int s72 = s % 72, cc = s72 / 9, tp = s72 - cc * 9;
        const char* wp = src + (size_t)(t >> 3) * 4608 + (t & 7) * 16 + tp * 512 + (cc >> 1) * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(wp + (size_t)(i * 32 + (s & 1) * 64) * 4608), (lptr_t)(smem + 49152 + (s % 3) * 8192 + i * 4096 + wave * 1024), 16, 0, 0);
      } else {
      const char* wp = wsrc + (size_t)(s & 127) * 8192;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wp + i * 4096), (lptr_t)(smem + 49152 + (s % 3) * 8192 + i * 4096 + wave * 1024), 16, 0, 0);
      }
      if (MODE == 3 && it % 9 == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(xsrc + i * 4096), (lptr_t)(smem + ((it / 9) & 1) * 24576 + i * 4096 + wave * 1024), 16, 0, 0);
      }
    }
    if (MODE >= 1) {
      const uint32_t tap = (it % 9) * xtap;
      __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = ds_read16_async(woff + i * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = ds_read16_async(xoff + tap + j * xj);
    }
#define GROUP(J, N)                                                                     \
    if (MODE >= 1) lgkm_wait<N>(xf[J]);                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
      acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]), __builtin_bit_cast(bf16x8, xf[J]), acc[i][J], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 1) { lgkm_wait<7>(wf[0]); lgkm_wait<7>(wf[1]); lgkm_wait<7>(wf[2]); lgkm_wait<7>(wf[3]); }
    GROUP(0, 7) GROUP(1, 6) GROUP(2, 5) GROUP(3, 4) GROUP(4, 3) GROUP(5, 2) GROUP(6, 1) GROUP(7, 0)
#undef GROUP
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + t] = s;
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// same loop skeleton on v_mfma_f32_32x32x16_bf16: 16 MFMAs per step (same FLOPs, same 12 fragment reads, same LDS-DMA), but an
// MFMA of this shape holds the SIMD's vector issue for 8 of its 32 cycles instead of 8 of 16 (MI355X_MICROARCH.md)
template <int MODE>
__global__ __launch_bounds__(256, 2) void peak32_kernel(float* out, int iters, const char* src) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 72 * 1024 / 16; i += 256) reinterpret_cast<chunk16*>(smem)[i] = chunk16{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  chunk16 wf[4], xf[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = chunk16{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
  for (int j = 0; j < 8; ++j) xf[j] = chunk16{0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  // fragment (32 rows x 16 k): lane -> row lane&31, 16 B at k-half (lane>>5); W fragment f / k-half h at +f*2048 + h*32
  const uint32_t woff = base + 49152 + (lane & 31) * 64 + (lane >> 5) * 16;
  const uint32_t xoff = base + ((t >> 6) >> 1) * 8192 + (lane & 31) * 64 + (lane >> 5) * 16;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const char* wsrc = src + t * 16;
  const char* xsrc = src + (1 << 20) + (size_t)(blockIdx.x & 1023) * 24576 + t * 16;
  if (MODE >= 3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + i * 4096), (lptr_t)(smem + 49152 + i * 4096 + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 8192 + i * 4096), (lptr_t)(smem + 49152 + 8192 + i * 4096 + wave * 1024), 16, 0, 0);
    }
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      const bool xfly = MODE == 3 && (it % 9) >= 1 && (it % 9) <= 2;
      if (xfly) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
    if (MODE >= 3) {
      const int s = it + 2;
      const char* wp = wsrc + (size_t)(s & 127) * 8192;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wp + i * 4096), (lptr_t)(smem + 49152 + (s % 3) * 8192 + i * 4096 + wave * 1024), 16, 0, 0);
      if (MODE == 3 && it % 9 == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(xsrc + i * 4096), (lptr_t)(smem + ((it / 9) & 1) * 24576 + i * 4096 + wave * 1024), 16, 0, 0);
      }
    }
    if (MODE >= 1) {
      const uint32_t tap = (it % 9) * 64;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      // W: 2 cout fragments x 2 k-halves; X: 4 pixel fragments x 2 k-halves
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = ds_read16_async(woff + (i >> 1) * 2048 + (i & 1) * 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = ds_read16_async(xoff + tap + (j >> 1) * 2048 + (j & 1) * 32);
    }
#define GROUP32(J, N)                                                                   \
    if (MODE >= 1) { lgkm_wait<N>(xf[2 * J]); lgkm_wait<N>(xf[2 * J + 1]); }               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                         \
      acc[i][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[2 * i]), __builtin_bit_cast(bf16x8, xf[2 * J]), acc[i][J], 0, 0, 0); \
      acc[i][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[2 * i + 1]), __builtin_bit_cast(bf16x8, xf[2 * J + 1]), acc[i][J], 0, 0, 0); \
    }                                                                                       \
    __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 1) { lgkm_wait<6>(wf[0]); lgkm_wait<6>(wf[1]); lgkm_wait<6>(wf[2]); lgkm_wait<6>(wf[3]); }
    GROUP32(0, 6) GROUP32(1, 4) GROUP32(2, 2) GROUP32(3, 0)
#undef GROUP32
  }
  float sres = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sres += acc[i][j][e];
  out[blockIdx.x * 256 + t] = sres;
}

template <int MODE>
static void run(const char* name, float* out, int blocks, const char* src) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(peak_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(peak_kernel<MODE>, dim3(blocks), dim3(256), 72 * 1024, 0, out, 200, src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(peak_kernel<MODE>, dim3(blocks), dim3(256), 72 * 1024, 0, out, iters, src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /*waves*/ * iters * 32 * (16.0 * 16 * 32 * 2);
  printf("%-34s blocks=%5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

template <int MODE>
static void run32(const char* name, float* out, int blocks, const char* src) {
  const int iters = 4000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peak32_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(peak32_kernel<MODE>, dim3(blocks), dim3(256), 72 * 1024, 0, out, 200, src);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(peak32_kernel<MODE>, dim3(blocks), dim3(256), 72 * 1024, 0, out, iters, src);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 16 * 2);
  printf("%-34s blocks=%5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 8192 * 256 * sizeof(float));
  char* src;
  hipMalloc(&src, (1 << 20) + 1024 * 24576 + 65536);
  {  // random bf16 operands in (-1, 1): all-zero operands draw less power and flatter the clock
    const size_t nb = (1 << 20) + 1024 * 24576 + 65536;
    uint16_t* h = (uint16_t*)malloc(nb);
    uint32_t r = 12345u;
    for (size_t i = 0; i < nb / 2; ++i) {
      r = r * 1664525u + 1013904223u;
      const float f = ((r >> 8) & 0xffff) / 32768.0f - 1.0f;
      uint32_t u; memcpy(&u, &f, 4);
      h[i] = getenv("PEAK_ZERO") ? 0 : (uint16_t)(u >> 16);
    }
    hipMemcpy(src, h, nb, hipMemcpyHostToDevice);
    free(h);
  }
  for (int xp = 1; xp <= 4; ++xp) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xpat), &xp, sizeof(xp));
    char nm[64]; snprintf(nm, sizeof(nm), "X pattern %d: mfma + reads", xp);
    run<1>(nm, out, 2048, src);
  }
  { int xp = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xpat), &xp, sizeof(xp)); }
  for (int blocks : {2048}) {
    run<0>("mfma only", out, blocks, src);
    run<1>("mfma + 12 ds_read_b128 / 32 mfma", out, blocks, src);
    run<2>("  + s_barrier per step", out, blocks, src);
    run<4>("  + LDS-DMA W ring (8 KiB/step)", out, blocks, src);
    run<3>("  + LDS-DMA W ring + X halo", out, blocks, src);
    run<5>("  W ring as 64-B rows, 4608 B apart", out, blocks, src);
    run<6>("  W ring as 128-B rows, 4608 B apart", out, blocks, src);
    run32<0>("32x32x16: mfma only", out, blocks, src);
    run32<1>("32x32x16: + 12 ds_read_b128", out, blocks, src);
    run32<2>("32x32x16:   + s_barrier", out, blocks, src);
    run32<4>("32x32x16:   + LDS-DMA W ring", out, blocks, src);
    run32<3>("32x32x16:   + W ring + X halo", out, blocks, src);
  }
  hipFree(out);
  return 0;
}
