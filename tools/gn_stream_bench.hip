// Developer diagnostic (not part of the library): what does a read+write stream of GroupNorm's shape reach on this chip, and which part of
// the one-workgroup-per-sample GroupNorm apply costs the difference to a plain copy?  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/gn_stream_bench tools/gn_stream_bench.hip && gpurun_out/gn_stream_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t chunk16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

template <int MODE>   // 0 copy, 1 affine, 2 affine + silu
__device__ __forceinline__ chunk16 xform(const chunk16 c, const float* sc, const float* sh) {
  if (MODE == 0) return c;
  const bf16x8 v = __builtin_bit_cast(bf16x8, c);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = (float)v[e] * sc[e] + sh[e];
    if (MODE == 2) f = silu_fast(f);
    o[e] = (__bf16)f;
  }
  return __builtin_bit_cast(chunk16, o);
}

// flat grid-stride stream, UNR chunks in flight per lane
template <int MODE, bool NT, int UNR>
__global__ __launch_bounds__(256) void flat_kernel(const chunk16* __restrict__ x, chunk16* __restrict__ y, size_t nchunks, int CP) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.0f + 0.01f * (float)((threadIdx.x + e) % CP); sh[e] = 0.1f; }
  for (; i + (UNR - 1) * stride < nchunks; i += UNR * stride) {
    chunk16 c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) c[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const chunk16 o = xform<MODE>(c[u], sc, sh);
      if (NT) __builtin_nontemporal_store(o, y + i + u * stride); else y[i + u * stride] = o;
    }
  }
  for (; i < nchunks; i += stride) y[i] = xform<MODE>(x[i], sc, sh);
}

// one workgroup of NTHR threads per sample of HW pixels x C channels (CP = C/8 chunks per pixel row), the structure of gn_image_kernel;
// QCHAIN: a dependent chain of `qparts` small loads in front (the producer's quad records), then two barriers
template <int MODE, bool NT, int UNR, int NTHR, bool QCHAIN>
__global__ __launch_bounds__(NTHR) void sample_kernel(const chunk16* __restrict__ x, chunk16* __restrict__ y, int HW, int CP, const float2* q, int qparts, int per) {
  __shared__ float red[128];
  const int t = threadIdx.x;
  const int n = blockIdx.x / per, part = blockIdx.x % per;
  const int TPR = CP, PL = NTHR / TPR;
  const int tc = t % TPR, pl = t / TPR;
  float sc[8], sh[8];
  float S = 0.f;
  if (QCHAIN) {
    if (t < 32) for (int p = 0; p < qparts; ++p) { const float2 v = q[((size_t)n * qparts + p) * 32 + t]; S += v.x + v.y; }
    __syncthreads();
    if (t < 32) red[t] = S;
    __syncthreads();
    S = red[tc & 31] * 1e-30f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.0f + 0.01f * (float)((tc + e) % CP) + S; sh[e] = 0.1f; }
  const int hwp = HW / per;
  const chunk16* src = x + ((size_t)n * HW + (size_t)part * hwp) * CP + tc;
  chunk16* dst = y + ((size_t)n * HW + (size_t)part * hwp) * CP + tc;
  int p = pl;
  for (; p + (UNR - 1) * PL < hwp; p += UNR * PL) {
    chunk16 c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) c[u] = NT ? __builtin_nontemporal_load(src + (size_t)(p + u * PL) * CP) : src[(size_t)(p + u * PL) * CP];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const chunk16 o = xform<MODE>(c[u], sc, sh);
      if (NT) __builtin_nontemporal_store(o, dst + (size_t)(p + u * PL) * CP); else dst[(size_t)(p + u * PL) * CP] = o;
    }
  }
  for (; p < hwp; p += PL) dst[(size_t)p * CP] = xform<MODE>(src[(size_t)p * CP], sc, sh);
}

// one short workgroup per contiguous span of NTHR x UNR chunks: every load issued before the first store, no loop
template <int MODE, bool NT, int UNR, int NTHR>
__global__ __launch_bounds__(NTHR) void span_kernel(const chunk16* __restrict__ x, chunk16* __restrict__ y, size_t nchunks, int CP) {
  const size_t base = (size_t)blockIdx.x * NTHR * UNR + threadIdx.x;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.0f + 0.01f * (float)((threadIdx.x + e) % CP); sh[e] = 0.1f; }
  chunk16 c[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) if (base + u * NTHR < nchunks) c[u] = NT ? __builtin_nontemporal_load(x + base + u * NTHR) : x[base + u * NTHR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) if (base + u * NTHR < nchunks) {
    const chunk16 o = xform<MODE>(c[u], sc, sh);
    if (NT) __builtin_nontemporal_store(o, y + base + u * NTHR); else y[base + u * NTHR] = o;
  }
}

template <typename F> static float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8000, HW = argc > 2 ? atoi(argv[2]) : 1024, C = argc > 3 ? atoi(argv[3]) : 128;
  const int CP = C / 8;
  const size_t nchunks = (size_t)n * HW * CP, bytes = nchunks * 16;
  chunk16 *x, *y; float2* q;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&q, (size_t)n * 4 * 32 * sizeof(float2)));
  CK(hipMemset(x, 0x3c, bytes)); CK(hipMemset(y, 0, bytes)); CK(hipMemset(q, 0, (size_t)n * 4 * 32 * sizeof(float2)));
  const int reps = 10;
  printf("n=%d HW=%d C=%d bf16: %.1f MB read + %.1f MB written per launch\n", n, HW, C, bytes / 1e6, bytes / 1e6);
  auto rep = [&](const char* name, float ms) { printf("  %-58s %8.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
  const int fg = 256 * 16;
#define FLAT(MODE, NT, UNR, G) rep("flat mode=" #MODE " nt=" #NT " unr=" #UNR " grid=" #G, time_ms([&] { hipLaunchKernelGGL((flat_kernel<MODE, NT, UNR>), dim3(G), dim3(256), 0, 0, x, y, nchunks, CP); }, reps))
  FLAT(0, false, 4, fg); FLAT(0, true, 4, fg); FLAT(0, false, 8, fg); FLAT(0, true, 8, fg);
  FLAT(0, false, 4, 256 * 8); FLAT(0, false, 4, 256 * 32);
  FLAT(1, false, 4, fg); FLAT(2, false, 4, fg); FLAT(2, true, 4, fg); FLAT(2, false, 8, fg);
#define SAMP(MODE, NT, UNR, NTHR, QC, PER) rep("sample mode=" #MODE " nt=" #NT " unr=" #UNR " thr=" #NTHR " qchain=" #QC " wg/sample=" #PER, \
    time_ms([&] { hipLaunchKernelGGL((sample_kernel<MODE, NT, UNR, NTHR, QC>), dim3(n * PER), dim3(NTHR), 0, 0, x, y, HW, CP, q, 4, PER); }, reps))
  SAMP(0, false, 4, 512, false, 1); SAMP(0, true, 4, 512, false, 1); SAMP(1, false, 4, 512, false, 1);
  SAMP(2, false, 4, 512, false, 1); SAMP(2, false, 4, 512, true, 1); SAMP(2, true, 4, 512, true, 1);
  SAMP(2, false, 8, 512, true, 1); SAMP(2, false, 2, 512, true, 1);
  SAMP(2, false, 4, 256, true, 1); SAMP(2, false, 8, 256, true, 1);
  SAMP(2, false, 4, 256, true, 2); SAMP(2, false, 4, 256, true, 4); SAMP(2, false, 4, 512, true, 2);
  SAMP(2, false, 4, 1024, true, 1);
#define SPAN(MODE, NT, UNR, NTHR) rep("span mode=" #MODE " nt=" #NT " unr=" #UNR " thr=" #NTHR, \
    time_ms([&] { hipLaunchKernelGGL((span_kernel<MODE, NT, UNR, NTHR>), dim3((unsigned)((nchunks + (size_t)NTHR * UNR - 1) / ((size_t)NTHR * UNR))), dim3(NTHR), 0, 0, x, y, nchunks, CP); }, reps))
  SPAN(0, false, 4, 256); SPAN(0, true, 4, 256); SPAN(2, false, 4, 256); SPAN(2, true, 4, 256);
  SPAN(2, false, 2, 256); SPAN(2, false, 8, 256); SPAN(2, true, 8, 256); SPAN(2, false, 16, 256); SPAN(2, true, 16, 256);
  SPAN(2, false, 4, 512); SPAN(2, true, 4, 512); SPAN(2, false, 8, 512); SPAN(2, true, 8, 512);
  SPAN(2, false, 4, 1024); SPAN(2, true, 4, 1024); SPAN(2, false, 8, 1024); SPAN(2, false, 2, 1024);
  SPAN(2, false, 4, 128); SPAN(2, false, 8, 128); SPAN(2, false, 4, 64); SPAN(2, false, 16, 64);
  SAMP(2, true, 4, 1024, true, 1); SAMP(2, false, 8, 1024, true, 1); SAMP(2, false, 2, 1024, true, 1);
  SAMP(2, false, 4, 256, true, 16); SAMP(2, false, 4, 512, true, 8); SAMP(2, false, 4, 512, true, 4); SAMP(2, true, 4, 256, true, 16);
  return 0;
}
