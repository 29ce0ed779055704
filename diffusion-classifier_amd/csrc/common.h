// common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libdcamd.
// Wave = 64 lanes everywhere; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dcamd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 16 raw bytes, the unit of every global / LDS transfer in this library (a native vector so it
// lives in 4 VGPRs; a struct-of-array form is demoted to scratch by hipcc).
typedef __attribute__((ext_vector_type(4))) uint32_t chunk16;

void dc_set_error(const char* fmt, ...);
int dc_check_launch(const char* what);

#define DC_REQUIRE(cond, code, ...) \
  do { if (!(cond)) { dc_set_error(__VA_ARGS__); return (code); } } while (0)

// ---- in-kernel clock stamps (diagnostic builds only: -DDC_CLOCK_STAMPS, tools/clock_probe.py) ------------------------------
// MI355X_MICROARCH.md "DVFS give-back" item 6: the clock a kernel actually holds is (delta s_memtime) / (delta s_memrealtime) x 100 MHz,
// stamped once around its main loop.  DC_CLOCK(k), k = 0 before / 1 after the loop, by thread 0 of every workgroup, into a buffer of
// its own (4 x u64 per workgroup) that nothing else reads; no stamp is compiled into the shipped library.
#ifdef DC_CLOCK_STAMPS
#define DC_CLOCK_DECL(tu)                                                                                                      \
  static __device__ unsigned long long* g_clk_buf;                                                                             \
  extern "C" void dc_debug_set_clk_##tu(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk_buf), &p, sizeof(p)); }
#define DC_CLOCK(k)                                                                                                            \
  do {                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    if (threadIdx.x == 0 && g_clk_buf) {                                                                                       \
      g_clk_buf[(size_t)blockIdx.x * 4 + 2 * (k)] = __builtin_amdgcn_s_memtime();                                              \
      g_clk_buf[(size_t)blockIdx.x * 4 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime();                                      \
    }                                                                                                                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  } while (0)
#else
#define DC_CLOCK_DECL(tu)
#define DC_CLOCK(k) do {} while (0)
#endif

// ---- element traits ---------------------------------------------------------------
template <typename T> struct Elem;

template <> struct Elem<float> {
  using vec = f32x4;
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<__bf16> {
  using vec = bf16x8;
  typedef __attribute__((ext_vector_type(4))) __bf16 vec4;      // 8 bytes: one 16x16x16 MFMA operand / one 4-element store
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
  static __device__ __forceinline__ __bf16 from_f(float v) { return (__bf16)v; }  // RNE, NaN-preserving
};
template <> struct Elem<_Float16> {
  using vec = f16x8;
  typedef __attribute__((ext_vector_type(4))) _Float16 vec4;
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
  static __device__ __forceinline__ _Float16 from_f(float v) { return (_Float16)v; }
};

template <typename T>
__device__ __forceinline__ void chunk_to_f(const chunk16 c, float* f) {
  const typename Elem<T>::vec v = __builtin_bit_cast(typename Elem<T>::vec, c);
#pragma unroll
  for (int i = 0; i < Elem<T>::EPC; ++i) f[i] = Elem<T>::to_f(v[i]);
}
template <typename T>
__device__ __forceinline__ chunk16 f_to_chunk(const float* f) {
  typename Elem<T>::vec v;
#pragma unroll
  for (int i = 0; i < Elem<T>::EPC; ++i) v[i] = Elem<T>::from_f(f[i]);
  return __builtin_bit_cast(chunk16, v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// 16-bit outputs: v_exp_f32 + v_rcp_f32 (~1 ulp each, the result is rounded to 8 / 11 mantissa bits anyway) instead of ocml expf
// and the IEEE divide (~20 VALU ops): the GroupNorm(+SiLU) pass is otherwise co-limited by VALU issue (~200 instructions per
// 16-byte chunk against ~6 TB/s of VALU-side ceiling), and the GEGLU epilogue was VALU-bound outright
__device__ __forceinline__ float silu_fast_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
template <typename T> __device__ __forceinline__ float silu_t(float x) { return sizeof(T) == 2 ? silu_fast_f(x) : silu_f(x); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute): 1 rcp + 1 exp + 5 fma instead of ocml erff's
// ~40 VALU ops — the GEGLU epilogue evaluates it for every output element (it was ~1/3 of that GEMM's time).
__device__ __forceinline__ float erf_as_f(float z) {
  const float az = fabsf(z);
  const float t = 1.0f / (1.0f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * expf(-az * az);
  return copysignf(r, z);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_fast_f(float x) {       // same formula on v_rcp_f32 / v_exp_f32 (16-bit outputs)
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
  return 0.5f * x * (1.0f + copysignf(r, z));
}
#ifdef DC_GEGLU_ABL   // timing-only diagnostic build: the GEGLU epilogue without its erf (results wrong on purpose)
template <typename T> __device__ __forceinline__ float gelu_erf_t(float x) { return x; }
#else
template <typename T> __device__ __forceinline__ float gelu_erf_t(float x) { return sizeof(T) == 2 ? gelu_erf_fast_f(x) : gelu_erf_f(x); }
#endif
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k = 0.79788456080286535588f;  // sqrt(2/pi)
  // 0.5 x (1 + tanh u) == x / (1 + exp(-2u)): one exp + one divide instead of ocml tanhf (the DiT fc1 epilogue was
  // VALU-bound on it); exp(-2u) -> inf / 0 at the tails gives the exact limits -0 / x
  const float u = k * (x + 0.044715f * x * x * x);
  return x / (1.0f + expf(-2.0f * u));
}

template <typename T> __device__ __forceinline__ float gelu_tanh_t(float x) {
  if (sizeof(T) != 2) return gelu_tanh_f(x);
  const float u = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));     // exp(-2u) on v_exp_f32
}

// store one value of runtime dtype
__device__ __forceinline__ void store_as(void* base, size_t idx, int dtype, float v) {
  if (dtype == DC_F32) reinterpret_cast<float*>(base)[idx] = v;
  else if (dtype == DC_BF16) reinterpret_cast<__bf16*>(base)[idx] = (__bf16)v;
  else reinterpret_cast<_Float16*>(base)[idx] = (_Float16)v;
}
__device__ __forceinline__ float load_as(const void* base, size_t idx, int dtype) {
  if (dtype == DC_F32) return reinterpret_cast<const float*>(base)[idx];
  if (dtype == DC_BF16) return (float)reinterpret_cast<const __bf16*>(base)[idx];
  return (float)reinterpret_cast<const _Float16*>(base)[idx];
}

static inline int dc_dtype_size(int dt) { return dt == DC_F32 ? 4 : 2; }

// ---- asynchronous LDS fragment reads with hand-counted waits -------------------------------------------------
// hipcc waits lgkmcnt(0) in front of the first MFMA that uses a ds_read result, and under register pressure it
// even re-serialises "read, wait, 4 MFMA" (seen in the ISA of both GEMM kernels).  These helpers issue the reads
// as opaque instructions and tie each counted wait to the registers it releases, so a block of N reads stays in
// flight behind the MFMAs: LDS returns data in issue order, later-issued scalar loads only make a wait stricter.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ chunk16 ds_read16_async(uint32_t addr) {
  chunk16 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// lgkmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt / expcnt fields all ones = no wait): the compiler's own waitcnt
// bookkeeping sees it, so it does not drop a second lgkmcnt(0) of its own into the middle of the asynchronous reads
template <int OFF> __device__ __forceinline__ chunk16 ds_read16_async_off(uint32_t addr) {     // addr + OFF (OFF < 65536, immediate)
  chunk16 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lgkm_fence0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
template <int N> __device__ __forceinline__ void lgkm_wait(chunk16& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void lgkm_wait(chunk16& a, chunk16& b, chunk16& c, chunk16& d, chunk16& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}
