// attention.hip — self-attention softmax(q k^T / sqrt(d)) v for the UNet token counts, gfx950.
//
// Replaces F.scaled_dot_product_attention in diffusers Attention (attn1 of
// BasicTransformerBlock) reached through reference nets/unet.py:186.  (Cross-attention over
// the single class token needs no kernel: softmax over one key is 1, see DESIGN.md.)
//
// UNet shapes are tiny (L<=256 tokens, d<=128; <0.5 % of the forward's FLOPs), so this kernel
// keeps everything on-chip and exact in fp32: K and V of one (sample, head) live in LDS as f32
// (longer sequences — the fp32 parity path of the DiTs — stream them through LDS in blocks),
// each query is owned by d/16 adjacent lanes holding a 16-wide slice of q and of the output,
// scores are reduced across those lanes with xor-shuffles, softmax is online (running max / sum).
#include <stdlib.h>
#include "common.h"

struct AttnArgs {
  const void* q; const void* k; const void* v; void* out;
  int n, L, heads, d, ld_qkv, ld_out; float scale;
  int KB;     // keys per LDS block: L when the whole sequence fits (the UNets), else 8192 / d (the fp32 parity path of the DiTs)
};

template <typename T>
__global__ __launch_bounds__(256) void attn_small_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kv[];  // K[L][d], V[L][d]
  const int t = threadIdx.x;
  const int DS = a.d >> 4;            // lanes per query (1,2,4,8)
  const int QT = 256 / DS;            // queries per workgroup
  const int qtiles = (a.L + QT - 1) / QT;
  int b = blockIdx.x;
  const int qt = b % qtiles; b /= qtiles;
  const int h = b % a.heads; const int n = b / a.heads;
  float* Ks = kv; float* Vs = kv + a.KB * a.d;
  const T* kb = reinterpret_cast<const T*>(a.k) + (size_t)n * a.L * a.ld_qkv + h * a.d;
  const T* vb = reinterpret_cast<const T*>(a.v) + (size_t)n * a.L * a.ld_qkv + h * a.d;
  const int sl = t % DS;                         // my 16-wide slice of d
  const int qi = qt * QT + t / DS;               // my query
  const bool live = qi < a.L;
  float qv[16], o[16];
  const T* qp = reinterpret_cast<const T*>(a.q) + ((size_t)n * a.L + (live ? qi : 0)) * a.ld_qkv + h * a.d + sl * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) { qv[e] = Elem<T>::to_f(qp[e]) * a.scale; o[e] = 0.f; }
  float m = -INFINITY, l = 0.f;
  // keys stream through LDS in blocks of KB (one block = the whole sequence for the UNets' token counts): the online softmax
  // does not care where a block ends, so long sequences (DiT-B/4: 1024 / 4096 tokens) take the same fp32-exact path
  for (int j0 = 0; j0 < a.L; j0 += a.KB) {
    const int nk = min(a.KB, a.L - j0);
    if (j0) __syncthreads();                     // everyone is done with the previous block
    for (int i = t; i < nk * a.d; i += 256) {
      const int r = i / a.d, c = i - r * a.d;
      Ks[i] = Elem<T>::to_f(kb[(size_t)(j0 + r) * a.ld_qkv + c]);
      Vs[i] = Elem<T>::to_f(vb[(size_t)(j0 + r) * a.ld_qkv + c]);
    }
    __syncthreads();
    for (int j = 0; j < nk; ++j) {
      const float* kj = Ks + j * a.d + sl * 16;
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) s += qv[e] * kj[e];
      for (int off = 1; off < DS; off <<= 1) s += __shfl_xor(s, off, 64);
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn);
      const float p = expf(s - mn);
      l = l * corr + p;
      const float* vj = Vs + j * a.d + sl * 16;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = o[e] * corr + p * vj[e];
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    T* op = reinterpret_cast<T*>(a.out) + ((size_t)n * a.L + qi) * a.ld_out + h * a.d + sl * 16;
#pragma unroll
    for (int e = 0; e < 16; ++e) op[e] = Elem<T>::from_f(o[e] * inv);
  }
}

// attention_mfma.hip: matrix-core kernel for 16-bit dtypes, L % 16 == 0, d % 32 == 0
bool dc_attn_mfma_applicable(int dtype, int L, int d);
int dc_attn_mfma_launch(const dc_attention_params* p, hipStream_t s);
bool dc_attn_wave_applicable(const dc_attention_params* p);  // L <= 64: one wave per (sample, head) pair
int dc_attn_wave_launch(const dc_attention_params* p, hipStream_t s);
bool dc_attn_flash_applicable(const dc_attention_params* p);   // long sequences (DiT), online softmax
int dc_attn_flash_launch(const dc_attention_params* p, hipStream_t s);

extern "C" int dc_attention(const dc_attention_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->q && p->k && p->v && p->out, DC_ERR_ARG, "dc_attention: null pointer");
  DC_REQUIRE(p->d == 16 || p->d == 32 || p->d == 64 || p->d == 128, DC_ERR_SHAPE, "dc_attention: head dim %d (16/32/64/128)", p->d);
  DC_REQUIRE(p->n > 0 && p->L > 0 && p->heads > 0, DC_ERR_SHAPE, "dc_attention: n/L/heads");
  DC_REQUIRE(p->ld_qkv >= p->heads * p->d && p->ld_out >= p->heads * p->d, DC_ERR_SHAPE, "dc_attention: ld");
  // the matrix-core kernels keep the running max of the RAW scores and fold the scale into the exponent's FMA: valid for scale > 0 only
  DC_REQUIRE(p->scale > 0.f, DC_ERR_ARG, "dc_attention: scale must be positive (got %g)", (double)p->scale);
  // L <= 64: one wave per pair (attn_wave_kernel); up to 128: the whole-sequence matrix-core kernel; beyond: the flash kernel, which
  // also wins at 256 tokens (CheXpert 16x16 level, d = 64: 2.37 -> 0.71 ms per step; IPMSA: 13.2 -> 4.8 ms)
  constexpr int mfma_maxl = 128;
  if (dc_attn_wave_applicable(p)) return dc_attn_wave_launch(p, reinterpret_cast<hipStream_t>(stream));
  if (p->L <= mfma_maxl && dc_attn_mfma_applicable(p->dtype, p->L, p->d)) return dc_attn_mfma_launch(p, reinterpret_cast<hipStream_t>(stream));
  const size_t lds_all = (size_t)2 * p->L * p->d * sizeof(float);
  if ((lds_all > 160 * 1024 || p->L > mfma_maxl) && dc_attn_flash_applicable(p))
    return dc_attn_flash_launch(p, reinterpret_cast<hipStream_t>(stream));
  // fp32 (the parity path) and shapes the matrix-core kernels do not take: exact fp32 kernel; K / V stay whole in LDS when they
  // fit 160 KiB (one block: the order of operations of the UNet parity path is unchanged), else stream in 64 KiB blocks
  const int KB = lds_all <= 160 * 1024 ? p->L : 8192 / p->d;
  const size_t lds = (size_t)2 * KB * p->d * sizeof(float);
  AttnArgs a{p->q, p->k, p->v, p->out, p->n, p->L, p->heads, p->d, p->ld_qkv, p->ld_out, p->scale, KB};
  const int DS = p->d / 16, QT = 256 / DS, qtiles = (p->L + QT - 1) / QT;
  const long long nb = (long long)p->n * p->heads * qtiles;
  DC_REQUIRE(nb < (1LL << 31), DC_ERR_SHAPE, "dc_attention: grid too large");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)nb), blk(256);
#define DC_ATTN_LAUNCH(T)                                                                                  \
  do {                                                                                                     \
    static bool done = false;                                                                              \
    if (!done) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_small_kernel<T>),                             \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      done = true;                                                                                         \
    }                                                                                                      \
    hipLaunchKernelGGL((attn_small_kernel<T>), grid, blk, lds, s, a);                                      \
  } while (0)
  if (p->dtype == DC_F32) DC_ATTN_LAUNCH(float);
  else if (p->dtype == DC_BF16) DC_ATTN_LAUNCH(__bf16);
  else if (p->dtype == DC_F16) DC_ATTN_LAUNCH(_Float16);
  else { dc_set_error("dc_attention: dtype %d", p->dtype); return DC_ERR_DTYPE; }
#undef DC_ATTN_LAUNCH
  return dc_check_launch("dc_attention");
}
