// tblock.hip — the attention half of a UNet transformer block in ONE launch, activations on chip (16-bit, 64 tokens x 256 channels:
// the 8x8-level blocks of the CIFAR UNet; the backbone behind /root/reference/nets/unet.py:186-195, diffusers BasicTransformerBlock):
//
//   h   = proj_in(x) + b                       x: the GroupNorm'd block input [n][64][C]
//   hn  = LayerNorm(h) * gamma + beta
//   q, k, v = hn Wq^T, hn Wk^T, hn Wv^T        (no bias), heads of 32 or 64 channels
//   o   = softmax(q k^T * scale) v             per (sample, head)
//   out = (to_out(o) + b_o) + rowvec[sample] + h        rowvec: the cross-attention's per-class vector (one context token)
//
// Launched separately these are five GEMM / attention / LayerNorm passes that move each 32 KiB sample through HBM ten times at
// 2-4 TB/s (1.1 ms per block and cfg2 step for 0.4 TFLOP).  Here one 4-wave workgroup owns one sample:
//   * two LDS images [64 tokens][C] (row pitch + 16 B: the 16 rows of a fragment read cover the 64 banks once) hold x -> hn -> o and
//     h -> out; a sample enters and leaves HBM once, as whole rows;
//   * wave w owns output channels 64 w .. + 63 of EVERY GEMM (one head of 64 channels or two of 32): its weight rows come straight from
//     L2 into registers as MFMA A-operand fragments (16 B per lane, two k-steps ahead), never through LDS — no other wave of the
//     workgroup wants them;
//   * every GEMM is D[cout][token] = W[cout][:] . X[token][:] (16x16x32 MFMA, A = W rows, B = X rows from the LDS image), so a lane
//     ends up with 4 consecutive channels of one token.  Packed in pairs of channel fragments those ARE the operands of the score
//     product (the k order inside an MFMA is free as long as A and B agree): S^T = K Q^T needs no LDS round trip.  V is computed with
//     the operands swapped (D[token][d]): a lane then holds 4 consecutive KEYS of one d column — the A operand of O^T += V^T P^T
//     beside the packed P^T fragments of the softmax (the trick of attn_flash_t_kernel, without its transposed LDS reads).
// Two workgroups per CU (71 KiB of LDS, <= 256 registers): one's LayerNorm / softmax / row stores run under the other's MFMAs.
// Rounding points are those of the separate launches (h, hn, q, k, v, p, o and out rounded to the 16-bit type where they were stored).
#include <stdlib.h>
#include <utility>
#include "igemm_common.h"

#ifdef DC_STAMPS
// diagnostic build only (tools/stamp_tblock.py): s_memtime stamps of thread 0 of every workgroup, 16 per workgroup; never compiled into the shipped library
static __device__ unsigned long long* g_tb_stamps;
extern "C" void dc_debug_set_tb_stamps(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tb_stamps), &p, sizeof(p)); }
#define TB_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && g_tb_stamps) g_tb_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TB_STAMP(k) do {} while (0)
#endif

namespace {
struct TbArgs {
  const void* x; const void* Wp; const float* bp;
  const float* ln_g; const float* ln_b;
  const void* Wqkv; const void* Wo; const float* bo;
  const float* rowvec; const int32_t* rowvec_map;
  void* out;
  int n, ldx, ld_out, rowvec_ld;
  float ln_eps, scale;
};

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// xor-16 / xor-32 butterflies over the four lanes that share a query column (attention_mfma.hip has the story of the builtin)
template <bool WIDE> __device__ __forceinline__ void tb_lane_swap(float x, float& lo, float& hi) {
  unsigned u = __builtin_bit_cast(unsigned, x), v = u;
  if constexpr (WIDE) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(v));
  lo = __builtin_bit_cast(float, u); hi = __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float tb_col4_max(float x) {
  float a, b;
  tb_lane_swap<false>(x, a, b);
  tb_lane_swap<true>(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float tb_col4_sum(float x) {
  float a, b;
  tb_lane_swap<false>(x, a, b);
  tb_lane_swap<true>(a + b, a, b);
  return a + b;
}

template <typename T> __device__ __forceinline__ u32x2 tb_pack4(const f32x4 v) {
  typename Elem<T>::vec4 p;
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] = Elem<T>::from_f(v[r]);
  return __builtin_bit_cast(u32x2, p);
}
__device__ __forceinline__ chunk16 tb_join(const u32x2 a, const u32x2 b) { return chunk16{a[0], a[1], b[0], b[1]}; }

// a loop whose index is a compile-time constant in every iteration (hipcc keeps a 40-step `#pragma unroll` loop of this size rolled and
// then indexes the fragment rings dynamically: scratch)
template <typename F, int... I> __device__ __forceinline__ void tb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void tb_static_for(F&& f) { tb_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS visibility inside the workgroup without draining the weight fragments in flight (a __syncthreads would wait vmcnt(0))
__device__ __forceinline__ void tb_lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
}  // namespace

// D: channels per head (32 or 64); a wave's 64 channels are 64 / D whole heads.
// (Two samples per eight-wave workgroup — waves w and w + 4 asking for the same weight fragments, the second request served by the CU's
//  L1 — was built and measured: 0.75 against 0.65 ms per 8000 samples.  One workgroup per CU runs its eight waves through the same
//  phase at the same time; two independent ones hide each other's LayerNorm / softmax / row stores.)
template <typename T, int D>
__global__ __launch_bounds__(256, 2) void tblock_front_kernel(const TbArgs a) {
  constexpr int C = 256, L = 64, PB = C * 2 + 16;     // channels, tokens, LDS row pitch in bytes
  constexpr int NKC = C / 32, NST = 5 * NKC;          // k-steps per GEMM; steps of the five GEMMs: proj_in, K, V, Q, to_out
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const img0 = smem;                            // x -> hn -> o
  char* const img1 = img0 + L * PB;                   // h -> out
  // fp32 vectors every wave reads in its epilogues, staged once (fetched per use they were 2.5 - 10 k cycles of exposed global-load
  // latency in front of the LayerNorm and the last epilogue): proj_in bias | LayerNorm gamma | beta | to_out bias | the sample's class vector
  float* const tab = reinterpret_cast<float*>(smem + 2 * L * PB);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int n = blockIdx.x;
  TB_STAMP(0);
  const int rvrow = a.rowvec_map ? a.rowvec_map[n] : n;        // (first: the class vector's load depends on it)

  // ---- the sample's rows into image 0 (16-byte pieces, whole rows) ----
  {
    const T* xg = reinterpret_cast<const T*>(a.x) + (size_t)n * L * a.ldx;
    chunk16 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 256 + t, row = idx >> 5, c = idx & 31;
      v[i] = *reinterpret_cast<const chunk16*>(xg + (size_t)row * a.ldx + c * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 256 + t, row = idx >> 5, c = idx & 31;
      *reinterpret_cast<chunk16*>(img0 + row * PB + c * 16) = v[i];
    }
  }
  tab[t] = a.bp[t];
  tab[C + t] = a.ln_g[t];
  tab[2 * C + t] = a.ln_b[t];
  tab[3 * C + t] = a.bo ? a.bo[t] : 0.f;
  tab[4 * C + t] = a.rowvec ? a.rowvec[(size_t)rvrow * a.rowvec_ld + t] : 0.f;

  // ---- weight stream: step s = (GEMM g = s / NKC, k-chunk s % NKC); the lane's fragment of channel block cf is 16 bytes of row
  // row0(g) + 16 cf + lr at k = 32 kc + 8 lq ----
  const T* const Wp = reinterpret_cast<const T*>(a.Wp);
  const T* const Wq = reinterpret_cast<const T*>(a.Wqkv);
  const T* const Wo = reinterpret_cast<const T*>(a.Wo);
  const int wrow = (wave * 64 + lr) * C + lq * 8;
  // two k-steps ahead (measured: 3 and 4 steps ahead, and fetching k-steps in pairs so that both halves of a 128-byte line are asked for
  // back to back, all run the launch in the same time +- 1 % — the GEMM phases sit at the ~26 B per cycle and CU that L2-resident vector
  // loads deliver with every CU streaming, whatever is in flight: DESIGN 9)
  constexpr int PD = 2, WR = PD + 1;
  chunk16 wf[WR][4];
  auto issue = [&](auto sc) {
    constexpr int s = decltype(sc)::value, g = s / NKC, kc = s % NKC;
    const T* wb = g == 0 ? Wp : (g == 4 ? Wo : Wq + (g == 1 ? C * C : (g == 2 ? 2 * C * C : 0)));
    wb += wrow + kc * 32;
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) wf[s % WR][cf] = *reinterpret_cast<const chunk16*>(wb + cf * 16 * C);
  };
  tb_static_for<PD>([&](auto sc) { issue(sc); });
  __syncthreads();
  TB_STAMP(1);

  const uint32_t xoff = lr * PB + lq * 16;            // fragment read: row 16 tf + lr, bytes 64 kc + 16 lq
  const float sc2 = a.scale * 1.4426950408889634f;
  u32x2 Kp[4][4], Vp[4][4], Qp[4][4];                 // [token fragment][channel fragment] / V: [d fragment][token fragment]
  f32x4 acc[4][4];

  tb_static_for<NST>([&](auto sc) {
    constexpr int s = decltype(sc)::value, g = s / NKC, kc = s % NKC;
    if constexpr (kc == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (s + PD < NST) issue(std::integral_constant<int, s + PD>{});
    __builtin_amdgcn_sched_barrier(0);          // the loads stay HERE, ahead of their MFMAs (hipcc otherwise sinks each to its first use)
    chunk16 xf[4];
#pragma unroll
    for (int tf = 0; tf < 4; ++tf) xf[tf] = *reinterpret_cast<const chunk16*>(img0 + xoff + tf * 16 * PB + kc * 64);
#pragma unroll
    for (int cf = 0; cf < 4; ++cf)
#pragma unroll
      for (int tf = 0; tf < 4; ++tf)
        acc[cf][tf] = g == 2 ? Mma<T>::run(xf[tf], wf[s % WR][cf], acc[cf][tf])      // V: rows = tokens, column = d
                             : Mma<T>::run(wf[s % WR][cf], xf[tf], acc[cf][tf]);     // rows = channels, column = token
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kc != NKC - 1) return;
    TB_STAMP(2 + 2 * g);                         // GEMM g's MFMAs issued: stamps 2, 4, 6, 8, 10

    if constexpr (g == 0) {
      // ---- h = proj_in + bias -> image 1 (rounded), then LayerNorm of its rows -> image 0 ----
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) {
        const int c = wave * 64 + cf * 16 + lq * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(tab + c);
#pragma unroll
        for (int tf = 0; tf < 4; ++tf) {
          f32x4 v = acc[cf][tf];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += b[r];
          *reinterpret_cast<u32x2*>(img1 + (tf * 16 + lr) * PB + c * 2) = tb_pack4<T>(v);
        }
      }
      tb_lds_barrier();
      {
        // four lanes per row: lane part p takes the 16-byte pieces p, p + 4, ... of its row
        const int row = t >> 2, part = t & 3;
        const char* hr = img1 + row * PB;
        float v[8][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          chunk_to_f<T>(*reinterpret_cast<const chunk16*>(hr + (part + 4 * k) * 16), v[k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += v[k][e];
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float mean = sum / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        const float rstd = rsqrtf(q / (float)C + a.ln_eps);
        char* yr = img0 + row * PB;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int ch = (part + 4 * k) * 8;
#pragma unroll
          for (int e4 = 0; e4 < 8; e4 += 4) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(tab + C + ch + e4);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(tab + 2 * C + ch + e4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = (v[k][e4 + e] - mean) * rstd;
              v[k][e4 + e] = x * gm[e] + bt[e];
            }
          }
          *reinterpret_cast<chunk16*>(yr + (part + 4 * k) * 16) = f_to_chunk<T>(v[k]);
        }
      }
      tb_lds_barrier();
      TB_STAMP(3);
    } else if constexpr (g == 1) {
#pragma unroll
      for (int tf = 0; tf < 4; ++tf)
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) Kp[tf][cf] = tb_pack4<T>(acc[cf][tf]);
    } else if constexpr (g == 2) {
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int tf = 0; tf < 4; ++tf) Vp[df][tf] = tb_pack4<T>(acc[df][tf]);
    } else if constexpr (g == 3) {
#pragma unroll
      for (int tf = 0; tf < 4; ++tf)
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) Qp[tf][cf] = tb_pack4<T>(acc[cf][tf]);
      tb_lds_barrier();                           // every wave is done reading hn: image 0 takes o
      // ---- attention of this wave's 64 / D heads (channel fragments [j D / 16, (j + 1) D / 16) of its 64 channels), in registers ----
      constexpr int FPH = D / 16;                  // channel fragments per head
#pragma unroll
      for (int j = 0; j < 64 / D; ++j) {
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          f32x4 S[4];
          float mx = -INFINITY;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < FPH / 2; ++p) {    // rows = keys, column = query; k slots: channels 4 lq .. + 3 of fragments f and f + 1
              const int f = j * FPH + 2 * p;
              sacc = Mma<T>::run(tb_join(Kp[kt][f], Kp[kt][f + 1]), tb_join(Qp[qt][f], Qp[qt][f + 1]), sacc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[r]);
            S[kt] = sacc;
          }
          mx = tb_col4_max(mx);
          const float nms = -mx * sc2;
          float ps = 0.f;
          u32x2 P[4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            f32x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], sc2, nms)); ps += pv[r]; }
            P[kt] = tb_pack4<T>(pv);
          }
          ps = tb_col4_sum(ps);
          const float inv = __builtin_amdgcn_rcpf(ps);
#pragma unroll
          for (int df = j * FPH; df < (j + 1) * FPH; ++df) {
            f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)         // rows = d, column = query; k slots: keys 4 lq .. + 3 of tiles 2 kp and 2 kp + 1
              oacc = Mma<T>::run(tb_join(Vp[df][2 * kp], Vp[df][2 * kp + 1]), tb_join(P[2 * kp], P[2 * kp + 1]), oacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[r] *= inv;
            *reinterpret_cast<u32x2*>(img0 + (qt * 16 + lr) * PB + (wave * 64 + df * 16 + lq * 4) * 2) = tb_pack4<T>(oacc);
          }
        }
      }
      tb_lds_barrier();
      TB_STAMP(9);
    } else {
      // ---- out = (to_out + bias) + rowvec + h, back into this wave's own columns of image 1 ----
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) {
        const int c = wave * 64 + cf * 16 + lq * 4;
        const f32x4 b = *reinterpret_cast<const f32x4*>(tab + 3 * C + c);
        const f32x4 rw = *reinterpret_cast<const f32x4*>(tab + 4 * C + c);
#pragma unroll
        for (int tf = 0; tf < 4; ++tf) {
          char* hp = img1 + (tf * 16 + lr) * PB + c * 2;
          const typename Elem<T>::vec4 hv = __builtin_bit_cast(typename Elem<T>::vec4, *reinterpret_cast<const u32x2*>(hp));
          f32x4 v = acc[cf][tf];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((v[r] + b[r]) + rw[r]) + Elem<T>::to_f(hv[r]);
          *reinterpret_cast<u32x2*>(hp) = tb_pack4<T>(v);
        }
      }
      tb_lds_barrier();
      TB_STAMP(11);
    }
  });

  // ---- the sample's output rows, whole rows ----
  {
    T* og = reinterpret_cast<T*>(a.out) + (size_t)n * L * a.ld_out;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 256 + t, row = idx >> 5, c = idx & 31;
      *reinterpret_cast<chunk16*>(og + (size_t)row * a.ld_out + c * 8) = *reinterpret_cast<const chunk16*>(img1 + row * PB + c * 16);
    }
  }
  TB_STAMP(12);
}

extern "C" int32_t dc_tblock_front_ok(const dc_tblock_front_params* p) {
  if (!p) return 0;
  static const bool off = getenv("DCAMD_NO_TBLOCK") != nullptr;
  if (off) return 0;
  if (p->dtype != DC_BF16 && p->dtype != DC_F16) return 0;
  if (p->L != 64 || p->C != 256 || (p->heads != 4 && p->heads != 8)) return 0;
  if (p->ldx % 8 || p->ld_out % 8) return 0;
  return 1;
}

extern "C" int dc_tblock_front(const dc_tblock_front_params* p, dc_stream stream) {
  DC_REQUIRE(p && p->x && p->Wp && p->bp && p->ln_g && p->ln_b && p->Wqkv && p->Wo && p->out, DC_ERR_ARG, "dc_tblock_front: null pointer");
  DC_REQUIRE(p->dtype == DC_BF16 || p->dtype == DC_F16, DC_ERR_DTYPE, "dc_tblock_front: 16-bit dtypes only (got %d)", p->dtype);
  DC_REQUIRE(p->L == 64 && p->C == 256 && (p->heads == 4 || p->heads == 8), DC_ERR_SHAPE,
             "dc_tblock_front: L=%d C=%d heads=%d (64 tokens x 256 channels in 4 or 8 heads only)", p->L, p->C, p->heads);
  DC_REQUIRE(p->n > 0 && p->ldx >= p->C && p->ld_out >= p->C && p->ldx % 8 == 0 && p->ld_out % 8 == 0, DC_ERR_SHAPE,
             "dc_tblock_front: n=%d ldx=%d ld_out=%d", p->n, p->ldx, p->ld_out);
  DC_REQUIRE(!p->rowvec || p->rowvec_ld >= p->C, DC_ERR_SHAPE, "dc_tblock_front: rowvec_ld=%d", p->rowvec_ld);
  DC_REQUIRE(p->scale > 0.f && p->ln_eps > 0.f, DC_ERR_ARG, "dc_tblock_front: scale and ln_eps must be positive");
  DC_REQUIRE((((uintptr_t)p->x | (uintptr_t)p->out | (uintptr_t)p->Wp | (uintptr_t)p->Wqkv | (uintptr_t)p->Wo | (uintptr_t)p->bp |
               (uintptr_t)p->bo | (uintptr_t)p->ln_g | (uintptr_t)p->ln_b | (uintptr_t)p->rowvec) & 15) == 0 && p->rowvec_ld % 4 == 0,
             DC_ERR_ALIGN, "dc_tblock_front: pointers must be 16-byte aligned");
  TbArgs a;
  a.x = p->x; a.Wp = p->Wp; a.bp = p->bp; a.ln_g = p->ln_g; a.ln_b = p->ln_b; a.Wqkv = p->Wqkv; a.Wo = p->Wo; a.bo = p->bo;
  a.rowvec = p->rowvec; a.rowvec_map = p->rowvec_map; a.out = p->out;
  a.n = p->n; a.ldx = p->ldx; a.ld_out = p->ld_out; a.rowvec_ld = p->rowvec_ld; a.ln_eps = p->ln_eps; a.scale = p->scale;
  constexpr int lds = 2 * 64 * (256 * 2 + 16) + 5 * 256 * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool bf = p->dtype == DC_BF16, h4 = p->heads == 4;
  void (*kern)(const TbArgs) = bf ? (h4 ? tblock_front_kernel<__bf16, 64> : tblock_front_kernel<__bf16, 32>)
                                  : (h4 ? tblock_front_kernel<_Float16, 64> : tblock_front_kernel<_Float16, 32>);
  static bool done[2][2] = {};
  bool& d = done[bf ? 0 : 1][h4 ? 0 : 1];
  if (!d) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); d = true; }
  hipLaunchKernelGGL(kern, dim3((unsigned)p->n), dim3(256), lds, s, a);
  return dc_check_launch("dc_tblock_front");
}
