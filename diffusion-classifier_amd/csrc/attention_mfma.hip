// attention_mfma.hip — self-attention on the matrix cores for the UNet token counts (16-bit dtypes).
//
// One 4-wave workgroup per group of (sample, head) pairs: 1 pair when L >= 64, 2 when L == 32,
// 4 when L == 16, so every wave owns whole 16-query tiles.  K [L][d] and V^T [d][L] of a pair
// live in LDS (V is transposed while it is staged, so both MFMA operands are k-contiguous);
// Q fragments come straight from global memory.  Per 16-query tile:
//   S = Q K^T   (L/16 x d/32 MFMA 16x16x32, fp32 accumulate; lane holds 4 queries x 1 key per tile)
//   softmax over keys in fp32 registers (max/sum: across tiles, then xor-shuffles over the 16 lanes
//   that share a query), P -> 16-bit into a per-wave LDS strip in [query][key] order
//   O = P V     (d/16 x L/32 MFMA), scaled by 1/rowsum, stored as 16-bit.
// The fp32 kernel of attention.hip stays the path for f32 (exact) and for shapes outside
// L % 16 == 0, d % 32 == 0.
#include <stdlib.h>
#include <type_traits>
#include "igemm_common.h"

struct AttnMArgs {
  const void* q; const void* k; const void* v; void* out;
  int n, L, heads, d, ld_qkv, ld_out, G, Lp; float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const AttnMArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int L = a.L, d = a.d, Lp = a.Lp, G = a.G;
  const int KS = d + 8, VS = Lp + 8, PS = Lp + 8;                // row strides in elements (+16 B pad)
  const int pair_bytes = (L * KS + d * VS) * 2;
  const int wpp = 4 / G;                                        // waves per pair
  const int g = wave / wpp, wl = wave - g * wpp;
  const long long pair = (long long)blockIdx.x * G + g;
  const long long npairs = (long long)a.n * a.heads;
  const bool pair_ok = pair < npairs;
  const int n = (int)((pair_ok ? pair : 0) / a.heads), h = (int)((pair_ok ? pair : 0) % a.heads);
  T* Ks = reinterpret_cast<T*>(smem + g * pair_bytes);
  T* Vt = Ks + L * KS;
  T* Pw = reinterpret_cast<T*>(smem + G * pair_bytes) + wave * 16 * PS;
  const T* qg = reinterpret_cast<const T*>(a.q) + (size_t)n * L * a.ld_qkv + h * d;
  const T* kg = reinterpret_cast<const T*>(a.k) + (size_t)n * L * a.ld_qkv + h * d;
  const T* vg = reinterpret_cast<const T*>(a.v) + (size_t)n * L * a.ld_qkv + h * d;

  // ---- stage K (row-major) and V^T into LDS, cooperatively by the waves of the pair ----
  const int tl = wl * 64 + lane, nth = wpp * 64;
  const int cpr = d / 8;                                        // 16-byte chunks per K/V row
  for (int idx = tl; idx < L * cpr; idx += nth) {
    const int r = idx / cpr, c = idx - r * cpr;
    const chunk16 kc = *reinterpret_cast<const chunk16*>(kg + (size_t)r * a.ld_qkv + c * 8);
    *reinterpret_cast<chunk16*>(Ks + r * KS + c * 8) = kc;
    const chunk16 vc = *reinterpret_cast<const chunk16*>(vg + (size_t)r * a.ld_qkv + c * 8);
    const typename Elem<T>::vec ve = __builtin_bit_cast(typename Elem<T>::vec, vc);
#pragma unroll
    for (int e = 0; e < 8; ++e) Vt[(c * 8 + e) * VS + r] = ve[e];
  }
  for (int idx = tl; idx < d * (Lp - L); idx += nth) {          // zero the padded keys (L == 16 -> Lp == 32)
    const int r = idx / (Lp - L), c = L + idx - r * (Lp - L);
    Vt[r * VS + c] = Elem<T>::from_f(0.f);
  }
  __syncthreads();
  if (!pair_ok) return;

  const int ktiles = L / 16, kblocks = d / 32;
  for (int qt = wl; qt < ktiles; qt += wpp) {
    const int q0 = qt * 16;
    chunk16 qf[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      if (kb < kblocks) qf[kb] = *reinterpret_cast<const chunk16*>(qg + (size_t)(q0 + lr) * a.ld_qkv + kb * 32 + lq * 8);
    // S tiles: queries q0 + lq*4 + r, key kt*16 + lr
    f32x4 S[16];
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      if (kt < ktiles) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (kb < kblocks) {
            const chunk16 kf = *reinterpret_cast<const chunk16*>(Ks + (kt * 16 + lr) * KS + kb * 32 + lq * 8);
            acc = Mma<T>::run(qf[kb], kf, acc);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] *= a.scale; m[r] = fmaxf(m[r], acc[r]); }
        S[kt] = acc;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) m[r] = fmaxf(m[r], __shfl_xor(m[r], o, 64));
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      if (kt < ktiles) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = expf(S[kt][r] - m[r]);
          sum[r] += p;
          Pw[(lq * 4 + r) * PS + kt * 16 + lr] = Elem<T>::from_f(p);
        }
      }
    }
    if (Lp != L) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Pw[(lq * 4 + r) * PS + L + lr] = Elem<T>::from_f(0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum[r] += __shfl_xor(sum[r], o, 64);
    __builtin_amdgcn_wave_barrier();     // P strip is private to this wave: LDS executes its ops in issue order
    // O tiles: queries q0 + lq*4 + r, channel dt*16 + lr
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      if (dt * 16 < d) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kb2 = 0; kb2 < Lp / 32; ++kb2) {
          const chunk16 pf = *reinterpret_cast<const chunk16*>(Pw + lr * PS + kb2 * 32 + lq * 8);
          const chunk16 vf = *reinterpret_cast<const chunk16*>(Vt + (dt * 16 + lr) * VS + kb2 * 32 + lq * 8);
          acc = Mma<T>::run(pf, vf, acc);
        }
        T* op = reinterpret_cast<T*>(a.out) + ((size_t)n * L + q0 + lq * 4) * a.ld_out + h * d + dt * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)r * a.ld_out] = Elem<T>::from_f(acc[r] / sum[r]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// (sample, head) pairs per workgroup.  L == 64 (8x8 tokens), d <= 64: four pairs, one wave each, instead of four waves on
// one pair — the four heads' 128-byte pieces of a qkv row are then fetched together (2.47 -> 2.67 TB/s on cfg2's blocks).
static int attn_pairs_per_wg(int L, int d) {
  if (L == 64 && d <= 64) return 4;
  return L >= 64 ? 1 : (L == 32 ? 2 : (L == 16 ? 4 : 0));
}

bool dc_attn_mfma_applicable(int dtype, int L, int d) {
  if (dtype == DC_F32) return false;
  if (L % 16 || L > 256 || d % 32 || d > 128) return false;
  const int G = attn_pairs_per_wg(L, d);
  if (!G) return false;
  const int Lp = (L + 31) / 32 * 32;
  const size_t lds = (size_t)G * (L * (d + 8) + d * (Lp + 8)) * 2 + (size_t)4 * 16 * (Lp + 8) * 2;
  return lds <= 160 * 1024;
}

int dc_attn_mfma_launch(const dc_attention_params* p, hipStream_t s) {
  AttnMArgs a;
  a.q = p->q; a.k = p->k; a.v = p->v; a.out = p->out; a.n = p->n; a.L = p->L; a.heads = p->heads; a.d = p->d;
  a.ld_qkv = p->ld_qkv; a.ld_out = p->ld_out; a.scale = p->scale;
  a.G = attn_pairs_per_wg(p->L, p->d);
  a.Lp = (p->L + 31) / 32 * 32;
  const size_t lds = (size_t)a.G * (a.L * (a.d + 8) + a.d * (a.Lp + 8)) * 2 + (size_t)4 * 16 * (a.Lp + 8) * 2;
  const long long npairs = (long long)p->n * p->heads;
  const long long nb = (npairs + a.G - 1) / a.G;
  if (nb >= (1LL << 31)) { dc_set_error("dc_attention: grid too large"); return DC_ERR_SHAPE; }
  if (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v) & 15 || (p->ld_qkv % 8)) {
    dc_set_error("dc_attention: q/k/v must be 16-byte aligned with ld %% 8 == 0");
    return DC_ERR_ALIGN;
  }
  static bool done_b = false, done_h = false;
  if (p->dtype == DC_BF16) {
    if (!done_b) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done_b = true; }
    hipLaunchKernelGGL((attn_mfma_kernel<__bf16>), dim3((unsigned)nb), dim3(256), lds, s, a);
  } else {
    if (!done_h) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done_h = true; }
    hipLaunchKernelGGL((attn_mfma_kernel<_Float16>), dim3((unsigned)nb), dim3(256), lds, s, a);
  }
  return dc_check_launch("dc_attention(mfma)");
}

// ------------------------------------------------------------------------------------------------
// Long sequences (DiT: 1024 / 4096 tokens; 256 tokens of the CheXpert / IPMSA UNets)
struct FlashArgs {
  const void* q; const void* k; const void* v; void* out;
  int n, L, heads, d, ld_qkv, ld_out; float scale;
};

// ------------------------------------------------------------------------------------------------
// Flash forward, transposed-score form (head dims 32 / 64 / 128): no transposed V image, no LDS round trip for P.
//   S^T = K Q^T     16x16x32 MFMA, A = K rows straight from a row-major LDS image, B = Q fragments held in registers;
//                   the D fragment gives a lane 4 consecutive KEYS of ONE query (column lane&15)
//   softmax         per query = per lane column: running max / sum live in every lane of the column, the cross-lane part
//                   is two xor-shuffles (lanes 16 and 32 apart); exponentials as v_exp_f32 on log2-scaled scores
//   O^T += V^T P^T  16x16x32 MFMA over PAIRS of key tiles: the two packed P^T D-fragments a lane holds ARE its B operand
//                   (the k order inside an MFMA is free as long as A agrees), and the matching A operand (4 + 4
//                   consecutive keys of one d column) is two ds_read_b64_tr_b16 of the ROW-MAJOR V image
// K / V blocks of 128 / 64 / 32 keys (d = 32 / 64 / 128) are register-staged (global loads of block i+1 issued before the MFMAs of block i, written to
// the other LDS buffer after them), one barrier per block.  (The first version — transposed V staged with 2-byte LDS writes, P through
// LDS, no prefetch — ran DiT-B/4 attention at 0.2 PF and was removed in round 4.)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

template <typename T> struct Mma16;
template <> struct Mma16<__bf16> {
  static __device__ __forceinline__ f32x4 run(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
};
template <> struct Mma16<_Float16> {
  static __device__ __forceinline__ f32x4 run(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
};

// (two workgroups per CU.  Three — launch bounds of 168 registers — were measured slower for D = 64: 10 spilled registers, 442 against
//  507 TFLOP/s on the DiT-B/4 shape, tools/bench_attention.py)
// Reductions over the four lanes that share a query column in the transposed-score form (lanes lr, lr + 16, lr + 32, lr + 48) on the
// VALU: v_permlane16_swap_b32 exchanges the odd 16-lane rows of one operand with the even rows of the other, v_permlane32_swap_b32 the
// upper half of one with the lower half of the other — with both operands the same register the two results are "rows 0 0 2 2" /
// "rows 1 1 3 3" and "lower lower" / "upper upper", so an op over each pair is the xor-16 / xor-32 butterfly.  The ds_bpermute shuffles
// these replace were four LDS round trips per query tile and key block in front of the exponentials.
// (as instructions, not through __builtin_amdgcn_permlane{16,32}_swap: hipcc 7.2 folds the builtin's two results into one once they meet in
//  an add or a max — it emitted v_add_f32 v, a0, a0 for a0 + a1, with the same or with different operands — which the hardware does not
//  do: tools/dev/permlane_probe.hip prints what the instruction returns.  The two wait states in front cover a VALU write of the
//  operands, as the compiler places them in front of its own.)
template <bool WIDE> static __device__ __forceinline__ void lane_swap(float x, float& lo, float& hi) {
  unsigned u = __builtin_bit_cast(unsigned, x), v = u;
  if constexpr (WIDE) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(v));
  lo = __builtin_bit_cast(float, u); hi = __builtin_bit_cast(float, v);
}
static __device__ __forceinline__ float col4_max(float x) {
  float a, b;
  lane_swap<false>(x, a, b);
  lane_swap<true>(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
static __device__ __forceinline__ float col4_sum(float x) {
  float a, b;
  lane_swap<false>(x, a, b);
  lane_swap<true>(a + b, a, b);
  return a + b;
}

template <typename T, int D>
__global__ __launch_bounds__(256, 2) void attn_flash_t_kernel(const FlashArgs a) {
  constexpr int KB = D <= 32 ? 128 : (D <= 64 ? 64 : 32);     // keys per block: sized so that scores + staged K/V fit the register file
  constexpr int NKT = KB / 16, NQT = 2, NDT = D / 16, NKB = D / 32;
  constexpr int PITCH = D + 8;                                // LDS row pitch in elements (+16 B)
  constexpr int CPR = D / 8;                                  // 16-byte chunks per row
  constexpr int NST = KB * CPR / 256;                         // staging chunks per thread and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* const Kl = reinterpret_cast<T*>(smem);                   // [2][KB][PITCH]
  T* const Vl = Kl + 2 * KB * PITCH;                          // [2][KB][PITCH]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int L = a.L;
  const int qblocks = (L + 127) / 128;
  int b = blockIdx.x;
  const int qb = b % qblocks; b /= qblocks;
  const int h = b % a.heads, n = b / a.heads;
  const T* qg = reinterpret_cast<const T*>(a.q) + (size_t)n * L * a.ld_qkv + h * D;
  const T* kg = reinterpret_cast<const T*>(a.k) + (size_t)n * L * a.ld_qkv + h * D;
  const T* vg = reinterpret_cast<const T*>(a.v) + (size_t)n * L * a.ld_qkv + h * D;
  const int q0 = qb * 128 + wave * 32;

  chunk16 qf[NQT][NKB];                                       // B operand of S^T: query lr, d = 32 kb + 8 lq .. +7
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int qi = q0 + qt * 16 + lr;
      qf[qt][kb] = *reinterpret_cast<const chunk16*>(qg + (size_t)(qi < L ? qi : L - 1) * a.ld_qkv + kb * 32 + lq * 8);
    }
  f32x4 O[NQT][NDT];                                          // O^T: rows d = 16 dt + 4 lq + r, column = query lr
  float m[NQT], l[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    m[qt] = -INFINITY; l[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) O[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float sc2 = a.scale * 1.4426950408889634f;            // scores in log2 units
  // The softmax keeps the running max of the RAW scores (the scale is positive: same arg max) and forms p = exp2(s * sc2 - m * sc2) as ONE
  // fused multiply-add per score in front of v_exp_f32; keys past L are masked only in a block that holds some (the DiT token counts have
  // none).  Per 64-key block and wave the kernel issues 34 v_exp_f32 (16 cycles each) and ~120 other VALU instructions against 512 cycles
  // of MFMA; the scale multiply, the subtraction and the per-score select were another ~90 (467 -> 507 TFLOP/s f16, 529 -> 578 bf16 on
  // 1024 tokens x 12 heads x 64; the serial MFMA -> max -> shuffle -> exp -> pack -> MFMA chain of a wave is what remains).

  chunk16 ks[NST], vs[NST];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = i * 256 + t, r = idx / CPR, c = idx - r * CPR;
      const int key = k0 + r;
      ks[i] = chunk16{0u, 0u, 0u, 0u}; vs[i] = ks[i];
      if (key < L) {
        ks[i] = *reinterpret_cast<const chunk16*>(kg + (size_t)key * a.ld_qkv + c * 8);
        vs[i] = *reinterpret_cast<const chunk16*>(vg + (size_t)key * a.ld_qkv + c * 8);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = i * 256 + t, r = idx / CPR, c = idx - r * CPR;
      *reinterpret_cast<chunk16*>(Kl + (buf * KB + r) * PITCH + c * 8) = ks[i];
      *reinterpret_cast<chunk16*>(Vl + (buf * KB + r) * PITCH + c * 8) = vs[i];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int nblk = (L + KB - 1) / KB;
  // one key block; RAGGED: it holds keys past L (only the last block of a sequence that is not a multiple of KB: its own copy of the
  // body behind the loop — as a condition inside one body hipcc turns the mask back into per-score selects)
  auto run_block = [&](int ib, auto raggedc) {
    constexpr bool ragged = decltype(raggedc)::value;
    const int k0 = ib * KB, buf = ib & 1;
    if (ib + 1 < nblk) fetch(k0 + KB);                        // lands under this block's MFMAs
    const T* Kb = Kl + buf * KB * PITCH;
    const T* Vb = Vl + buf * KB * PITCH;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      f32x4 S[NKT];
      float mx = m[qt];                                       // running max of the raw scores
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          const chunk16 kf = *reinterpret_cast<const chunk16*>(Kb + (kt * 16 + lr) * PITCH + kb * 32 + lq * 8);
          acc = Mma<T>::run(kf, qf[qt][kb], acc);             // rows = keys, column = query
        }
        if constexpr (ragged) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + lq * 4 + r >= L) acc[r] = -INFINITY;     // keys past L never win the max nor add to the sum
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[r]);
        S[kt] = acc;
      }
      mx = col4_max(mx);
      const float nms = -mx * sc2;
      const float corr = __builtin_amdgcn_exp2f(__builtin_fmaf(m[qt], sc2, nms));    // exp2(-inf) = 0 on the first block
      m[qt] = mx;
      float ps = 0.f;
      s16x4 P[NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], sc2, nms)); ps += pv[r]; }
        typename Elem<T>::vec4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = Elem<T>::from_f(pv[r]);
        P[kt] = __builtin_bit_cast(s16x4, pk);
      }
      l[qt] = l[qt] * corr + ps;                              // per-lane partial row sum (corr is the same in the column's four lanes): reduced once, behind the loop
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        f32x4 acc = O[qt][dt];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= corr;
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
          // one 16x16x32 MFMA per PAIR of key tiles: the sum over k is order-free as long as both operands agree, so lane
          // group lq takes as its 8 k-slots the keys 4 lq .. +3 of tile 2 kp and of tile 2 kp + 1 — exactly the two packed P^T
          // D-fragments it already holds (B operand), and two transposed reads of the row-major V image (A operand): for each,
          // the 16-lane group takes keys 16 kt + 4 lq .. +3 x d columns 16 dt .. +15; lane 4q+p supplies the address of key
          // row q, columns 4p .. 4p+3 and receives column lr of the four rows
          s16x4 vf[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const T* vp = Vb + ((2 * kp + u) * 16 + lq * 4 + (lr >> 2)) * PITCH + dt * 16 + (lr & 3) * 4;
            vf[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(const_cast<T*>(vp)));
          }
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 av = __builtin_shufflevector(vf[0], vf[1], 0, 1, 2, 3, 4, 5, 6, 7);
          const s16x8 bv = __builtin_shufflevector(P[2 * kp], P[2 * kp + 1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc = Mma<T>::run(__builtin_bit_cast(chunk16, av), __builtin_bit_cast(chunk16, bv), acc);   // rows = d, column = query
        }
        O[qt][dt] = acc;
      }
    }
    if (ib + 1 < nblk) stash(buf ^ 1);
    __syncthreads();
  };
  const int nfull = L / KB;
  for (int ib = 0; ib < nfull; ++ib) run_block(ib, std::false_type{});
  if (nfull < nblk) run_block(nfull, std::true_type{});
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    const int qi = q0 + qt * 16 + lr;
    const float inv = 1.0f / col4_sum(l[qt]);                 // (all lanes take part in the swaps: before the bounds test)
    if (qi >= L) continue;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      typename Elem<T>::vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = Elem<T>::from_f(O[qt][dt][r] * inv);
      *reinterpret_cast<typename Elem<T>::vec4*>(reinterpret_cast<T*>(a.out) + ((size_t)n * L + qi) * a.ld_out + h * D + dt * 16 + lq * 4) = o;
    }
  }
}

template <typename T, int D>
static int launch_flash_t(const FlashArgs& a, long long nb, hipStream_t s) {
  constexpr int KB = D <= 32 ? 128 : (D <= 64 ? 64 : 32);
  constexpr size_t lds = (size_t)4 * KB * (D + 8) * 2;
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_flash_t_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  hipLaunchKernelGGL((attn_flash_t_kernel<T, D>), dim3((unsigned)nb), dim3(256), lds, s, a);
  return dc_check_launch("dc_attention(flash_t)");
}

// ------------------------------------------------------------------------------------------------
// Token counts of the UNets' deep levels (L = 16 ... 64: 4x4 / 8x8 images): ONE WAVE per (sample, head) pair, four pairs per
// workgroup, nothing shared between waves and therefore no workgroup barrier.  Same arithmetic as attn_flash_t_kernel with a
// single key block: K and V of the pair are staged row-major (16-byte chunks) in a wave-private LDS strip; S^T = K Q^T
// (D-fragment = 4 consecutive keys of one query per lane), softmax in registers on log2-scaled scores (v_exp_f32, two
// xor-shuffles per query), the packed P^T D-fragments fed back as the B operand of O^T += V^T P^T with the A operand read by
// ds_read_b64_tr_b16, 8-byte output stores.  The first matrix-core kernel for these shapes (attn_mfma_kernel above: V transposed
// by 2-byte LDS writes, P through LDS, 2-byte strided output stores, one workgroup barrier) ran them at 2.5 - 2.9 TB/s of the
// q/k/v + output bytes they are bound by.
template <typename T, int D, int NKT>                          // NKT: 16-key tiles per pair (even; L <= 16 NKT, keys past L masked)
__global__ __launch_bounds__(256) void attn_wave_kernel(const FlashArgs a) {
  constexpr int LP = NKT * 16, NDT = D / 16, NKB = D / 32;
  constexpr int PITCH = D + 8, CPR = D / 8;
  constexpr int NCH = LP * CPR / 64;                           // staging chunks per lane and operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int L = a.L;
  const long long pair = (long long)blockIdx.x * 4 + wave;
  if (pair >= (long long)a.n * a.heads) return;               // whole waves leave: nothing below is shared between waves
  const int n = (int)(pair / a.heads), h = (int)(pair % a.heads);
  T* const Kl = reinterpret_cast<T*>(smem) + (size_t)wave * 2 * LP * PITCH;
  T* const Vl = Kl + LP * PITCH;
  const T* qg = reinterpret_cast<const T*>(a.q) + (size_t)n * L * a.ld_qkv + h * D;
  const T* kg = reinterpret_cast<const T*>(a.k) + (size_t)n * L * a.ld_qkv + h * D;
  const T* vg = reinterpret_cast<const T*>(a.v) + (size_t)n * L * a.ld_qkv + h * D;
  // ---- every query fragment of the pair up front (B operand of S^T: query lr, d = 32 kb + 8 lq .. +7): their latency then sits
  // under the K / V staging instead of in front of each query tile ----
  chunk16 qfa[NKT][NKB];
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int qi = qt * 16 + lr;
      qfa[qt][kb] = *reinterpret_cast<const chunk16*>(qg + (size_t)(qi < L ? qi : L - 1) * a.ld_qkv + kb * 32 + lq * 8);
    }
  // ---- stage K and V (rows past L as zeros: a masked key has P = 0, and 0 x garbage must not be a NaN) ----
#pragma unroll
  for (int i0 = 0; i0 < NCH; i0 += 8) {
    chunk16 kc[8], vc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u < NCH) {
        const int idx = (i0 + u) * 64 + lane, r = idx / CPR, c = idx - r * CPR;
        kc[u] = chunk16{0u, 0u, 0u, 0u}; vc[u] = kc[u];
        if (r < L) {
          kc[u] = *reinterpret_cast<const chunk16*>(kg + (size_t)r * a.ld_qkv + c * 8);
          vc[u] = *reinterpret_cast<const chunk16*>(vg + (size_t)r * a.ld_qkv + c * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u < NCH) {
        const int idx = (i0 + u) * 64 + lane, r = idx / CPR, c = idx - r * CPR;
        *reinterpret_cast<chunk16*>(Kl + r * PITCH + c * 8) = kc[u];
        *reinterpret_cast<chunk16*>(Vl + r * PITCH + c * 8) = vc[u];
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float sc2 = a.scale * 1.4426950408889634f;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int nqt = (L + 15) >> 4;
#pragma unroll
  for (int qt = 0; qt < NKT; ++qt) {                           // (unrolled: qfa is indexed statically; the bound is wave-uniform)
    if (qt >= nqt) break;
    const int qi = qt * 16 + lr;
    chunk16 qf[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) qf[kb] = qfa[qt][kb];
    f32x4 S[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const chunk16 kf = *reinterpret_cast<const chunk16*>(Kl + (kt * 16 + lr) * PITCH + kb * 32 + lq * 8);
        acc = Mma<T>::run(kf, qf[kb], acc);                   // rows = keys, column = query
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kvalid = kt * 16 + lq * 4 + r < L;
        acc[r] = kvalid ? acc[r] : -INFINITY;                 // raw scores: the scale is positive, it enters with the max below
        mx = fmaxf(mx, acc[r]);
      }
      S[kt] = acc;
    }
    mx = col4_max(mx);
    const float nms = -mx * sc2;
    float ps = 0.f;
    s16x4 P[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], sc2, nms)); ps += pv[r]; }
      typename Elem<T>::vec4 pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = Elem<T>::from_f(pv[r]);
      P[kt] = __builtin_bit_cast(s16x4, pk);
    }
    ps = col4_sum(ps);
    const float inv = __builtin_amdgcn_rcpf(ps);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kp = 0; kp < NKT / 2; ++kp) {
        s16x4 vf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const T* vp = Vl + ((2 * kp + u) * 16 + lq * 4 + (lr >> 2)) * PITCH + dt * 16 + (lr & 3) * 4;
          vf[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(const_cast<T*>(vp)));
        }
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 av = __builtin_shufflevector(vf[0], vf[1], 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8 bv = __builtin_shufflevector(P[2 * kp], P[2 * kp + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        acc = Mma<T>::run(__builtin_bit_cast(chunk16, av), __builtin_bit_cast(chunk16, bv), acc);   // rows = d, column = query
      }
      if (qi < L) {
        typename Elem<T>::vec4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = Elem<T>::from_f(acc[r] * inv);
        *reinterpret_cast<typename Elem<T>::vec4*>(reinterpret_cast<T*>(a.out) + ((size_t)n * L + qi) * a.ld_out + h * D + dt * 16 + lq * 4) = o;
      }
    }
  }
}

template <typename T, int D, int NKT>
static int launch_attn_wave(const FlashArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)4 * 2 * NKT * 16 * (D + 8) * 2;
  static bool done = false;
  if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_wave_kernel<T, D, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
  const long long nb = ((long long)a.n * a.heads + 3) / 4;
  if (nb >= (1LL << 31)) { dc_set_error("dc_attention: grid too large"); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL((attn_wave_kernel<T, D, NKT>), dim3((unsigned)nb), dim3(256), lds, s, a);
  return dc_check_launch("dc_attention(wave)");
}

// one wave per pair: 16-bit, L <= 64, d in {32, 64, 128}, 8-byte aligned output rows
bool dc_attn_wave_applicable(const dc_attention_params* p) {
  if (p->dtype == DC_F32 || p->L > 64 || !(p->d == 32 || p->d == 64 || p->d == 128)) return false;
  if ((((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v) & 15) || (p->ld_qkv % 8)) return false;
  return p->ld_out % 4 == 0 && (((uintptr_t)p->out) & 7) == 0;
}

int dc_attn_wave_launch(const dc_attention_params* p, hipStream_t s) {
  FlashArgs a{p->q, p->k, p->v, p->out, p->n, p->L, p->heads, p->d, p->ld_qkv, p->ld_out, p->scale};
  const bool bf = p->dtype == DC_BF16;
  const bool two = p->L <= 32;                                 // 2 or 4 key tiles (an even count: the P.V MFMAs take key tiles in pairs)
#define DC_AW(D) (two ? (bf ? launch_attn_wave<__bf16, D, 2>(a, s) : launch_attn_wave<_Float16, D, 2>(a, s)) \
                      : (bf ? launch_attn_wave<__bf16, D, 4>(a, s) : launch_attn_wave<_Float16, D, 4>(a, s)))
  if (p->d == 32) return DC_AW(32);
  if (p->d == 64) return DC_AW(64);
  return DC_AW(128);
#undef DC_AW
}

bool dc_attn_flash_applicable(const dc_attention_params* p) {
  return p->dtype != DC_F32 && (p->d == 32 || p->d == 64 || p->d == 128) && p->L >= 1 && p->ld_out % 4 == 0 && (((uintptr_t)p->out) & 7) == 0 &&
         ((((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v) & 15) == 0) && p->ld_qkv % 8 == 0;
}

int dc_attn_flash_launch(const dc_attention_params* p, hipStream_t s) {
  FlashArgs a{p->q, p->k, p->v, p->out, p->n, p->L, p->heads, p->d, p->ld_qkv, p->ld_out, p->scale};
  const long long nb = (long long)p->n * p->heads * ((p->L + 127) / 128);
  if (nb >= (1LL << 31)) { dc_set_error("dc_attention: grid too large"); return DC_ERR_SHAPE; }
  const bool bf = p->dtype == DC_BF16;
  if (p->d == 32) return bf ? launch_flash_t<__bf16, 32>(a, nb, s) : launch_flash_t<_Float16, 32>(a, nb, s);
  if (p->d == 64) return bf ? launch_flash_t<__bf16, 64>(a, nb, s) : launch_flash_t<_Float16, 64>(a, nb, s);
  return bf ? launch_flash_t<__bf16, 128>(a, nb, s) : launch_flash_t<_Float16, 128>(a, nb, s);
}
