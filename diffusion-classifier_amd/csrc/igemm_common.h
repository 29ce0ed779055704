// igemm_common.h — pieces shared by the two implicit-GEMM kernels (igemm.hip: register-staged
// 128xN tile; igemm_pipe.hip: LDS-DMA 3-stage pipelined 256x128 tile).
#pragma once
#include "common.h"

struct IgemmArgs {
  const void* src0; const int32_t* map0; const void* src1; const int32_t* map1;
  const void* W; const float* bias;
  const float* rowvec; const int32_t* rowvec_map;
  const float* gate; const int32_t* gate_map;
  const void* residual; const int32_t* res_map;
  void* out;
  const float* gn_scale; const float* gn_shift; int gn_silu;
  const void* src2; const int32_t* map2; const void* W2; int C2, ld2;   // 1x1 side source (conv3_halo only)
  float ln_eps;             // > 0: row LayerNorm of the A operand (igemm_xreg only)
  float* qstats;            // per (sample, part, channel quad) (mean, M2) of the stored output (conv3_halo only)
  // producer-side GroupNorm (epi_pn.h; conv3_halo one-image-per-patch form only): pn_out receives act(gn(v)) of the output v, `out`
  // (may then be null) the raw v; pn_cnt: one zeroed arrival counter per (sample, N tile)
  void* pn_out; const float* pn_gamma; const float* pn_beta; unsigned* pn_cnt; int pn_ld, pn_groups, pn_silu; float pn_eps;
  int C0, C1, ld0, ld1, rowvec_ld, gate_ld, res_dtype, res_ld, out_dtype, out_ld, act;
  int taps, stride, upsample, Hin, Win, Hout, Wout, Cout;
  int M, Ktot, c0chunks, cpt, nk, tiles_m, tiles_n;
  int n_fast;   // tile order: 1 = N fastest (small weight matrix: every N tile of an M tile runs back to back, X read once)
};

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
  static __device__ __forceinline__ f32x4 run(const chunk16 w, const chunk16 x, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  }
};
template <> struct Mma<_Float16> {
  static __device__ __forceinline__ f32x4 run(const chunk16 w, const chunk16 x, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // Lane group q=lane>>4 holds k = 4q..4q+3 of a 16-wide k block; MFMA e consumes element e, so
  // the four MFMAs cover all 16 k (a consistent k permutation on both operands).
  static __device__ __forceinline__ f32x4 run(const chunk16 w, const chunk16 x, f32x4 acc) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[0], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[1], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[2], xf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[3], xf[3], acc, 0, 0, 0);
    return acc;
  }
};

// LDS image of a K-step: rows of 128 B; the 16-B chunk index is XOR-swizzled by (row>>1)&7 so the
// ds_read_b128 fragment reads of 16 consecutive rows at one logical chunk hit 16 distinct slots of
// the 256-B bank row (conflict-free for every 16-lane service group).
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware, bijective block -> (tile_m, tile_n): each XCD (private L2) gets a contiguous range of the tile
// list.  Large weight matrices: M fastest inside one N panel, so blocks sharing an L2 stream the same weight
// panel.  Small ones (<= 2 MiB, resident in every L2): N fastest, so the activation tile is fetched from HBM
// once instead of once per N tile (the 16-tile GEGLU projection showed 4x the algorithmic FETCH_SIZE).
__device__ __forceinline__ void tile_of_block(const IgemmArgs& a, int& tile_m, int& tile_n) {
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  if (a.n_fast) {
    tile_m = lid / a.tiles_n;
    tile_n = lid - tile_m * a.tiles_n;
  } else {
    tile_n = lid / a.tiles_m;
    tile_m = lid - tile_n * a.tiles_m;
  }
}

// Epilogue of one wave: acc[i][j] is the 16x16 tile (cout block i, pixel block j); the lane holds 4
// consecutive couts (lq*4 + r) of pixel lr.  m0 / n0: first pixel row / first packed cout of the wave.
template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[TN][TM], int m0, int n0, int lr, int lq) {
  const int HWo = a.Hout * a.Wout;
  const bool geglu = a.act == DC_ACT_GEGLU;
  const int cout_out = geglu ? (a.Cout >> 1) : a.Cout;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + j * 16 + lr;
    if (m >= a.M) continue;
    const int n = m / HWo;
    const float* rv = a.rowvec ? a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[n] : n) * a.rowvec_ld : nullptr;
    const float* gt = a.gate ? a.gate + (size_t)(a.gate_map ? a.gate_map[n] : n) * a.gate_ld : nullptr;
    size_t rrow = 0;
    if (a.residual) rrow = (a.res_map ? (size_t)a.res_map[n] * HWo + (m - n * HWo) : (size_t)m) * a.res_ld;
    const size_t orow = (size_t)m * a.out_ld;
#pragma unroll
    for (int i = 0; i < TN; i += 1) {
      const int pc = n0 + i * 16 + lq * 4;  // packed channel index
      float v[4];
      int oc;
      if (geglu) {
        if (i & 1) continue;
        oc = ((n0 + i * 16) >> 1) + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float val = acc[i][j][r], g = acc[(i + 1) % TN][j][r];
          if (a.bias) { val += a.bias[pc + r]; g += a.bias[pc + 16 + r]; }
          v[r] = val * gelu_erf_f(g);
        }
      } else {
        oc = pc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = acc[i][j][r];
          const int c = oc + r;
          if (c < cout_out) {
            if (a.bias) x += a.bias[c];
            if (rv) x += rv[c];
            if (a.act == DC_ACT_SILU) x = silu_f(x);
            else if (a.act == DC_ACT_GELU_TANH) x = gelu_tanh_f(x);
            if (gt) x *= gt[c];
          }
          v[r] = x;
        }
      }
      if (oc >= cout_out) continue;
      if (a.residual) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (oc + r < cout_out) v[r] += load_as(a.residual, rrow + oc + r, a.res_dtype);
      }
      if (oc + 3 < cout_out && (a.out_ld & 3) == 0) {
        if (a.out_dtype == DC_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + orow + oc) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (a.out_dtype == DC_BF16) {
          __bf16 h[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.out) + orow + oc) = *reinterpret_cast<uint2*>(h);
        } else {
          _Float16 h[4] = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(a.out) + orow + oc) = *reinterpret_cast<uint2*>(h);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (oc + r < cout_out) store_as(a.out, orow + oc + r, a.out_dtype, v[r]);
      }
    }
  }
}

// defined in igemm_pipe.hip; returns DC_ERR_UNSUPPORTED-free status (always handles tile_n == 128)
int dc_igemm_launch_pipe(const IgemmArgs& a, int dtype, hipStream_t s);
int dc_igemm_pipe_shape(const IgemmArgs& a);   // 0: 128x128, 1: 256x128, 2: 256x256 tile
// igemm_wide.hip: the 256 x 256 tile on the 8-phase main loop
int dc_igemm_launch_wide8(const IgemmArgs& a, int dtype, hipStream_t s);
// hipcc expands the integer divisions of tile_of_block on the VECTOR unit, so tile_m / tile_n (and everything derived from them: weight
// panel offsets, sample indices, descriptor bases) live in VGPRs although they are wave-uniform — and every `buffer_load ... lds` that
// takes such a value as its SCALAR offset is wrapped in a waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec / branch: ~10
// instructions and a serialising round trip per LDS-DMA piece: 40 of the 70 LDS-DMA instructions of conv3_halo's one-image-per-patch
// form were; cdna_hip_programming.md T20).  One readfirstlane each makes the whole chain scalar.  Only for the kernels that feed
// buffer descriptors: the per-lane-address forms sit at the register limit and spill (48-269 VGPRs) when their scalar count changes.
__device__ __forceinline__ void tile_of_block_scalar(const IgemmArgs& a, int& tile_m, int& tile_n) {
  tile_of_block(a, tile_m, tile_n);
  tile_m = __builtin_amdgcn_readfirstlane(tile_m);
  tile_n = __builtin_amdgcn_readfirstlane(tile_n);
  asm volatile("" : "+s"(tile_m), "+s"(tile_n));      // pin them in SGPRs (a bare readfirstlane of a uniform value may be folded away)
}

// conv3_halo.hip
bool dc_conv3_halo_applicable(const IgemmArgs& a, int dtype);
int dc_conv3_halo_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s);
bool dc_conv3_up4_applicable(const IgemmArgs& a, int dtype);   // upsample + 3x3 conv as four 2x2-tap phases
int dc_conv3_up4_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s);
int dc_igemm_launch_pipe_up4(const IgemmArgs& a, int dtype, hipStream_t s);   // the same on the tap-gather kernel (sources < 8x8)
bool dc_conv3_halo_pn_ok(const IgemmArgs& a, int dtype, bool up4);   // producer-side GroupNorm possible (epi_pn.h)
unsigned dc_conv3_halo_pn_timeouts();                          // reads and clears the device-side failure counter (synchronous)
// conv3_ws.hip: wave-specialised halo conv (loader / transform waves + MFMA waves) with the input's GroupNorm(+SiLU) fused in
bool dc_conv3_ws_ok(const IgemmArgs& a, int dtype);
int dc_conv3_ws_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s);
// igemm_xreg.hip: activation-stationary GEMM for K <= 256 (16-bit, 1 tap)
bool dc_igemm_xreg_applicable(const IgemmArgs& a, int dtype);
int dc_igemm_xreg_launch(const IgemmArgs& a, int dtype, hipStream_t s);
// conv3_halo.hip: thin-output 3x3 conv (Cout <= 16, bias only)
bool dc_conv3_thin_applicable(const IgemmArgs& a, int dtype);
int dc_conv3_thin_launch(const IgemmArgs& a, int dtype, int n_img, hipStream_t s);
