// pack.hip — weight packing for dc_igemm on the device.  The packed layouts are part of the C-ABI (include/dcamd.h): a host in any
// language hands over the framework's fp32 parameter tensors (diffusers layouts: Conv2d [Cout, Cin, 3, 3], Linear [Cout, K]) and
// gets the buffers dc_igemm consumes.  Run once per (weights, dtype); element-wise gathers, nothing to tune.
#include "common.h"

static __device__ __forceinline__ void put(void* out, size_t i, int dtype, float v) { store_as(out, i, dtype, v); }

// out[r, k] (r < cout_pad, k < kpad) = r < cout && k < K ? w[perm ? perm[r] : r][k] * (col_scale ? col_scale[k] : 1) : 0
__global__ __launch_bounds__(256) void pack_matrix_kernel(const float* __restrict__ w, int cout, int K, long long ldw, const int32_t* __restrict__ perm,
                                                          const float* __restrict__ col_scale, void* __restrict__ out, int cout_pad, int kpad, int dtype) {
  const long long total = (long long)cout_pad * kpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / kpad), k = (int)(i - (long long)r * kpad);
    float v = 0.f;
    if (r < cout && k < K) {
      v = w[(long long)(perm ? perm[r] : r) * ldw + k];
      if (col_scale) v *= col_scale[k];
    }
    put(out, (size_t)i, dtype, v);
  }
}

// Conv2d weight [cout, cin, 3, 3], input channels [c_lo, c_hi) -> out[r, tap * C + c], C = c_hi - c_lo, tap = ky*3 + kx; columns >= 9C zero
__global__ __launch_bounds__(256) void pack_conv3_kernel(const float* __restrict__ w, int cout, int cin, int c_lo, int C, void* __restrict__ out,
                                                         int cout_pad, int kpad, int dtype) {
  const long long total = (long long)cout_pad * kpad;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / kpad), k = (int)(i - (long long)r * kpad);
    float v = 0.f;
    if (r < cout && k < 9 * C) {
      const int tap = k / C, c = k - tap * C;
      v = w[((long long)r * cin + c_lo + c) * 9 + tap];
    }
    put(out, (size_t)i, dtype, v);
  }
}

// four-phase form of "nearest-2x upsample, then 3x3 conv" (dc_igemm_params.up4): out[phase = 2a+b][r][(dy*2+dx)*cin + c] = sum of the
// 3x3 taps (ky, kx) that read source pixel (y+a-1+dy, x+b-1+dx): rows a=0 -> {0} | {1,2}, a=1 -> {0,1} | {2}; columns alike.
// Summed in fp32, ky outer / kx inner, before the rounding to dtype.
__global__ __launch_bounds__(256) void pack_up4_kernel(const float* __restrict__ w, int cout, int cin, void* __restrict__ out, int cout_pad, int dtype) {
  const int K4 = 4 * cin;
  const long long total = 4LL * cout_pad * K4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i % K4); long long q = i / K4;
    const int r = (int)(q % cout_pad), ph = (int)(q / cout_pad);
    float v = 0.f;
    if (r < cout) {
      const int a = ph >> 1, b = ph & 1, t = k / cin, c = k - t * cin, dy = t >> 1, dx = t & 1;
      const int ky0 = a == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky1 = a == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
      const int kx0 = b == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx1 = b == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
      const float* wp = w + ((long long)r * cin + c) * 9;
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) v = v + wp[ky * 3 + kx];
    }
    put(out, (size_t)i, dtype, v);
  }
}

// y[r] = (b ? b[r] : 0) + sum_k w[r, k] * v[k], fp32, k ascending (LayerNorm beta folded into a bias: W beta + c)
__global__ __launch_bounds__(256) void fold_bias_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ v,
                                                        const int32_t* __restrict__ perm, int cout, int K, float* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= cout) return;
  const int src = perm ? perm[r] : r;
  float s = 0.f;
  if (v) for (int k = 0; k < K; ++k) s += w[(long long)src * K + k] * v[k];
  out[r] = s + (b ? b[src] : 0.f);
}

// GEGLU row order: 16-row blocks alternate value / gate halves — packed row 32*blk + i = value row 16*blk + i, 32*blk + 16 + i = gate row
__global__ void geglu_perm_kernel(int n_half, int32_t* __restrict__ perm) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= 2 * n_half) return;
  const int blk = r >> 5, i = r & 31;
  perm[r] = i < 16 ? 16 * blk + i : n_half + 16 * blk + (i - 16);
}

static unsigned pk_grid(long long total) { const long long b = (total + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }
static bool pk_dtype_ok(int dt) { return dt == DC_F32 || dt == DC_BF16 || dt == DC_F16; }

extern "C" int64_t dc_packed_bytes(int32_t cout, int32_t K, int32_t dtype, int32_t tile_n) {
  return (int64_t)dc_igemm_cout_pad(cout, tile_n) * K * dc_dtype_size(dtype);
}

extern "C" int dc_pack_weights_matrix(const float* w, int32_t cout, int32_t K, int32_t kpad, const int32_t* row_perm, const float* col_scale,
                                      void* out, int32_t dtype, int32_t tile_n, dc_stream s) {
  DC_REQUIRE(w && out && cout > 0 && K > 0 && kpad >= K, DC_ERR_ARG, "dc_pack_weights_matrix: bad args");
  DC_REQUIRE(pk_dtype_ok(dtype), DC_ERR_DTYPE, "dc_pack_weights_matrix: dtype %d", dtype);
  const int cp = dc_igemm_cout_pad(cout, tile_n);
  hipLaunchKernelGGL(pack_matrix_kernel, dim3(pk_grid((long long)cp * kpad)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), w, cout, K,
                     (long long)K, row_perm, col_scale, out, cp, kpad, dtype);
  return dc_check_launch("dc_pack_weights_matrix");
}

extern "C" int dc_pack_weights_conv3x3(const float* w, int32_t cout, int32_t cin, int32_t c_lo, int32_t c_hi, int32_t kpad, void* out,
                                       int32_t dtype, int32_t tile_n, dc_stream s) {
  DC_REQUIRE(w && out && cout > 0 && cin > 0 && c_lo >= 0 && c_hi > c_lo && c_hi <= cin && kpad >= 9 * (c_hi - c_lo), DC_ERR_ARG,
             "dc_pack_weights_conv3x3: bad args (cout=%d cin=%d slice [%d,%d) kpad=%d)", cout, cin, c_lo, c_hi, kpad);
  DC_REQUIRE(pk_dtype_ok(dtype), DC_ERR_DTYPE, "dc_pack_weights_conv3x3: dtype %d", dtype);
  const int cp = dc_igemm_cout_pad(cout, tile_n);
  hipLaunchKernelGGL(pack_conv3_kernel, dim3(pk_grid((long long)cp * kpad)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), w, cout, cin, c_lo,
                     c_hi - c_lo, out, cp, kpad, dtype);
  return dc_check_launch("dc_pack_weights_conv3x3");
}

extern "C" int dc_pack_weights_up4(const float* w, int32_t cout, int32_t cin, void* out, int32_t dtype, int32_t tile_n, dc_stream s) {
  DC_REQUIRE(w && out && cout > 0 && cin > 0, DC_ERR_ARG, "dc_pack_weights_up4: bad args");
  DC_REQUIRE(pk_dtype_ok(dtype), DC_ERR_DTYPE, "dc_pack_weights_up4: dtype %d", dtype);
  const int cp = dc_igemm_cout_pad(cout, tile_n);
  hipLaunchKernelGGL(pack_up4_kernel, dim3(pk_grid(16LL * cp * cin)), dim3(256), 0, reinterpret_cast<hipStream_t>(s), w, cout, cin, out, cp, dtype);
  return dc_check_launch("dc_pack_weights_up4");
}

extern "C" int dc_pack_weights_geglu(const float* w, const float* bias, int32_t n_half, int32_t K, const float* ln_gamma, const float* ln_beta,
                                     void* out_w, float* out_bias, int32_t* perm_ws, int32_t dtype, dc_stream s) {
  DC_REQUIRE(w && out_w && perm_ws && n_half > 0 && n_half % 16 == 0 && K > 0, DC_ERR_ARG, "dc_pack_weights_geglu: bad args (n_half %d must be a multiple of 16)", n_half);
  DC_REQUIRE(pk_dtype_ok(dtype), DC_ERR_DTYPE, "dc_pack_weights_geglu: dtype %d", dtype);
  DC_REQUIRE((out_bias != nullptr) == (bias != nullptr || ln_beta != nullptr), DC_ERR_ARG, "dc_pack_weights_geglu: out_bias must be given exactly when there is a bias or a folded beta");
  hipStream_t st = reinterpret_cast<hipStream_t>(s);
  const int cout = 2 * n_half;
  hipLaunchKernelGGL(geglu_perm_kernel, dim3((cout + 255) / 256), dim3(256), 0, st, n_half, perm_ws);
  const int cp = dc_igemm_cout_pad(cout, 128);
  hipLaunchKernelGGL(pack_matrix_kernel, dim3(pk_grid((long long)cp * K)), dim3(256), 0, st, w, cout, K, (long long)K, perm_ws, ln_gamma, out_w, cp, K, dtype);
  if (out_bias) hipLaunchKernelGGL(fold_bias_kernel, dim3((cout + 255) / 256), dim3(256), 0, st, w, bias, ln_beta, perm_ws, cout, K, out_bias);
  return dc_check_launch("dc_pack_weights_geglu");
}

extern "C" int dc_fold_layernorm_bias(const float* w, const float* bias, const float* ln_beta, int32_t cout, int32_t K, float* out_bias, dc_stream s) {
  DC_REQUIRE(w && ln_beta && out_bias && cout > 0 && K > 0, DC_ERR_ARG, "dc_fold_layernorm_bias: bad args");
  hipLaunchKernelGGL(fold_bias_kernel, dim3((cout + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(s), w, bias, ln_beta, nullptr, cout, K, out_bias);
  return dc_check_launch("dc_fold_layernorm_bias");
}

// workspace requirements (bytes) of the ops: only GroupNorm needs one; the GEMM / attention / LayerNorm / element-wise kernels keep
// everything in registers and LDS.  Declared per op so a host sizes its arena without knowing that.
extern "C" int64_t dc_workspace_bytes_groupnorm(const dc_groupnorm_params* p) {
  if (!p) return 0;
  const int splits = p->splits > 0 ? p->splits : dc_groupnorm_splits(p->n, p->HW, p->C + p->C1);
  return 4 * dc_groupnorm_ws_floats(p->n, p->groups, splits);
}
extern "C" int64_t dc_workspace_bytes_igemm(const dc_igemm_params*) { return 0; }
extern "C" int64_t dc_workspace_bytes_attention(const dc_attention_params*) { return 0; }
extern "C" int64_t dc_workspace_bytes_layernorm(const dc_layernorm_params*) { return 0; }
