// igemm_pipe.hip — the throughput implicit-GEMM kernel: 256 pixels x 128 couts per 512-thread
// workgroup (8 waves as 4(M) x 2(N), 64x64 per wave), K-step = 128 B per row, operands brought
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging) into a 3-stage ring with a
// prefetch distance of two K-steps, ONE raw s_barrier per K-step and counted vmcnt waits (never 0
// inside the loop), so two K-steps of loads are always in flight behind the MFMAs.
//
// LDS-DMA writes base + lane*16, i.e. the LDS image is lane-linear; the XOR swizzle of lds_off()
// is therefore applied on the per-lane SOURCE address (which 16-B chunk of the row a lane fetches)
// and again on the fragment reads.  Out-of-image taps (3x3 padding) and rows past M fetch from a
// 256-B zero page, so the gather needs no branches and no LDS zero-fill.
//
// Pipeline (S = 3 stages; group g = the loads of K-step g, 6 LDS-DMA instructions per lane):
//   prologue: issue g0, g1
//   iteration ks:  s_waitcnt vmcnt(6|0)   own loads of g[ks] have landed (g[ks+1] may be in flight)
//                  s_barrier              everybody's have, and everybody finished compute(ks-1)
//                  issue g[ks+2]          into the stage compute(ks-1) has just released
//                  compute(ks)            16 ds_read_b128 + 32 MFMA 16x16x32 per wave
#include "igemm_common.h"

__device__ chunk16 g_zero_page[16];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
__global__ __launch_bounds__(512, 2) void igemm_pipe_kernel(const IgemmArgs a) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int BM = 256, BN = 128, S = 3;
  constexpr int XST = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int TM = 4, TN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;
  int tile_m, tile_n;
  tile_of_block(a, tile_m, tile_n);

  const int HWo = a.Hout * a.Wout;
  const int pad = (a.taps == 9) ? 1 : 0;
  const int Hs = a.upsample ? (a.Hin >> 1) : a.Hin;
  const int Ws = a.upsample ? (a.Win >> 1) : a.Win;
  const int lrow = t >> 3;                                   // 0..63; loader rows are lrow + 64*i
  const int lchunk = (t & 7) ^ ((t >> 4) & 7);               // logical chunk this lane fetches (same for every i)

  // per loader row: 64-bit sample base of each source (map lookups happen HERE, never in the loop),
  // top-left tap coordinates; in-sample offsets stay 32-bit.
  const T* base0[4]; const T* base1[4]; int iy0[4], ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * BM + lrow + 64 * i;
    const bool vm = m < a.M;
    const int mm = vm ? m : 0;
    const int n = mm / HWo;
    const int rem = mm - n * HWo;
    const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
    const int n0 = a.map0 ? a.map0[n] : n;
    const int n1 = a.src1 ? (a.map1 ? a.map1[n] : n) : 0;
    base0[i] = reinterpret_cast<const T*>(a.src0) + (size_t)n0 * Hs * Ws * a.ld0;
    base1[i] = reinterpret_cast<const T*>(a.src1) + (size_t)n1 * Hs * Ws * a.ld1;
    iy0[i] = vm ? oy * a.stride - pad : -(1 << 20);
    ix0[i] = ox * a.stride - pad;
    // consume the map loads now: an ordinary load still pending inside the loop would make hipcc
    // drain the LDS-DMA queue (vmcnt(0)) at its first use there
    asm volatile("" ::"v"(base0[i]), "v"(base1[i]));
  }
  const T* wbase = reinterpret_cast<const T*>(a.W) + (size_t)(tile_n * BN + lrow) * a.Ktot + lchunk * EPC;
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (t & 7) * 16;
  const int Hm1 = a.Hin - 1, Wm1 = a.Win - 1;

  int itap = 0, icc = 0;   // (tap, channel chunk) of the next K-step to issue
  auto issue = [&](int ks) {
    const int st = ks % S;
    int ky = 0, kx = 0;
    if (a.taps == 9) { ky = itap / 3; kx = itap - ky * 3; }
    const bool s1 = icc >= a.c0chunks;
    const int ld = s1 ? a.ld1 : a.ld0;
    const int coff = (s1 ? icc - a.c0chunks : icc) * BKE + lchunk * EPC;
    char* xs = smem + st * STAGE + wave * 1024;              // wave-uniform LDS base; HW adds lane*16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = (unsigned)iy <= (unsigned)Hm1 && (unsigned)ix <= (unsigned)Wm1;
      const int cy = min(max(iy, 0), Hm1), cx = min(max(ix, 0), Wm1);     // always a real pixel: no branch
      const int sy = a.upsample ? (cy >> 1) : cy, sx = a.upsample ? (cx >> 1) : cx;
      const int off = (sy * Ws + sx) * ld + coff;
      const char* gp = reinterpret_cast<const char*>((s1 ? base1[i] : base0[i]) + off);
      gp = ok ? gp : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(xs + i * 8192), 16, 0, 0);
    }
    char* ws = smem + st * STAGE + XST + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* gp = reinterpret_cast<const char*>(wbase + (size_t)(64 * i) * a.Ktot + (size_t)ks * BKE);
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(ws + i * 8192), 16, 0, 0);
    }
    if (++icc == a.cpt) { icc = 0; ++itap; }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (a.nk > 1) issue(1);
  for (int ks = 0; ks < a.nk; ++ks) {
    if (ks + 1 < a.nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks + 2 < a.nk) issue(ks + 2);
    const char* Xs = smem + (ks % S) * STAGE;
    const char* Wsm = Xs + XST;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int c = sub * 4 + lq;
      chunk16 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const chunk16*>(Xs + lds_off(wm * 64 + j * 16 + lr, c));
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const chunk16*>(Wsm + lds_off(wn * 64 + i * 16 + lr, c));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Mma<T>::run(wf[i], xf[j], acc[i][j]);
    }
  }
  // ---- epilogue, staged through LDS so that HBM sees whole 16-B chunks of whole rows ----
  // phase 1 (registers -> LDS, fp32): bias, per-sample row vector, activation, gate.  A lane holds 4
  // consecutive couts of one pixel per 16x16 tile; rows are padded by 16 B so the 8-lane write groups of
  // ds_write_b128 fall on distinct banks.  phase 2 (LDS -> HBM): 16 consecutive lanes cover one output
  // row; residual is read, and the result written, as one 16-byte access per 8 (16-bit) or 4 (f32) couts.
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the operand stages
  constexpr int OLD = BN + 4;                        // padded row, in floats
  float* otile = reinterpret_cast<float*>(smem);
  const bool geglu = a.act == DC_ACT_GEGLU;
  const int cout_out = geglu ? (a.Cout >> 1) : a.Cout;
  const int tcols = geglu ? BN / 2 : BN;             // output columns of this tile
  const int col0 = tile_n * tcols;
  {
    const int n0 = tile_n * BN + wn * 64;            // first packed cout of the wave
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int rloc = wm * 64 + j * 16 + lr;
      const int m = tile_m * BM + rloc;
      const int mm = m < a.M ? m : 0;
      const int n = mm / HWo;
      const float* rv = a.rowvec ? a.rowvec + (size_t)(a.rowvec_map ? a.rowvec_map[n] : n) * a.rowvec_ld : nullptr;
      const float* gt = a.gate ? a.gate + (size_t)(a.gate_map ? a.gate_map[n] : n) * a.gate_ld : nullptr;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int pc = n0 + i * 16 + lq * 4;
        float v[4];
        int lc;                                      // column inside the tile
        if (geglu) {
          if (i & 1) continue;
          lc = ((wn * 64 + i * 16) >> 1) + lq * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float val = acc[i][j][r], g = acc[(i + 1) % TN][j][r];
            if (a.bias) { val += a.bias[pc + r]; g += a.bias[pc + 16 + r]; }
            v[r] = val * gelu_erf_f(g);
          }
        } else {
          lc = wn * 64 + i * 16 + lq * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][r];
            const int c = pc + r;
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
        *reinterpret_cast<f32x4*>(otile + rloc * OLD + lc) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  }
  __syncthreads();
  {
    const int es_out = a.out_dtype == DC_F32 ? 4 : 8;          // couts per 16-byte output chunk
    const int cpr = tcols / es_out;                            // chunks per tile row
    const bool vec_ok = (cout_out % es_out == 0) && (a.out_ld % es_out == 0) &&
                        (!a.residual || ((a.res_ld % 8 == 0) && a.res_dtype != DC_F32) || ((a.res_ld % 4 == 0) && a.res_dtype == DC_F32));
    for (int idx = t; idx < BM * cpr; idx += 512) {
      const int rloc = idx / cpr, ch = idx - rloc * cpr;
      const int m = tile_m * BM + rloc;
      const int c = col0 + ch * es_out;
      if (m >= a.M || c >= cout_out) continue;
      const float* src = otile + rloc * OLD + ch * es_out;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
      f32x4 hi = {0.f, 0.f, 0.f, 0.f};
      if (es_out == 8) hi = *reinterpret_cast<const f32x4*>(src + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      size_t rrow = 0;
      if (a.residual) {
        const int n = m / HWo;
        rrow = (a.res_map ? (size_t)a.res_map[n] * HWo + (m - n * HWo) : (size_t)m) * a.res_ld + c;
      }
      const size_t o = (size_t)m * a.out_ld + c;
      if (vec_ok) {
        if (a.residual) {
          if (a.res_dtype == DC_F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              if (h * 4 < es_out) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.residual) + rrow + h * 4);
                v[h * 4] += r4[0]; v[h * 4 + 1] += r4[1]; v[h * 4 + 2] += r4[2]; v[h * 4 + 3] += r4[3];
              }
          } else {
            // 16-bit residual: es_out == 8 reads 16 B, es_out == 4 (f32 out) reads 8 B
            if (es_out == 8) {
              const chunk16 rc = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(a.residual) + rrow * 2);
              float rf[8];
              if (a.res_dtype == DC_BF16) chunk_to_f<__bf16>(rc, rf); else chunk_to_f<_Float16>(rc, rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rf[e];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += load_as(a.residual, rrow + e, a.res_dtype);
            }
          }
        }
        if (a.out_dtype == DC_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else if (a.out_dtype == DC_BF16) {
          *reinterpret_cast<chunk16*>(reinterpret_cast<__bf16*>(a.out) + o) = f_to_chunk<__bf16>(v);
        } else {
          *reinterpret_cast<chunk16*>(reinterpret_cast<_Float16*>(a.out) + o) = f_to_chunk<_Float16>(v);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < es_out && c + e < cout_out) {
            float x = v[e];
            if (a.residual) x += load_as(a.residual, rrow + e, a.res_dtype);
            store_as(a.out, o + e, a.out_dtype, x);
          }
      }
    }
  }
}

template <typename T>
static int launch_pipe(const IgemmArgs& a0, hipStream_t s) {
  constexpr int lds = 3 * (256 + 128) * 128;   // 144 KiB of the CU's 160 KiB
  static bool attr_done = false;
  auto kern = igemm_pipe_kernel<T>;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  IgemmArgs a = a0;
  a.tiles_m = (a.M + 255) / 256;
  const long long nblk = (long long)a.tiles_m * a.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) { dc_set_error("dc_igemm: bad grid %lld", nblk); return DC_ERR_SHAPE; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), lds, s, a);
  return dc_check_launch("dc_igemm(pipe)");
}

int dc_igemm_launch_pipe(const IgemmArgs& a, int dtype, hipStream_t s) {
  if (dtype == DC_BF16) return launch_pipe<__bf16>(a, s);
  if (dtype == DC_F16) return launch_pipe<_Float16>(a, s);
  return launch_pipe<float>(a, s);
}
