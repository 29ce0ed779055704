// epi_pn.h — producer-side GroupNorm: the epilogue of a 3x3 halo conv that stores its output ALREADY normalised
// (y = act(gn(v)), and optionally the raw v beside it) for the GroupNorm -> SiLU -> Conv2d / GroupNorm -> proj_in site that
// consumes it (diffusers ResnetBlock2D.norm2 / norm1 of the next block, Transformer2DModel.norm; behind reference
// nets/unet.py:186-195).
//
// Why: the GroupNorm between two convs is a pure read + write pass over the tensor (13-16 % of a UNet scoring step), and folding it
// into the CONSUMER's loaders (conv3_ws.hip) costs that conv its second workgroup per CU (0.75 instead of 1.0-1.06 PF).  Here the
// PRODUCER normalises: its accumulators already hold every value in fp32, the statistics it needs are the quad records it writes
// anyway (igemm_epilogue.h, IgemmArgs::qstats), and the consumer stays the plain two-workgroup halo conv.  What a producer tile
// lacks is the rest of its sample: a 256-pixel tile of a 32x32 image sees a quarter of every group.  So the workgroups of one
// (sample, N tile) exchange their records through memory:
//
//   1. every wave writes its (mean, M2) quad records (one 128-pixel part x 16 quads = one 128-byte line) with ONE write-through
//      store instruction, waits for it, the workgroup meets at a barrier and ONE lane bumps the arrival counter of the (sample, N tile);
//   2. ONE wave polls that counter (relaxed agent-scope loads, s_sleep between polls) until all `tiles` workgroups of the sample
//      arrived, and the workgroup meets again;
//   3. every wave copies the sample's records of ITS 64 channels into LDS (16-byte `sc1` loads, two records per lane and load, all
//      loads in flight together) and the 16 lanes of a row fold the groups of their two 8-channel runs together (pn_fold_row: the GroupNorm
//      kernels' shifted one-pass form over the same records, the parts dealt to the lanes);
//   4. y = act(v * a + b) in place and the normalised store.  The raw tensor, where something reads it (a ResNet's shortcut, the transformer's
//      residual), is stored between the arrival and the poll: its stores ride inside the counter's round trip.
// A sample that IS one tile (16x16) exchanges nothing through memory: the two parts meet in LDS.  The four-phase upsample conv takes the
// same path with the four phases of every low-resolution tile as the group.
//
// Forward progress: a workgroup waits only for workgroups of its own (sample, N tile), which the launch places on consecutive
// positions of one XCD's dispatch queue (conv3_halo.hip, PN block order).  Workgroups are dispatched in index order, so when one member
// of a group is resident every earlier group of that queue is resident or finished, and the members not yet dispatched are next in
// line: they start as soon as ANY resident workgroup of an earlier, complete group retires — which those can always do.  (If the
// hardware dealt blocks to XCDs in another way the members would merely sit further apart in the one global order; same argument.)
// The host refuses groups of more than 16 workgroups — a quarter of ONE XCD's 64 workgroup slots (dc_igemm_pn_ok).  The poll loop is bounded all the same (PN_TIMEOUT_TICKS of the
// 100 MHz real-time counter): a wave that times out raises g_pn_timeouts (checked by the host, dc_pn_timeouts) and goes on with
// whatever records are there — wrong numbers and an error, never a hang.
#pragma once
#include "igemm_epilogue.h"

#ifndef PN_TIMEOUT_TICKS
#define PN_TIMEOUT_TICKS 3000000ull          // 30 ms at 100 MHz; a real wait is the dispatch skew inside one group (microseconds)
#endif

struct PnCtx {
  int sample;            // output sample of the workgroup (one image per patch)
  int part;              // this wave's 128-pixel part of the sample
  int parts;             // parts per sample (HW / 128)
  int tiles;             // workgroups per (sample, N tile) = HW / 256
  int qpg;               // quads per GroupNorm group (channels per group / 4): 1, 2, 4 or 8
  unsigned* cnt;         // arrival counter of this (sample, N tile): monotonic, zeroed ONCE by the caller, never reset
  unsigned* timeouts;    // library-wide failure counter
  const float* gam;      // LDS: gamma / beta of the lane's run 0 (run k: 32 k floats further)
  const float* bet;
  float2* scr;           // LDS scratch, parts x 16 records of the wave's 64 channels, written only after the workgroup's barrier below: per
                         // WAVE when the sample spans several tiles, shared by the two waves of an N half when it is one tile
  int* flag;             // LDS word: the polling wave tells the others whether the wait completed
  float eps;
  int silu;
};

// Fold the records of one GroupNorm group — local quads [q0, q0 + qpg) of every part, in scr[part * 16 + quad] — with the 16 lanes of a
// row sharing the parts (lane lr takes parts lr, lr + 16, ...): equal counts, means shifted by the first record's (gn_fold_rec's form:
// mean = piv + S1 / R, M2 = S2 + nq (S3 - S1^2 / R)), lane partials joined by the DPP butterfly, so every lane of the row ends with the
// same bits.  A fixed order that depends on (parts, qpg) only.  A lane folding all parts itself (gn_fold_rec) waits one LDS latency per
// record: 10 k cycles per tile at 32 parts, against a few hundred here.
__device__ __forceinline__ void pn_fold_row(const float2* scr, int parts, int q0, int qpg, int lr, float nq, float& mean, float& var) {
#pragma clang fp contract(off)
  auto row_sum = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
  };
  const float piv = scr[q0].x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int p = lr; p < parts; p += 16)
    for (int q = q0; q < q0 + qpg; ++q) {
      const float2 v = scr[p * 16 + q];
      const float d = v.x - piv;
      s1 += d; s3 = __builtin_fmaf(d, d, s3); s2 += v.y;
    }
  s1 = row_sum(s1); s2 = row_sum(s2); s3 = row_sum(s3);
  const float R = (float)(parts * qpg), iR = 1.0f / R;
  mean = __builtin_fmaf(s1, iR, piv);
  const float m2 = __builtin_fmaf(nq, fmaxf(__builtin_fmaf(-s1 * s1, iR, s3), 0.f), s2);
  var = fmaxf(m2 / (R * nq), 0.f);
}

// acc[i][j]: cout fragment i, pixel fragment j of the wave (128 pixels x 64 couts); brv: bias (+ row vector) of the lane's run 0 in LDS
template <typename T, typename RowFn>
__device__ __forceinline__ void epi_halo_pn(const IgemmArgs& a, f32x4 (&acc)[4][8], int tile_n, int wn, int lq, RowFn rowfn,
                                            const float* brv, const PnCtx& c) {
  constexpr int TM = 8, NK = 2;
  constexpr bool res16 = sizeof(T) == 2;
  const int abl = DC_HALO_ABL();                             // 0 outside diagnostic builds (timing-only ablations: 16 no wait, 32 no SiLU)
  const int c0 = tile_n * 128 + wn * 64 + lq * 8;            // run k starts at c0 + 32 k (Cout is a multiple of 128: every run is real)
  // ---- A1: bias + row vector (staged in LDS at kernel start) ----
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(brv + 32 * k), hi = *reinterpret_cast<const f32x4*>(brv + 32 * k + 4);
#pragma unroll
    for (int j = 0; j < TM; ++j) { acc[2 * k][j] += lo; acc[2 * k + 1][j] += hi; }
  }
  // ---- A2: residual, every load of the wave in front of every store (vmcnt is one in-order counter) ----
  int orow[TM];
  {
    EpiRow row[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) { rowfn(j, row[j]); orow[j] = row[j].o; }
    if (a.residual) {
      if (res16) {
        chunk16 rc[TM][NK];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int k = 0; k < NK; ++k)
            rc[j][k] = *reinterpret_cast<const chunk16*>(reinterpret_cast<const char*>(a.residual) + ((size_t)row[j].r * a.res_ld + c0 + 32 * k) * 2);
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            float rf[8];
            chunk_to_f<T>(rc[j][k], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[2 * k + (e >> 2)][j][e & 3] += rf[e];
          }
      } else {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            float rf[8];
            ld8(reinterpret_cast<const float*>(a.residual) + (size_t)row[j].r * a.res_ld + c0 + 32 * k, rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[2 * k + (e >> 2)][j][e & 3] += rf[e];
          }
      }
    }
  }
  DC_STAMP(3);
  f32x4 rec;
  // ---- quad statistics of the wave's part: the arithmetic of igemm_epilogue.h's `emit` (whole-wave form), so that the records
  // are the ones a plain conv3_halo launch writes into IgemmArgs::qstats ----
  {
#pragma clang fp contract(off)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    auto share0 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xF, 0xF, true)); };
    auto row_sum = [](float v) {
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
      return v;
    };
    float piv[NK][2];
    f32x2 sm[NK][2], sq[NK][2];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        piv[k][qd] = share0(acc[2 * k + qd][0][0]);
        sm[k][qd] = f32x2{0.f, 0.f}; sq[k][qd] = f32x2{0.f, 0.f};
      }
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 d = f32x2{acc[2 * k + qd][j][e], acc[2 * k + qd][j][e + 1]} - f32x2{piv[k][qd], piv[k][qd]};
            sm[k][qd] += d;
            sq[k][qd] = __builtin_elementwise_fma(d, d, sq[k][qd]);
          }
    const float inv_n = 1.0f / (float)(TM * 16 * 4);
    float r[NK][4];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const float S = row_sum(sm[k][qd][0] + sm[k][qd][1]), Q = row_sum(sq[k][qd][0] + sq[k][qd][1]);
        r[k][2 * qd] = __builtin_fmaf(S, inv_n, piv[k][qd]);
        r[k][2 * qd + 1] = fmaxf(__builtin_fmaf(-S * S, inv_n, Q), 0.f);
      }
    // every lane of a 16-lane row holds the row's records (the DPP butterflies leave the totals everywhere): lane lr = 0 publishes run
    // 0's pair of quads, lane lr = 1 run 1's — ONE 16-byte store instruction of the wave writes its whole 128-byte record line
    const int lr = threadIdx.x & 15;
    rec = lr == 0 ? f32x4{r[0][0], r[0][1], r[0][2], r[0][3]} : f32x4{r[1][0], r[1][1], r[1][2], r[1][3]};
  }
  const int lrp = threadIdx.x & 15;
  const int qpub = 2 * lq + 8 * (lrp & 1);                   // first quad (among the wave's 16) of the pair this lane publishes
  float* const grec = a.qstats + (((size_t)c.sample * c.parts + c.part) * (a.Cout >> 2) + ((tile_n * 128 + wn * 64) >> 2) + qpub) * 2;
  // the raw output (when something reads it): stored while the arrival counter's round trip is in flight
  auto store_raw = [&]() {
    if (a.out) {
      if (a.out_dtype == DC_F32) {
  #pragma unroll
        for (int j = 0; j < TM; ++j)
  #pragma unroll
          for (int k = 0; k < NK; ++k) {
            float* op = reinterpret_cast<float*>(a.out) + (size_t)orow[j] * a.out_ld + c0 + 32 * k;
            *reinterpret_cast<f32x4*>(op) = acc[2 * k][j]; *reinterpret_cast<f32x4*>(op + 4) = acc[2 * k + 1][j];
          }
      } else {
  #pragma unroll
        for (int j = 0; j < TM; ++j)
  #pragma unroll
          for (int k = 0; k < NK; ++k) {
            float v[8];
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * k + (e >> 2)][j][e & 3];
            *reinterpret_cast<chunk16*>(reinterpret_cast<T*>(a.out) + (size_t)orow[j] * a.out_ld + c0 + 32 * k) = f_to_chunk<T>(v);
          }
      }
    }
  };
  bool ok = true;
  if (c.tiles == 1) {
    // ---- the whole sample is this workgroup's tile (16x16 images): the two parts meet in LDS, nothing crosses a workgroup.  The records
    // still go to memory (plain stores) for any OTHER GroupNorm that reads this tensor's statistics later.
    if (lrp < 2) *reinterpret_cast<f32x4*>(grec) = rec;
    __syncthreads();                                         // every wave has left the tap loop: the scratch (a halo buffer) is free
    DC_STAMP(4);
    if (lrp < 2) *reinterpret_cast<f32x4*>(c.scr + c.part * 16 + qpub) = rec;
    __syncthreads();
    DC_STAMP(5);
    store_raw();
  } else {
    // ---- 1./2. publish, arrive, wait for the rest of the sample.  The hand-off is the counter form measured in MI355X_MICROARCH.md
    // (inter-workgroup visibility, third row of the table of `sc1` hand-offs): every record line is written whole by ONE `sc1`
    // (write-through) 16-byte store instruction of one wave; every storing wave drains its stores (vmcnt 0); the workgroup meets; ONE
    // lane adds to the agent-scope counter; ONE wave polls it with `sc1` loads; the workgroup meets again; every record is then
    // loaded with a 16-byte `sc1` load.  No L2 write-back, no cache invalidate: a fence pair cost this epilogue 30 % of the conv.
    // The counter is never reset: every launch adds exactly `tiles` arrivals, the ticket my add returns tells which launch this is, and
    // the target is the next multiple of `tiles` above it — an old value of the counter line can only be too small.
    if (lrp < 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(grec), "v"(rec) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // every wave's records are out; every wave has left the tap loop (LDS is free)
    DC_STAMP(4);
    unsigned ticket = 0;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(c.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    store_raw();               // 64 KiB of stores per workgroup behind the add: vmcnt returns in order, so the ticket does not wait for them
    if (threadIdx.x < 64) {
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      const unsigned target = (ticket & ~(unsigned)(c.tiles - 1)) + (unsigned)c.tiles;        // tiles is a power of two
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      ok = false;
      for (;;) {
        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if ((int)(seen - target) >= 0 || (abl & 16)) { ok = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > PN_TIMEOUT_TICKS) break;
        __builtin_amdgcn_s_sleep(4);
      }
      if (threadIdx.x == 0) {
        *c.flag = ok ? 1 : 0;
        if (!ok) __hip_atomic_fetch_add(c.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    ok = *c.flag != 0;
    DC_STAMP(5);
    // ---- 3. the sample's records of this wave's 16 quads -> LDS: two records per lane and load, at most four loads per lane (32 parts),
    // all in flight together
    {
      const int lane = threadIdx.x & 63;
      const int CQ = a.Cout >> 2;
      const float* recs = a.qstats + ((size_t)c.sample * c.parts * CQ + ((tile_n * 128 + wn * 64) >> 2)) * 2;
      const int npair = c.parts * 8;
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = lane + 64 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < npair) {
          const float* p = recs + ((size_t)(i >> 3) * CQ + 2 * (i & 7)) * 2;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = lane + 64 * u;
        if (i < npair) *reinterpret_cast<f32x4*>(c.scr + 2 * i) = v[u];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0): the wave's own LDS writes (its lanes read each other's records)
  __builtin_amdgcn_wave_barrier();
  float ga[NK][8], gb[NK][8];
  {
    const float nq = 512.0f;                                 // values per record: 128 pixels x 4 channels
    const int lr = threadIdx.x & 15;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int ql = 2 * lq + 8 * k;                         // the run's first quad among the wave's 16
      float mean[2], rstd[2];
      if (c.qpg == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float var;
          pn_fold_row(c.scr, c.parts, ql + h, 1, lr, nq, mean[h], var);
          rstd[h] = rsqrtf(var + c.eps);
        }
      } else {
        float var;
        pn_fold_row(c.scr, c.parts, (ql / c.qpg) * c.qpg, c.qpg, lr, nq, mean[0], var);
        mean[1] = mean[0]; rstd[0] = rstd[1] = rsqrtf(var + c.eps);
      }
      if (!ok) rstd[0] = rstd[1] = __builtin_nanf("");     // a timed-out wait: poison what is stored, so the failure also shows downstream
      float gm[8], bt[8];
      ld8(c.gam + 32 * k, gm);
      ld8(c.bet + 32 * k, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s = rstd[e >> 2] * gm[e];
        ga[k][e] = s; gb[k][e] = bt[e] - mean[e >> 2] * s;
      }
    }
  }
  DC_STAMP(6);
  // ---- 4. the normalised tensor (the raw one, if it has a reader, went out while the workgroup waited) ----
  // (the activation switch sits OUTSIDE the unrolled loops: a taken scalar branch per element costs ~20 cycles on this core)
  auto store_norm = [&](auto actc) {
    constexpr bool ACT = decltype(actc)::value;
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = acc[2 * k + (e >> 2)][j][e & 3] * ga[k][e] + gb[k][e];
          if (ACT) x = silu_t<T>(x);
          v[e] = x;
        }
        T* op = reinterpret_cast<T*>(a.pn_out) + (size_t)orow[j] * a.pn_ld + c0 + 32 * k;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(op) + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          *reinterpret_cast<chunk16*>(op) = f_to_chunk<T>(v);
        }
      }
  };
  if (c.silu && !(abl & 32)) store_norm(EpiIC<1>{});
  else store_norm(EpiIC<0>{});
}
