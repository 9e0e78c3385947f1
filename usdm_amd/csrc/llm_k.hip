// Mistral-7B decode/prefill kernels other than the shared GEMM / flash-attention / norm kernels.
//
// The decode step is HBM-bound (14.3 GB of bf16 weights per token at batch 1), so the design rule is:
// every weight byte is read exactly once, 16 B per lane, non-temporal, many loads in flight, and
// everything else (RMSNorm, residual add, SwiGLU, ban-mask + argmax) is fused into the GEMV that
// produces or consumes the vector.  Rounding points follow HF transformers' bf16 Mistral
// (third-party arithmetic of the reference): every Linear output, RMSNorm, RoPE
// product and residual add is rounded to bf16; logits are bf16 values compared in fp32.
#include "common.h"
#include <stdlib.h>
#include "../../include/usdm_hip.h"
#include "p2p.h"

#ifndef USDM_UNR1
#define USDM_UNR1 8   // ring depth of the one-row-per-wave variants (o_proj / down_proj)
#endif
#ifndef USDM_GEMV_X_FIRST
#define USDM_GEMV_X_FIRST 1   // 0: the input vector is requested behind the first weight ring and read twice by the RMSNorm prologue (A/B builds)
#endif
#ifndef USDM_GEMV_RES_PREFETCH
#define USDM_GEMV_RES_PREFETCH 1   // 0: the residual is read in the epilogue (A/B builds)
#endif
namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = w[i], b = x[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// GEMV: y = W x, W bf16 [N][ldw] row-major (the nn.Linear layout).
//   * workgroup = 4 waves; a wave owns RW output rows (GLU: RW gate + RW up rows) and streams them
//     together, lanes striding K in 16-B pieces, non-temporal loads;
//   * a ring of UNR K-iterations per row is always in flight (NR*UNR 16-B loads per lane), and the first
//     ring is issued BEFORE x is staged / RMS-normalised into LDS, so the prologue hides under HBM latency;
//   * RW is chosen by the launcher so that the grid is a whole number of workgroups per CU.
// ---------------------------------------------------------------------------------------------
#ifdef USDM_GEMV_TRACE
// debugging aid (tools/gemv_trace.py): per-workgroup phase timestamps (100 MHz wall clock)
__device__ unsigned long long g_gemv_trace[8192 * 8];
#define GTR(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemv_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define GTR(i) do { } while (0)
#endif

// MRG: the merged-attention input form (usdm_gemv_args.mrg_*) is a separate instantiation: its 16 partial loads per thread cost
// ~64 VGPRs, which lowered the occupancy of EVERY shape when the branch lived in the common kernel (measured: all GEMVs 20-30 %
// slower, profiles/r02_decode_ablation.txt).
// P2P: the fused peer-to-peer all-reduce epilogue (tensor-parallel decode) is likewise its own instantiation, so the single-GPU
// kernels carry none of it.
// CMB: the hand-off form of the merged-attention input (usdm_gemv_args.cmb_gran): one combine per head by the launch's first
// workgroups, granules to everyone, the wait under the first weight ring.  Its own instantiation for the same reason.
template <int RW, bool GLU, int NWV, bool MRG = false, bool P2P = false, bool CMB = false>
__global__ __launch_bounds__(NWV * 64) void gemv_kernel(const usdm_gemv_args a) {
  constexpr int NTH = NWV * 64;
  constexpr int NR = GLU ? 2 * RW : RW;   // rows streamed together by one wave
  constexpr int UNR = (NR >= 8) ? 2 : (NR >= 4) ? 4 : (NR == 3 ? 5 : (NR == 2 ? 8 : USDM_UNR1));  // ring depth: NR*UNR = 15..16 loads in flight per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;  // [Kpad] bf16, zero padded
  __shared__ float red[NWV];
  __shared__ float sv[NWV];
  __shared__ int si[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int skipv = a.skip ? *a.skip : 0;   // requested now, tested once the first ring is in flight (no exposed latency)
  unsigned p2p_epoch = 0u;      // likewise requested now, used only in the epilogue
  bool p2p_failed = false;
  if constexpr (P2P) { p2p_epoch = p2p_load_epoch(a.p2p); p2p_failed = p2p_load_err(a.p2p) != 0u; }
  GTR(0);
  const int K = a.K;
  const int Kpad = (K + 511) & ~511;
  const int nit = Kpad >> 9;

  // ---- rows of this wave
  const int rows_per_block = NWV * RW;                    // output features per workgroup
  const int ob = blockIdx.x * rows_per_block + wave * RW; // first output feature of this wave
  const u32x4* wp[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    int r;
    if (GLU) {  // packed layout: blocks of 32 rows = 16 gate + 16 up
      const int o = ob + (j % RW);
      r = (o >> 4) * 32 + (o & 15) + (j >= RW ? 16 : 0);
    } else {
      r = ob + j;
    }
    r = r < a.N ? r : a.N - 1;
    wp[j] = (const u32x4*)((const bf16_t*)a.W + (int64_t)r * a.ldw) + lane;
  }
  const bool tail_ok = ((nit - 1) << 9) + lane * 8 < K;  // is this lane inside K on the last iteration?
  // lm_head mode: a wave whose rows are ALL banned (the reference's bad_words_ids mask whole id ranges: 76 % of the vocabulary
  // in the text->unit round, inference.py:51-53) streams nothing; its logits are -inf either way
  bool active = true;
  if (a.part_val && a.ban) {
    // whole workgroup banned: publish "no candidate" and leave before anything is staged (no barrier has been reached yet)
    const int wb = blockIdx.x * rows_per_block;
    bool wg_active = false;
    for (int r = wb; r < wb + rows_per_block && r < a.N; ++r) wg_active |= (a.ban[r] == 0);
    if (!wg_active) {
      if (tid == 0) { a.part_val[blockIdx.x] = -INFINITY; a.part_idx[blockIdx.x] = 0x7fffffff; }
      if (a.y32 && tid < rows_per_block && wb + tid < a.N) a.y32[wb + tid] = -INFINITY;
      return;
    }
    active = false;
#pragma unroll
    for (int j = 0; j < NR; ++j)
      if (ob + j < a.N) active |= (a.ban[ob + j] == 0);
    active = __builtin_amdgcn_readfirstlane(active);
  }
  auto wload = [&](int j, int it) -> u32x4 {
    if (!active) return u32x4{0u, 0u, 0u, 0u};
    // last iteration may run past K: redirect to the row start (x is zero there in LDS, contributes 0)
    const u32x4* p = (it == nit - 1 && !tail_ok) ? wp[j] - lane : wp[j] + it * 64;
    return __builtin_nontemporal_load(p);
  };
  // The residual values of this wave's rows are requested HERE, at the start, not in the epilogue (round 4): there the
  // load was a full memory latency at the very end of every o_proj / down_proj launch, with nothing left to hide it.  Unconditional
  // (a predicated load is waited for at the join of its predicate): without a residual the address is the weight row, never used.
  unsigned short resraw[GLU ? 1 : NR];
  if constexpr (!GLU && USDM_GEMV_RES_PREFETCH) {
    const bf16_t* rbase = a.residual ? (const bf16_t*)a.residual : (const bf16_t*)a.W;
#pragma unroll
    for (int j = 0; j < NR; ++j) resraw[j] = rbase[min(ob + j, a.N - 1)];
  }
  // (round 4) This thread's first piece of the input vector and its RMSNorm weights are requested BEFORE the weight ring.  Loads
  // return in order: behind the ring they landed only when the whole ring had (~2 us), and the RMSNorm prologue - sum of squares,
  // barrier, a SECOND read of x and of the weights, normalise, barrier - started after that, with no weight load of this workgroup
  // in flight.  Now the prologue runs under the ring's latency and reads nothing twice.  Unconditional loads (clamped index; a
  // dummy address without RMSNorm): a predicated load would be waited for at the join of its predicate, i.e. before the ring is
  // issued.  Not in the merged-input variants (their x is not in memory).
  constexpr bool EARLY = USDM_GEMV_X_FIRST && !CMB && !MRG;
  const int i0 = tid * 8;
  // The three loads are HAND-COUNTED (asm, invisible to the compiler; cdna_hip_programming.md 5.7): the ring below is issued under
  // run-time tests (u < nit), so the wait-count pass cannot know how many loads are younger than x0 and would wait for all but the
  // few it is sure of - the whole ring again.  The wait further down names the exact count for the full-ring case.
  u32x4 x0 = {0u, 0u, 0u, 0u}, ge0v = x0, ge1v = x0;
  if constexpr (EARLY) {
    const int ic = min(i0, K - 8);
    // (without RMSNorm the two weight loads re-read the first 16 bytes of x - always there, K >= 8 - and are not used)
    const float* gp0 = a.norm_w ? a.norm_w + ic : (const float*)a.x;
    const float* gp1 = a.norm_w ? gp0 + 4 : gp0;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x0) : "v"((const bf16_t*)a.x + ic) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ge0v) : "v"(gp0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ge1v) : "v"(gp1) : "memory");
  }
  // The first ring is ALWAYS NR x UNR loads (slots past the row's K, or of a wave whose rows are all banned, re-read the row start
  // and are never multiplied / their results never used): the wait for the early loads below can then name an exact count.
  u32x4 ring[NR][UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const bool real = active && u < nit;
      const u32x4* p = real ? ((u == nit - 1 && !tail_ok) ? wp[j] - lane : wp[j] + u * 64) : wp[j] - lane;
      ring[j][u] = __builtin_nontemporal_load(p);
    }
  GTR(1);
  if constexpr (EARLY) {
    // in-order completion: "all but the NY youngest" = the three early loads have landed, the ring stays in flight.  ONE wait site on
    // every path (two sites in two branches made the compiler copy the registers of the loads at the branch - before the wait)
    constexpr int NY = UNR * NR;      // (the residual loads above are older: nothing else is issued between the early loads and here)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(x0), "+v"(ge0v), "+v"(ge1v) : "n"(NY));
  }
  if (skipv) return;   // the sequence ended in an earlier step of this host chunk (usdm_decode_state.done)
  const float4 ge0 = __builtin_bit_cast(float4, ge0v), ge1 = __builtin_bit_cast(float4, ge1v);
  // ---- stage x into LDS (optionally fused RMSNorm with HF rounding) while the first ring is in flight
  const bf16_t* xg = (const bf16_t*)a.x;
  // 8 consecutive elements of the input vector; with x_delta the pending residual add of the tensor-parallel path is applied
  // on the fly (HF rounding: bf16(h + bf16(delta))) and workgroup 0 publishes the updated residual stream
  auto with_delta = [&](u32x4 v, int i, bool publish) -> u32x4 {
    if (a.x_delta) {
      const float4 d0 = *(const float4*)(a.x_delta + i), d1 = *(const float4*)(a.x_delta + i + 4);
      const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = pack_bf2(bf2f(v[e] & 0xffff) + round_bf(dl[2 * e]), bf2f(v[e] >> 16) + round_bf(dl[2 * e + 1]));
      if (publish && a.x_out && blockIdx.x == 0) *(u32x4*)((bf16_t*)a.x_out + i) = v;
    }
    return v;
  };
  auto ldx = [&](int i, bool publish) -> u32x4 { return with_delta(*(const u32x4*)(xg + i), i, publish); };
  const int iloop = EARLY ? i0 + NTH * 8 : i0;               // EARLY: the first piece is x0, the loops below take the rest
  if (a.norm_w) {
    float ss = 0.f;
    u32x4 v0 = x0;
    if constexpr (EARLY) {
      if (i0 < K) {
        v0 = with_delta(x0, i0, false);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf2f(v0[e] & 0xffff), hi = bf2f(v0[e] >> 16);
          ss += lo * lo + hi * hi;
        }
      }
    }
    for (int i = iloop; i < K; i += NTH * 8) {
      const u32x4 v = ldx(i, false);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf2f(v[e] & 0xffff), hi = bf2f(v[e] >> 16);
        ss += lo * lo + hi * hi;
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) tot += red[w];
    const float rstd = rsqrtf(tot / (float)K + a.eps);
    if constexpr (EARLY) {
      if (i0 < Kpad) {
        u32x4 o = {0, 0, 0, 0};
        if (i0 < K) {
          if (a.x_delta && a.x_out && blockIdx.x == 0) *(u32x4*)((bf16_t*)a.x_out + i0) = v0;
          const float gw[8] = {ge0.x, ge0.y, ge0.z, ge0.w, ge1.x, ge1.y, ge1.z, ge1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bf2f(v0[e] & 0xffff), hi = bf2f(v0[e] >> 16);
            o[e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
          }
        }
        *(u32x4*)(xs + i0) = o;
      }
    }
    for (int i = iloop; i < Kpad; i += NTH * 8) {
      u32x4 o = {0, 0, 0, 0};
      if (i < K) {
        const u32x4 v = ldx(i, true);
        const float4 g0 = *(const float4*)(a.norm_w + i), g1 = *(const float4*)(a.norm_w + i + 4);
        const float gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf2f(v[e] & 0xffff), hi = bf2f(v[e] >> 16);
          o[e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
        }
      }
      *(u32x4*)(xs + i) = o;
    }
  } else if constexpr (CMB) {
    // ---- (1) the first K/128 workgroups combine one head each: attn_combine_kernel's arithmetic on threads 0..127
    const int NS = a.mrg_ns;
    float* cw = (float*)smem;                                // [64] split weights + [1] 1/l (the x staging area is still free)
    if ((int)blockIdx.x < (K >> 7)) {
      const int hq = blockIdx.x, d = tid;
      if (d < 64) {
        const float mv = d < NS ? a.mrg_pm[hq * NS + d] : -1e30f;
        const float m = wave_max(mv);
        const float e = d < NS ? __expf(mv - m) : 0.f;
        const float l = wave_sum(d < NS ? a.mrg_pl[hq * NS + d] * e : 0.f);
        cw[d] = e;
        if (d == 0) cw[64] = 1.0f / l;
      }
      __syncthreads();
      if (d < 128) {
        const float* p = a.mrg_po + (int64_t)hq * NS * 128 + d;
        float o = 0.f;
        int s = 0;
        for (; s + 4 <= NS; s += 4) {
          const float a0 = p[(s + 0) * 128], a1 = p[(s + 1) * 128], a2 = p[(s + 2) * 128], a3 = p[(s + 3) * 128];
          o += (a0 * cw[s] + a1 * cw[s + 1]) + (a2 * cw[s + 2] + a3 * cw[s + 3]);
        }
        for (; s < NS; ++s) o += p[s * 128] * cw[s];
        const unsigned v = f2bf(o * cw[64]);
        const unsigned hi = __shfl_down(v, 1, 64);           // element d + 1 (d even: same wave)
        if (!(d & 1))
          __hip_atomic_store(a.cmb_gran + hq * 64 + (d >> 1), (1ull << 32) | (unsigned long long)(v | (hi << 16)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();                                       // cw is dead before the gather overwrites the staging area
    }
    // ---- (2) every workgroup gathers the K/2 granules into its LDS copy of x (bounded re-reads of the ones not there yet)
    {
      const unsigned long long tmo = (unsigned long long)(a.cmb_timeout_ms > 0 ? a.cmb_timeout_ms : 200) * 100000ull;
      const unsigned long long t0 = wall_clock64();
      bool late = false;
      for (int i = tid; i < (Kpad >> 1); i += NTH) {
        unsigned val = 0u;
        if (i < (K >> 1)) {
          unsigned long long g = __hip_atomic_load(a.cmb_gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spins = 0;
          while ((unsigned)(g >> 32) != 1u && !late) {
            __builtin_amdgcn_s_sleep(1);
            g = __hip_atomic_load(a.cmb_gran + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((++spins & 63) == 63 && wall_clock64() - t0 > tmo) late = true;
          }
          val = (unsigned)(g >> 32) == 1u ? (unsigned)g : 0u;
        }
        *(unsigned*)(xs + 2 * i) = val;
      }
      if (late && a.cmb_err) atomicOr((int*)a.cmb_err, 1);
    }
  } else if constexpr (MRG) {
    // o_proj of the decode step: x is MERGED here from the context-split attention partials (usdm_gemv_args.mrg_*), four
    // elements of one head per thread; the partial loads of a chunk of 8 splits are requested before the first is used.
    const int NS = a.mrg_ns;
    for (int i = tid * 4; i < Kpad; i += NTH * 4) {
      u32x2 r = {0u, 0u};
      if (i < K) {
        const int hq = i >> 7, d = i & 127;
        const float* pm = a.mrg_pm + hq * NS;
        const float* pl = a.mrg_pl + hq * NS;
        const float* po = a.mrg_po + (int64_t)hq * NS * 128 + d;
        float m = -1e30f;
        for (int s = 0; s < NS; ++s) m = fmaxf(m, pm[s]);
        float l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        for (int s0 = 0; s0 < NS; s0 += 8) {
          float4 p[8];
          float w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int s = min(s0 + u, NS - 1);
            p[u] = *(const float4*)(po + (int64_t)s * 128);
            w[u] = (s0 + u < NS) ? __expf(pm[s] - m) : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            l = fmaf((s0 + u < NS) ? pl[min(s0 + u, NS - 1)] : 0.f, w[u], l);
            o0 = fmaf(p[u].x, w[u], o0); o1 = fmaf(p[u].y, w[u], o1); o2 = fmaf(p[u].z, w[u], o2); o3 = fmaf(p[u].w, w[u], o3);
          }
        }
        const float inv = 1.0f / l;
        r[0] = pack_bf2(o0 * inv, o1 * inv);
        r[1] = pack_bf2(o2 * inv, o3 * inv);
      }
      *(u32x2*)(xs + i) = r;
    }
  } else {
    if constexpr (EARLY) {
      if (i0 < Kpad) {
        u32x4 v = {0, 0, 0, 0};
        if (i0 < K) v = with_delta(x0, i0, true);
        *(u32x4*)(xs + i0) = v;
      }
    }
    for (int i = iloop; i < Kpad; i += NTH * 8) {
      u32x4 v = {0, 0, 0, 0};
      if (i < K) v = ldx(i, true);
      *(u32x4*)(xs + i) = v;
    }
  }
  __syncthreads();
  GTR(2);

  // ---- stream: consume ring slot, immediately refill it UNR iterations ahead
  float acc[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) acc[j] = 0.f;
  for (int it0 = 0; it0 < nit; it0 += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = it0 + u;
      if (it < nit) {
        const u32x4 xv = *(const u32x4*)(xs + (it * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          acc[j] = dot8(ring[j][u], xv, acc[j]);
          if (it + UNR < nit) ring[j][u] = wload(j, it + UNR);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) acc[j] = wave_sum(acc[j]);
  GTR(3);

  // ---- epilogue
  if (a.part_val) {  // lm_head: bf16-rounded logits, ban mask, per-block arg-max (ties -> lowest id)
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int n = ob + j;
      if (n < a.N && !(a.ban && a.ban[n])) {
        const float v = round_bf(acc[j]);
        if (a.y32 && lane == 0) a.y32[n] = v;
        if (v > bv) { bv = v; bi = n; }
      } else if (n < a.N && a.y32 && lane == 0) {
        a.y32[n] = -INFINITY;
      }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NWV; ++w)
        if (sv[w] > bv) { bv = sv[w]; bi = si[w]; }
      a.part_val[blockIdx.x] = bv;
      a.part_idx[blockIdx.x] = bi == 0x7fffffff ? bi : bi + a.idx_offset;
    }
    return;
  }
  if constexpr (P2P && !GLU) {
    // Row-parallel projection of the tensor-parallel decode: the all-reduce of the f32 partial sums happens HERE, between the
    // workgroups that own the same rows on every rank (protocol: include/usdm_hip.h, usdm_allreduce_p2p_*).  No workgroup
    // waits for another workgroup of its own rank, so progress never depends on how much of the grid is resident.
    const usdm_p2p_dev* d = a.p2p;
    constexpr int RPB = NWV * RW;
    float* prow = (float*)smem;          // [RPB] this rank's partials (x in LDS is dead after the K loop)
    float* pg = prow + RPB;              // [world][RPB] gathered partials
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NR; ++j) prow[wave * RW + j] = acc[j];
    }
    __syncthreads();
    const int wb = blockIdx.x * RPB, world = d->world, me = d->rank;
    for (int t = tid; t < world * RPB; t += NTH) {          // put: one granule per (peer, row), coalesced per peer
      const int peer = t / RPB, r = t - peer * RPB;
      if (wb + r < a.N) p2p_put(p2p_slot(d, peer, p2p_epoch, a.p2p_site, me) + wb + r, p2p_epoch, prow[r]);
    }
    if (a.p2p_mode == 2) return;                            // split mode: usdm_allreduce_p2p_reduce finishes
    for (int t0 = 0; t0 < world * RPB; t0 += NTH) {         // get: uniform trip count (p2p_get is a wave-uniform bounded loop)
      const int t = t0 + tid;
      const int src = t / RPB, r = t - src * RPB;
      const bool in = t < world * RPB;
      const bool want = in && wb + r < a.N && src != me;
      const float v = p2p_get(d, p2p_slot(d, me, p2p_epoch, a.p2p_site, want ? src : 0) + (want ? wb + r : 0), p2p_epoch, want,
                              USDM_P2P_ERR_TIMEOUT_ROWS, p2p_failed);
      if (in) pg[src * RPB + r] = (src == me) ? prow[r] : v;
    }
    __syncthreads();
    if (tid < RPB && wb + tid < a.N) {
      const int n = wb + tid;
      float s = 0.f;
      for (int src = 0; src < world; ++src) s += pg[src * RPB + tid];   // fixed rank order: identical bits on every rank
      const float v = round_bf(round_bf(s) + bf2f(((const bf16_t*)a.residual)[n]));
      ((bf16_t*)a.y16)[n] = f2bf(v);
    }
    return;
  }
  if (lane != 0) return;
  if (GLU) {
#pragma unroll
    for (int j = 0; j < RW; ++j) {
      const int o = ob + j;
      if (2 * o >= a.N) continue;
      const float g = acc[j], u = acc[j + RW];
      float r;
      if (a.round_bf16) {
        const float gt = round_bf(g), up = round_bf(u);
        r = round_bf(round_bf(gt / (1.0f + __expf(-gt))) * up);
      } else {
        r = (g / (1.0f + __expf(-g))) * u;
      }
      if (a.y16) ((bf16_t*)a.y16)[o] = f2bf(r);
      if (a.y32) a.y32[o] = r;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int n = ob + j;
      if (n >= a.N) continue;
      float v = acc[j];
      if (a.round_bf16) v = round_bf(v);
      if (a.residual) {
        v += USDM_GEMV_RES_PREFETCH ? bf2f(resraw[j]) : bf2f(((const bf16_t*)a.residual)[n]);
        if (a.round_bf16) v = round_bf(v);
      }
      if (a.y16) ((bf16_t*)a.y16)[n] = f2bf(v);
      if (a.y32) a.y32[n] = v;
    }
  }
}

// Rows per wave: HBM streaming wants >= ~4 workgroups (16 waves) per CU in flight AND a grid that is a whole
// number of workgroups per CU (256 CUs); take the largest RW that gives both, else the best balanced one.
static int gemv_pick_rw(int nout, bool glu) {
  const int ncand = glu ? 2 : 4;
  const int cands[4] = {glu ? 2 : 4, glu ? 1 : 3, 2, 1};
  int best = cands[ncand - 1];
  double best_score = -1.0;
  for (int c = 0; c < ncand; ++c) {
    const int rw = cands[c];
    const int blocks = cdiv(nout, 4 * rw);
    const double eff = (blocks / 256.0) / (double)((blocks + 255) / 256);  // 1.0 = perfectly balanced
    if (blocks >= 1024 && eff >= 0.9) return rw;
    const double score = eff * (blocks >= 512 ? 1.0 : 0.5 + blocks / 1024.0);
    if (score > best_score) { best_score = score; best = rw; }
  }
  return best;
}

// final arg-max over the per-block partials; advances the device-side decode state
// (nseg segments of nparts partials per sequence, seg_stride apart: the tensor-parallel batched step gathers [rank][sequence][nparts])
__global__ void argmax_final_kernel(const float* pv, const int* pi, int nparts, int nseg, int64_t seg_stride, usdm_decode_state st,
                                    const bf16_t* E, int Hd, bf16_t* h_out) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int s_tok;
  const int b = blockIdx.x;   // sequence of a batched step (grid = 1 when single)
  if (st.done && st.done[b]) return;
  pv += (int64_t)b * nparts; pi += (int64_t)b * nparts;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int sg = 0; sg < nseg; ++sg)
    for (int i = threadIdx.x; i < nparts; i += 256) {
      const float v = pv[sg * seg_stride + i];
      const int id = pi[sg * seg_stride + i];
      if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
    }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s];
      const int id = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && id < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = id; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // every id banned (no candidate anywhere): fall back to id 0 so that nothing indexes the embedding table out of range
    const int tok = (si[0] == 0x7fffffff ? 0 : si[0]) + st.id_offset;
    const int step = st.step[b];
    st.next_token[b] = tok;
    if (step < st.max_out) st.out_tokens[(int64_t)b * st.max_out + step] = tok;
    st.step[b] = step + 1;
    if (st.advance_pos) st.pos[b] = st.pos[b] + 1;
    if (st.done && st.eos) {   // device-side EOS: eos = {count, min_new, ids...}
      const int n = st.eos[0], mn = st.eos[1];
      bool hit = false;
      for (int i = 0; i < n && i < 6; ++i) hit |= (st.eos[2 + i] == tok);
      if (hit && step + 1 >= mn) st.done[b] = 1;
    }
    s_tok = tok;
  }
  if (E) {  // fused nn.Embedding lookup of the token the next decode step consumes
    __syncthreads();
    const u32x4* src = (const u32x4*)(E + (int64_t)s_tok * Hd);
    u32x4* dst = (u32x4*)(h_out + (int64_t)b * Hd);
    for (int i = threadIdx.x; i < Hd / 8; i += 256) dst[i] = src[i];
  }
}

// h[0:Hd] = E[token] (bf16 row copy)  — nn.Embedding of HF MistralModel
__global__ void embed_kernel(const bf16_t* E, const int64_t* ids, const int* next_token, int n, int Hd, bf16_t* out) {
  const int r = blockIdx.x;
  const int64_t id = ids ? ids[r] : (int64_t)(*next_token);
  const u32x4* src = (const u32x4*)(E + id * Hd);
  u32x4* dst = (u32x4*)(out + (int64_t)r * Hd);
  for (int i = threadIdx.x; i < Hd / 8; i += blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// RoPE (HF apply_rotary_pos_emb in bf16) helpers.  cos/sin tables are bf16 [maxpos][64].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
  // o1 = x1*cos + (-x2)*sin ; o2 = x2*cos + x1*sin ; every product and sum rounded to bf16
  o1 = round_bf(round_bf(x1 * c) + round_bf(-x2 * s));
  o2 = round_bf(round_bf(x2 * c) + round_bf(x1 * s));
}

// prefill: in-place RoPE of q,k inside qkv [S][(Hq+2Hkv)*128]; K,V appended to the caches; V^T scratch
__global__ void rope_cache_kernel(const usdm_rope_args a) {
  const int s = blockIdx.x, hh = blockIdx.y;  // hh over Hq + 2*Hkv heads
  const int d = threadIdx.x;                  // 0..63
  const int pos = a.pos0 + s;
  bf16_t* row = (bf16_t*)a.qkv + (int64_t)s * a.ld + hh * 128;
  const float c = bf2f(a.cos[(int64_t)pos * 64 + d]), sn = bf2f(a.sin[(int64_t)pos * 64 + d]);
  if (hh < a.Hq) {
    float o1, o2;
    rope_pair(bf2f(row[d]), bf2f(row[d + 64]), c, sn, o1, o2);
    row[d] = f2bf(o1); row[d + 64] = f2bf(o2);
  } else if (hh < a.Hq + a.Hkv) {
    const int kh = hh - a.Hq;
    float o1, o2;
    rope_pair(bf2f(row[d]), bf2f(row[d + 64]), c, sn, o1, o2);
    bf16_t* kc = (bf16_t*)a.kcache + ((int64_t)kh * a.ctx_max + pos) * 128;
    kc[d] = f2bf(o1); kc[d + 64] = f2bf(o2);
  } else {
    const int vh = hh - a.Hq - a.Hkv;
    bf16_t* vc = (bf16_t*)a.vcache + ((int64_t)vh * a.ctx_max + pos) * 128;
    vc[d] = row[d]; vc[d + 64] = row[d + 64];
    if (a.vt) {
      bf16_t* vt = (bf16_t*)a.vt + (int64_t)vh * 128 * a.vt_ld;
      vt[(int64_t)d * a.vt_ld + s] = row[d];
      vt[(int64_t)(d + 64) * a.vt_ld + s] = row[d + 64];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Decode attention, split over the context: grid (Hkv, NS).  Each block ropes the new q (G heads of
// its kv head) and the new k itself, so no block depends on another block's cache write.
// ---------------------------------------------------------------------------------------------
constexpr int DA_KMAX = 512;  // max keys per split
// PIPE (round 4, the many-sequence step: few splits of up to 512 keys each): the K / V rows of the NEXT batch of keys are requested
// before the current batch is consumed (second register set).  The batch-1 step (NS = 32: ~20 keys per split, one batch) keeps
// the plain form.  Same keys per thread in the same order: bit-identical results (profiles/r04_decode_ablation.txt 11).
template <int G, bool PIPE = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(const usdm_attn_decode_args a) {
  __shared__ float qs[G][128];
  __shared__ float knew[128], vnew[128];
  __shared__ float sc[G][DA_KMAX];
  __shared__ float red[8][G][128];
  __shared__ float lsum[G], lmax[G];
  const int skipv = a.skip ? *a.skip : 0;        // tested after the K/V requests below are issued
  const int kh = blockIdx.x, sp = blockIdx.y, NS = gridDim.y;
  const int bi = blockIdx.z;                      // sequence of a batched decode step (0 when single)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = a.pos[bi];
  if (a.cmb_gran) {   // clear the tags of the o_proj hand-off granules (usdm_gemv cmb_gran) for the launch that follows.  BEFORE the
    // position check below: on that (caller-bug) path the o_proj launch must not find the previous token's granules still tagged
    const int gi = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 256 + tid;
    if (bi == 0 && gi < a.Hq * 64) a.cmb_gran[gi] = 0ull;
  }
  if ((unsigned)pos >= (unsigned)a.ctx_max) return;   // never append past the cache / rope table (a caller bug: the host bounds every sequence)
  const int ctx = pos + 1;
  const int lo = (a.window > 0 && ctx > a.window) ? ctx - a.window : 0;   // sliding window: keys lo .. pos
  const int chunk = (ctx - lo + NS - 1) / NS;
  const int k0 = lo + sp * chunk, k1 = min(ctx, k0 + chunk);
  const int nk = max(0, k1 - k0);
  const bf16_t* qkv = (const bf16_t*)a.qkv + (int64_t)bi * a.qkv_bs;
  bf16_t* kcache_b = (bf16_t*)a.kcache + (int64_t)bi * a.cache_bs;
  bf16_t* vcache_b = (bf16_t*)a.vcache + (int64_t)bi * a.cache_bs;
  const bf16_t* Kc = kcache_b + (int64_t)kh * a.ctx_max * 128;
  const bf16_t* Vc = vcache_b + (int64_t)kh * a.ctx_max * 128;
  float* pm_b = a.pm + (int64_t)bi * a.Hq * NS;
  float* pl_b = a.pl + (int64_t)bi * a.Hq * NS;
  float* po_b = a.po + (int64_t)bi * a.Hq * NS * 128;
  bf16_t* out_b = (bf16_t*)a.out + (int64_t)bi * a.out_bs;
  // ---- everything that depends only on pos is requested NOW: the first batch of K rows (scores layout) and of V rows
  // (PV layout) is in flight while q/k are roped; at ~25 keys per split that is the whole split, so the kernel pays one
  // memory latency instead of three (rope inputs -> K -> V).
  constexpr int SW = 4, PW = 4;
  const int j = lane & 7, gk = (wave << 3) + (lane >> 3);  // scores: 8 lanes per key, key slot within a 32-key sweep
  const int d4 = (tid & 31) * 4, kl = tid >> 5;            // PV: thread = (4 d's, key lane)
  u32x4 r0[SW], r1[SW];
  u32x2 rv[PW];
  if (nk > 0) {
#pragma unroll
    for (int w = 0; w < SW; ++w) {
      const int kk = min(32 * w + gk, nk - 1);
      const bf16_t* kp = Kc + (int64_t)(k0 + kk) * 128 + j * 16;
      r0[w] = *(const u32x4*)kp;
      r1[w] = *(const u32x4*)(kp + 8);
    }
#pragma unroll
    for (int w = 0; w < PW; ++w) {
      const int kk = min(8 * w + kl, nk - 1);
      rv[w] = *(const u32x2*)(Vc + (int64_t)(k0 + kk) * 128 + d4);
    }
  }
  if (skipv) return;   // sequence already ended (usdm_decode_state.done): nothing may be appended to the cache
  // ---- rope q (G heads) and the new k; stash v
  for (int i = tid; i < (G + 1) * 64; i += 256) {
    const int hsel = i >> 6, d = i & 63;
    const float c = bf2f(a.cos[(int64_t)pos * 64 + d]), sn = bf2f(a.sin[(int64_t)pos * 64 + d]);
    const bf16_t* src = hsel < G ? qkv + (kh * G + hsel) * 128 : qkv + (a.Hq + kh) * 128;
    float o1, o2;
    rope_pair(bf2f(src[d]), bf2f(src[d + 64]), c, sn, o1, o2);
    if (hsel < G) { qs[hsel][d] = o1; qs[hsel][d + 64] = o2; }
    else { knew[d] = o1; knew[d + 64] = o2; }
  }
  if (tid < 128) vnew[tid] = bf2f(qkv[(a.Hq + a.Hkv + kh) * 128 + tid]);
  __syncthreads();
  if (sp == 0 && tid < 128) {  // designated writer of the new cache row
    kcache_b[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(knew[tid]);
    vcache_b[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(vnew[tid]);
  }
  // ---- scores: 8 lanes per key, 16 d each; the K rows of SW sweeps are requested before any of them is used
  {
    float qr[G][16];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) qr[h][e] = qs[h][j * 16 + e];
    u32x4 n0[PIPE ? SW : 1], n1[PIPE ? SW : 1];      // PIPE: the next batch's rows
    for (int base = 0; base < nk; base += 32 * SW) {
      if constexpr (PIPE) {
        // unconditional (clamped) requests: a branch here would make the wait-count pass drain everything at the join
#pragma unroll
        for (int w = 0; w < SW; ++w) {
          const int kk = min(base + 32 * SW + 32 * w + gk, nk - 1);
          const bf16_t* kp = Kc + (int64_t)(k0 + kk) * 128 + j * 16;
          n0[w] = *(const u32x4*)kp;
          n1[w] = *(const u32x4*)(kp + 8);
        }
      } else if (base > 0) {
#pragma unroll
        for (int w = 0; w < SW; ++w) {
          const int kk = min(base + 32 * w + gk, nk - 1);
          const bf16_t* kp = Kc + (int64_t)(k0 + kk) * 128 + j * 16;
          r0[w] = *(const u32x4*)kp;
          r1[w] = *(const u32x4*)(kp + 8);
        }
      }
#pragma unroll
      for (int w = 0; w < SW; ++w) {
        const int kk = base + 32 * w + gk;
        if (kk >= nk) continue;
        const bool is_new = (k0 + kk) == pos;     // the new token's K is not in the cache yet: take it from LDS
        float kv[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kv[2 * e] = bf2f(r0[w][e] & 0xffff); kv[2 * e + 1] = bf2f(r0[w][e] >> 16);
          kv[8 + 2 * e] = bf2f(r1[w][e] & 0xffff); kv[8 + 2 * e + 1] = bf2f(r1[w][e] >> 16);
        }
        if (is_new) {
#pragma unroll
          for (int e = 0; e < 16; ++e) kv[e] = knew[j * 16 + e];
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
          float sdot = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) sdot = fmaf(qr[h][e], kv[e], sdot);
          sdot += __shfl_xor(sdot, 1, 64); sdot += __shfl_xor(sdot, 2, 64); sdot += __shfl_xor(sdot, 4, 64);
          if (j == 0) sc[h][kk] = sdot * a.scale;
        }
      }
      if constexpr (PIPE) {
#pragma unroll
        for (int w = 0; w < SW; ++w) { r0[w] = n0[w]; r1[w] = n1[w]; }
      }
    }
  }
  __syncthreads();
  // ---- softmax statistics per head (wave h <-> head h when G <= 4)
  for (int h = wave; h < G; h += 4) {
    float m = -1e30f;
    for (int kk = lane; kk < nk; kk += 64) m = fmaxf(m, sc[h][kk]);
    m = wave_max(m);
    float l = 0.f;
    for (int kk = lane; kk < nk; kk += 64) {
      const float p = __expf(sc[h][kk] - m);
      l += p;
      sc[h][kk] = round_bf(p);  // P is bf16 for the PV product (flash-attention semantics), l stays fp32
    }
    l = wave_sum(l);
    if (lane == 0) { lsum[h] = l; lmax[h] = m; }
  }
  __syncthreads();
  // ---- PV: thread = (4 d's, key lane)
  {
    float acc[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = 0.f;
    u32x2 nv[PIPE ? PW : 1];
    for (int base = 0; base < nk; base += 8 * PW) {
      if constexpr (PIPE) {
#pragma unroll
        for (int w = 0; w < PW; ++w) {
          const int kk = min(base + 8 * PW + 8 * w + kl, nk - 1);
          nv[w] = *(const u32x2*)(Vc + (int64_t)(k0 + kk) * 128 + d4);
        }
      } else if (base > 0) {
#pragma unroll
        for (int w = 0; w < PW; ++w) {
          const int kk = min(base + 8 * w + kl, nk - 1);
          rv[w] = *(const u32x2*)(Vc + (int64_t)(k0 + kk) * 128 + d4);
        }
      }
#pragma unroll
      for (int w = 0; w < PW; ++w) {
        const int kk = base + 8 * w + kl;
        if (kk >= nk) continue;
        float v[4] = {bf2f(rv[w][0] & 0xffff), bf2f(rv[w][0] >> 16), bf2f(rv[w][1] & 0xffff), bf2f(rv[w][1] >> 16)};
        if (k0 + kk == pos) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vnew[d4 + e];
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
          const float p = sc[h][kk];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][e] = fmaf(p, v[e], acc[h][e]);
        }
      }
      if constexpr (PIPE) {
#pragma unroll
        for (int w = 0; w < PW; ++w) rv[w] = nv[w];
      }
    }
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[kl][h][d4 + e] = acc[h][e];
  }
  __syncthreads();
  for (int i = tid; i < G * 128; i += 256) {
    const int h = i >> 7, d = i & 127;
    float s = 0.f;
#pragma unroll
    for (int kl2 = 0; kl2 < 8; ++kl2) s += red[kl2][h][d];
    const int hq = kh * G + h;
    // agent-scope (write-through) stores: the merging workgroup may sit on another XCD, whose L2 is not coherent with ours
    __hip_atomic_store(po_b + ((int64_t)hq * NS + sp) * 128 + d, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == 0) {
      __hip_atomic_store(pm_b + hq * NS + sp, nk > 0 ? lmax[h] : -1e30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pl_b + hq * NS + sp, nk > 0 ? lsum[h] : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!a.counters) return;   // partials are merged by attn_combine_kernel
  // ---- fused merge: the workgroup that finishes this kv head last combines the NS partials of its G heads.
  // No agent-scope fences (an L2 write-back / invalidate per workgroup cost ~30 us per launch): the partials travel
  // as write-through stores and are read back with agent-scope loads; ordering = every thread waits for its own
  // stores to be acknowledged, workgroup barrier, then one relaxed increment of the kv head's counter.
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(a.counters + bi * a.Hkv + kh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == NS - 1);
    if (old == NS - 1) __hip_atomic_store(a.counters + bi * a.Hkv + kh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  {
    float* wgt = &sc[0][0];                       // [G][64] merge weights (sc is free now; DA_KMAX >= 64)
    for (int h = wave; h < G; h += 4) {           // wave per head: NS <= 64 lanes
      const int hq = kh * G + h;
      const float mv = lane < NS ? __hip_atomic_load(pm_b + hq * NS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1e30f;
      const float m = wave_max(mv);
      const float e = lane < NS ? __expf(mv - m) : 0.f;
      const float l = wave_sum(lane < NS ? __hip_atomic_load(pl_b + hq * NS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * e : 0.f);
      wgt[h * 64 + lane] = e;
      if (lane == 0) lsum[h] = 1.0f / l;
    }
    __syncthreads();
    for (int i = tid; i < G * 128; i += 256) {
      const int h = i >> 7, d = i & 127;
      const int hq = kh * G + h;
      const float* p = po_b + (int64_t)hq * NS * 128 + d;
      auto ld = [&](int si) { return __hip_atomic_load(p + si * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      float o = 0.f;
      int s = 0;
      for (; s + 16 <= NS; s += 16) {             // 16 partials in flight per output; same association as attn_combine_kernel
        float pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = ld(s + u);
#pragma unroll
        for (int u = 0; u < 16; u += 4)
          o += (pv[u] * wgt[h * 64 + s + u] + pv[u + 1] * wgt[h * 64 + s + u + 1]) +
               (pv[u + 2] * wgt[h * 64 + s + u + 2] + pv[u + 3] * wgt[h * 64 + s + u + 3]);
      }
      for (; s + 4 <= NS; s += 4) {
        const float a0 = ld(s), a1 = ld(s + 1), a2 = ld(s + 2), a3 = ld(s + 3);
        o += (a0 * wgt[h * 64 + s] + a1 * wgt[h * 64 + s + 1]) + (a2 * wgt[h * 64 + s + 2] + a3 * wgt[h * 64 + s + 3]);
      }
      for (; s < NS; ++s) o += ld(s) * wgt[h * 64 + s];
      out_b[hq * 128 + d] = f2bf(o * lsum[h]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Decode attention, one 16-wave workgroup per kv head (no context split, no partials, no combine launch):
// at batch 1 the phase is latency-bound, not bandwidth-bound (<= ~0.8 MB of K/V per kv head), so the fastest form is
// the one with the fewest dependent steps: rope q/k -> scores (16 waves x 8 keys per sweep) -> softmax (wave per head)
// -> PV (32 key lanes x 32 d-quads) -> in-LDS reduction -> bf16 output.  Same numerics as the split version.
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(1024) void attn_decode1_kernel(const usdm_attn_decode_args a) {
  if (a.skip && *a.skip) return;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  float* sc = (float*)dsm;                          // [G][ctx_pad]
  const int pos = *a.pos;
  if ((unsigned)pos >= (unsigned)a.ctx_max) return;
  const int ctx = pos + 1;
  const int ctx_pad = (a.ctx_max + 3) & ~3;
  float* red = sc + G * ctx_pad;                    // [16][G][128]
  __shared__ float qs[G][128];
  __shared__ float knew[128], vnew[128];
  __shared__ float lsum[G], lmax[G];
  const int kh = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  for (int i = tid; i < (G + 1) * 64; i += 1024) {
    const int hsel = i >> 6, d = i & 63;
    const float c = bf2f(a.cos[(int64_t)pos * 64 + d]), sn = bf2f(a.sin[(int64_t)pos * 64 + d]);
    const bf16_t* src = hsel < G ? qkv + (kh * G + hsel) * 128 : qkv + (a.Hq + kh) * 128;
    float o1, o2;
    rope_pair(bf2f(src[d]), bf2f(src[d + 64]), c, sn, o1, o2);
    if (hsel < G) { qs[hsel][d] = o1; qs[hsel][d + 64] = o2; }
    else { knew[d] = o1; knew[d + 64] = o2; }
  }
  if (tid >= 512 && tid < 640) vnew[tid - 512] = bf2f(qkv[(a.Hq + a.Hkv + kh) * 128 + tid - 512]);
  __syncthreads();
  const bf16_t* Kc = (const bf16_t*)a.kcache + (int64_t)kh * a.ctx_max * 128;
  const bf16_t* Vc = (const bf16_t*)a.vcache + (int64_t)kh * a.ctx_max * 128;
  if (tid < 128) {
    ((bf16_t*)a.kcache)[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(knew[tid]);
    ((bf16_t*)a.vcache)[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(vnew[tid]);
  }
  // ---- scores
  {
    constexpr int SW = 2;
    const int j = lane & 7, gk = (wave << 3) + (lane >> 3);   // 128 keys per block sweep
    float qr[G][16];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) qr[h][e] = qs[h][j * 16 + e];
    for (int base = 0; base < ctx; base += 128 * SW) {
      u32x4 r0[SW], r1[SW];
#pragma unroll
      for (int w = 0; w < SW; ++w) {
        const int kk = min(base + 128 * w + gk, ctx - 1);
        const bf16_t* kp = Kc + (int64_t)kk * 128 + j * 16;
        r0[w] = *(const u32x4*)kp;
        r1[w] = *(const u32x4*)(kp + 8);
      }
#pragma unroll
      for (int w = 0; w < SW; ++w) {
        const int kk = base + 128 * w + gk;
        if (kk >= ctx) continue;
        float kv[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kv[2 * e] = bf2f(r0[w][e] & 0xffff); kv[2 * e + 1] = bf2f(r0[w][e] >> 16);
          kv[8 + 2 * e] = bf2f(r1[w][e] & 0xffff); kv[8 + 2 * e + 1] = bf2f(r1[w][e] >> 16);
        }
        if (kk == pos) {
#pragma unroll
          for (int e = 0; e < 16; ++e) kv[e] = knew[j * 16 + e];
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
          float sdot = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) sdot = fmaf(qr[h][e], kv[e], sdot);
          sdot += __shfl_xor(sdot, 1, 64); sdot += __shfl_xor(sdot, 2, 64); sdot += __shfl_xor(sdot, 4, 64);
          if (j == 0) sc[h * ctx_pad + kk] = sdot * a.scale;
        }
      }
    }
  }
  __syncthreads();
  // ---- softmax statistics: wave h <-> head h
  if (wave < G) {
    const int h = wave;
    float m = -1e30f;
    for (int kk = lane; kk < ctx; kk += 64) m = fmaxf(m, sc[h * ctx_pad + kk]);
    m = wave_max(m);
    float l = 0.f;
    for (int kk = lane; kk < ctx; kk += 64) {
      const float p = __expf(sc[h * ctx_pad + kk] - m);
      l += p;
      sc[h * ctx_pad + kk] = round_bf(p);
    }
    l = wave_sum(l);
    if (lane == 0) { lsum[h] = l; lmax[h] = m; }
  }
  __syncthreads();
  // ---- PV: 32 key lanes x 32 d-quads; lanes l and l^32 of a wave share the d-quad
  {
    constexpr int PW = 4;
    const int d4 = (tid & 31) * 4, kl = tid >> 5;   // kl in [0, 32)
    float acc[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = 0.f;
    for (int base = 0; base < ctx; base += 32 * PW) {
      u32x2 rv[PW];
#pragma unroll
      for (int w = 0; w < PW; ++w) {
        const int kk = min(base + 32 * w + kl, ctx - 1);
        rv[w] = *(const u32x2*)(Vc + (int64_t)kk * 128 + d4);
      }
#pragma unroll
      for (int w = 0; w < PW; ++w) {
        const int kk = base + 32 * w + kl;
        if (kk >= ctx) continue;
        float v[4] = {bf2f(rv[w][0] & 0xffff), bf2f(rv[w][0] >> 16), bf2f(rv[w][1] & 0xffff), bf2f(rv[w][1] >> 16)};
        if (kk == pos) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = vnew[d4 + e];
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
          const float p = sc[h * ctx_pad + kk];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][e] = fmaf(p, v[e], acc[h][e]);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[h][e];
        v += __shfl_xor(v, 32, 64);
        if (lane < 32) red[(wave * G + h) * 128 + d4 + e] = v;
      }
  }
  __syncthreads();
  for (int i = tid; i < G * 128; i += 1024) {
    const int h = i >> 7, d = i & 127;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[(w * G + h) * 128 + d];
    ((bf16_t*)a.out)[(kh * G + h) * 128 + d] = f2bf(s / lsum[h]);
  }
}

__global__ __launch_bounds__(128) void attn_combine_kernel(const float* pm, const float* pl, const float* po, int NS, bf16_t* out,
                                                           int64_t out_bs, const int* skip) {
  if (skip && *skip) return;
  __shared__ float w[64];
  __shared__ float linv;
  const int hq = blockIdx.x, d = threadIdx.x;  // 128 threads
  const int bi = blockIdx.y, Hq = gridDim.x;   // sequence of a batched step
  pm += (int64_t)bi * Hq * NS; pl += (int64_t)bi * Hq * NS; po += (int64_t)bi * Hq * NS * 128; out += (int64_t)bi * out_bs;
  if (d < 64) {
    const float mv = d < NS ? pm[hq * NS + d] : -1e30f;
    const float m = wave_max(mv);
    const float e = d < NS ? __expf(mv - m) : 0.f;
    const float l = wave_sum(d < NS ? pl[hq * NS + d] * e : 0.f);
    w[d] = e;
    if (d == 0) linv = 1.0f / l;
  }
  __syncthreads();
  const float* p = po + (int64_t)hq * NS * 128 + d;
  float o = 0.f;
  int s = 0;
  for (; s + 4 <= NS; s += 4) {
    const float a0 = p[(s + 0) * 128], a1 = p[(s + 1) * 128], a2 = p[(s + 2) * 128], a3 = p[(s + 3) * 128];
    o += (a0 * w[s] + a1 * w[s + 1]) + (a2 * w[s + 2] + a3 * w[s + 3]);
  }
  for (; s < NS; ++s) o += p[s * 128] * w[s];
  out[hq * 128 + d] = f2bf(o * linv);
}

// h = bf16(h + bf16(delta))  — residual add after a tensor-parallel all-reduce of fp32 partial sums
__global__ void residual_add_kernel(bf16_t* h, const float* delta, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) h[i] = f2bf(bf2f(h[i]) + round_bf(delta[i]));
}
}  // namespace

extern "C" int usdm_gemv(const usdm_gemv_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->W && (pa->x || pa->mrg_po), "usdm_gemv: null args");
  const usdm_gemv_args& a = *pa;
  USDM_CHECK_ARG(a.N > 0 && a.K > 0 && a.K % 8 == 0 && a.ldw % 8 == 0 && a.ldw >= a.K, "usdm_gemv: bad N/K/ldw");
  USDM_CHECK_ARG(a.K <= 16384, "usdm_gemv: K too large for the LDS-resident input vector");
  const bool glu = a.act == USDM_ACT_SWIGLU;
  USDM_CHECK_ARG(!glu || a.N % 32 == 0, "usdm_gemv: swiglu needs N %% 32 == 0");
  USDM_CHECK_ARG(a.y16 || a.y32 || a.part_val, "usdm_gemv: no output");
  USDM_CHECK_ARG(!a.part_val || (a.part_idx && !glu), "usdm_gemv: part_idx missing / lm_head mode is not GLU");
  USDM_CHECK_ARG(!a.norm_w || a.K % 8 == 0, "usdm_gemv: K");
  USDM_CHECK_ARG(!a.x_out || (a.x_delta && a.x_out != a.x), "usdm_gemv: x_out needs x_delta and must not alias x");
  USDM_CHECK_ARG(!a.mrg_po || (a.mrg_pm && a.mrg_pl && a.mrg_ns >= 1 && a.mrg_ns <= 64 && a.K % 128 == 0 && !a.norm_w && !a.x_delta),
                 "usdm_gemv: merged-attention input needs pm/pl/po, 1 <= splits <= 64, K a multiple of 128, no norm / x_delta");
  USDM_CHECK_ARG(a.p2p_mode >= 0 && a.p2p_mode <= 2, "usdm_gemv: p2p_mode");
  USDM_CHECK_ARG(!a.p2p_mode || (a.p2p && a.p2p_site >= 0 && a.act == USDM_ACT_NONE && !a.part_val && a.residual && a.y16),
                 "usdm_gemv: the fused all-reduce needs a plain row-parallel projection with residual + y16");
  // (its LDS scratch, 9 x rows-per-workgroup floats <= 864 B, reuses the x staging area: Kpad * 2 >= 1024 B always)
  const int nout = glu ? a.N / 2 : a.N;
  const int Kpad = (a.K + 511) & ~511;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)Kpad * 2;
  // Wide workgroups for the mid-size projections: one workgroup per CU with 12-16 waves stages x (and the fused
  // RMSNorm) once per 16-24 rows instead of once per 4, at the same number of loads in flight.
  if (a.cmb_gran) {   // hand-off form of the merged-attention input
    USDM_CHECK_ARG(a.mrg_po && a.mrg_pm && a.mrg_pl && a.mrg_ns >= 1 && a.mrg_ns <= 64 && !a.p2p_mode && !glu && !a.part_val && !a.norm_w && !a.x_delta &&
                       nout % 256 == 0 && nout / 256 == 16 && a.K % 128 == 0 && a.K / 128 <= 256,
                   "usdm_gemv: cmb_gran needs the mrg_* partials, a plain 4096-output projection and K / 128 <= 256 heads");
    hipLaunchKernelGGL((gemv_kernel<1, false, 16, false, false, true>), dim3(256), dim3(1024), lds, st, a);
    USDM_LAUNCH_CHECK();
    return 0;
  }
  if (a.mrg_po || a.p2p_mode) {   // o_proj with the attention merge in its prologue and / or a row-parallel projection with the
    // peer-to-peer all-reduce in its epilogue: the 4096-output shape of the 7B (one 16-wave workgroup per CU) or the general form
    USDM_CHECK_ARG(!glu && !a.part_val, "usdm_gemv: merged-attention input / fused all-reduce are for plain projections");
    const bool big = nout % 256 == 0 && nout / 256 == 16;
    const dim3 gb(256), bb(1024), gs(cdiv(nout, 4)), bs(256);
    if (a.mrg_po && a.p2p_mode) {
      if (big) hipLaunchKernelGGL((gemv_kernel<1, false, 16, true, true>), gb, bb, lds, st, a);
      else hipLaunchKernelGGL((gemv_kernel<1, false, 4, true, true>), gs, bs, lds, st, a);
    } else if (a.mrg_po) {
      if (big) hipLaunchKernelGGL((gemv_kernel<1, false, 16, true, false>), gb, bb, lds, st, a);
      else hipLaunchKernelGGL((gemv_kernel<1, false, 4, true, false>), gs, bs, lds, st, a);
    } else {
      if (big) hipLaunchKernelGGL((gemv_kernel<1, false, 16, false, true>), gb, bb, lds, st, a);
      else hipLaunchKernelGGL((gemv_kernel<1, false, 4, false, true>), gs, bs, lds, st, a);
    }
    USDM_LAUNCH_CHECK();
    return 0;
  }
  if (!glu && !a.part_val && nout % 256 == 0) {
    const int rows_per_cu = nout / 256;
    if (rows_per_cu == 16) {
      hipLaunchKernelGGL((gemv_kernel<1, false, 16>), dim3(256), dim3(1024), lds, st, a);
      USDM_LAUNCH_CHECK();
      return 0;
    }
    if (rows_per_cu == 24) {
      hipLaunchKernelGGL((gemv_kernel<2, false, 12>), dim3(256), dim3(768), lds, st, a);
      USDM_LAUNCH_CHECK();
      return 0;
    }
  }
  // (a 14-wave GLU variant with one workgroup per CU was measured 15 % slower than 7 four-wave workgroups per CU)
  // gate/up of the 7B (14336 outputs): 7-wave workgroups of 14 outputs = 1024 workgroups = exactly two rounds of two
  // workgroups per CU, instead of 1792 four-wave workgroups = 1.75 rounds of four
  if (glu && nout % 14 == 0 && (nout / 14) % 512 == 0) {
    hipLaunchKernelGGL((gemv_kernel<2, true, 7>), dim3(nout / 14), dim3(448), lds, st, a);
    USDM_LAUNCH_CHECK();
    return 0;
  }
  const int rw = a.part_val ? 4 : gemv_pick_rw(nout, glu);
  dim3 grid(cdiv(nout, 4 * rw)), block(256);
  if (glu) {
    if (rw == 2) hipLaunchKernelGGL((gemv_kernel<2, true, 4>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gemv_kernel<1, true, 4>), grid, block, lds, st, a);
  } else {
    if (rw == 4) hipLaunchKernelGGL((gemv_kernel<4, false, 4>), grid, block, lds, st, a);
    else if (rw == 3) hipLaunchKernelGGL((gemv_kernel<3, false, 4>), grid, block, lds, st, a);
    else if (rw == 2) hipLaunchKernelGGL((gemv_kernel<2, false, 4>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gemv_kernel<1, false, 4>), grid, block, lds, st, a);
  }
  USDM_LAUNCH_CHECK();
  return 0;
}
#ifdef USDM_GEMV_TRACE
extern "C" int usdm_dbg_gemv_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemv_trace), sizeof(unsigned long long) * n);
}
#endif

extern "C" int usdm_gemv_nblocks(int32_t N, int32_t act) { return cdiv(N, 16); }

// threads per workgroup of the variant usdm_gemv picks for this projection (the fused RMSNorm's partial sums follow that
// partition; usdm_gemv_engine reproduces it to stay bit-identical).  Keep in step with usdm_gemv below.
extern "C" int usdm_gemv_threads(const usdm_gemv_args* pa) {
  const usdm_gemv_args& a = *pa;
  const bool glu = a.act == USDM_ACT_SWIGLU;
  const int nout = glu ? a.N / 2 : a.N;
  if (a.mrg_po || a.p2p_mode) return (nout % 256 == 0 && nout / 256 == 16) ? 1024 : 256;
  if (!glu && !a.part_val && nout % 256 == 0) {
    if (nout / 256 == 16) return 1024;
    if (nout / 256 == 24) return 768;
  }
  if (glu && nout % 14 == 0 && (nout / 14) % 512 == 0) return 448;
  return 256;
}

extern "C" int usdm_argmax_final_seg(const float* part_val, const int32_t* part_idx, int32_t nparts, int32_t nseg, int64_t seg_stride,
                                     const usdm_decode_state* st, const void* embed_table, int32_t Hd, void* h_out,
                                     usdm_stream_t stream) {
  USDM_CHECK_ARG(part_val && part_idx && nparts > 0 && st && st->next_token && st->out_tokens && st->step && st->pos,
                 "usdm_argmax_final: bad args");
  USDM_CHECK_ARG(nseg >= 1 && (nseg == 1 || seg_stride >= (int64_t)nparts * (st->batch > 1 ? st->batch : 1)), "usdm_argmax_final_seg: segments overlap");
  USDM_CHECK_ARG(!embed_table || (h_out && Hd > 0 && Hd % 8 == 0), "usdm_argmax_final: embedding output missing");
  USDM_CHECK_ARG(st->batch >= 0 && st->batch <= 64, "usdm_argmax_final: batch");
  hipLaunchKernelGGL(argmax_final_kernel, dim3(st->batch > 1 ? st->batch : 1), dim3(256), 0, (hipStream_t)stream, part_val, part_idx, nparts, nseg,
                     seg_stride, *st, (const bf16_t*)embed_table, Hd, (bf16_t*)h_out);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_argmax_final(const float* part_val, const int32_t* part_idx, int32_t nparts,
                                 const usdm_decode_state* st, const void* embed_table, int32_t Hd, void* h_out,
                                 usdm_stream_t stream) {
  return usdm_argmax_final_seg(part_val, part_idx, nparts, 1, 0, st, embed_table, Hd, h_out, stream);
}

extern "C" int usdm_embed_rows(const void* table, const int64_t* ids, const int32_t* next_token, int32_t n, int32_t Hd,
                               void* out, usdm_stream_t stream) {
  USDM_CHECK_ARG(table && out && (ids || next_token) && n > 0 && Hd % 8 == 0, "usdm_embed_rows: bad args");
  hipLaunchKernelGGL(embed_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table, ids, next_token, n, Hd,
                     (bf16_t*)out);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_rope_cache(const usdm_rope_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->qkv && pa->cos && pa->sin && pa->kcache && pa->vcache, "usdm_rope_cache: null args");
  const usdm_rope_args& a = *pa;
  USDM_CHECK_ARG(a.S > 0 && a.pos0 >= 0 && a.pos0 + a.S <= a.ctx_max && a.pos0 + a.S <= a.max_pos, "usdm_rope_cache: positions exceed the cache / rope table");
  USDM_CHECK_ARG(!a.vt || a.vt_ld >= a.S, "usdm_rope_cache: vt_ld");
  hipLaunchKernelGGL(rope_cache_kernel, dim3(a.S, a.Hq + 2 * a.Hkv), dim3(64), 0, (hipStream_t)stream, a);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_attn_decode(const usdm_attn_decode_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->qkv && pa->pos && pa->kcache && pa->vcache && (pa->out || pa->defer_merge), "usdm_attn_decode: null args");
  USDM_CHECK_ARG(!pa->defer_merge || (pa->NS > 1 && !pa->counters && pa->batch <= 1), "usdm_attn_decode: defer_merge needs NS > 1, no counters, one sequence");
  USDM_CHECK_ARG(pa->NS == 1 || (pa->pm && pa->pl && pa->po), "usdm_attn_decode: partial buffers missing");
  const usdm_attn_decode_args& a = *pa;
  USDM_CHECK_ARG(a.Hkv > 0 && a.Hq % a.Hkv == 0 && a.NS > 0 && a.NS <= 64, "usdm_attn_decode: heads / NS (<= 64)");
  USDM_CHECK_ARG(a.batch <= 1 || (a.NS > 1 && a.batch <= 64 && a.qkv_bs > 0 && a.out_bs > 0 && a.cache_bs > 0),
                 "usdm_attn_decode: batched form needs NS > 1 and the three strides");
  USDM_CHECK_ARG(a.window >= 0 && (a.window == 0 || a.NS > 1), "usdm_attn_decode: window >= 0, and only with the split form (NS > 1)");
  const int span = (a.window > 0 && a.window < a.ctx_max) ? a.window : a.ctx_max;      // most keys a step can see
  USDM_CHECK_ARG(a.NS == 1 || cdiv(span, a.NS) <= DA_KMAX, "usdm_attn_decode: visible keys / NS exceeds %d keys per split", DA_KMAX);
  const int G = a.Hq / a.Hkv;
  hipStream_t st = (hipStream_t)stream;
  if (a.NS == 1) {   // single-workgroup-per-kv-head form: no partials, no combine
    const int ctx_pad = (a.ctx_max + 3) & ~3;
    const size_t lds = (size_t)(G * ctx_pad + 16 * G * 128) * sizeof(float);
    USDM_CHECK_ARG(lds <= 120 * 1024, "usdm_attn_decode: ctx_max too large for the one-workgroup form (use NS > 1)");
    static bool attr_set = false;
    if (!attr_set) {   // allow > 64 KiB of dynamic LDS (gfx950 has 160 KiB per workgroup)
      (void)hipFuncSetAttribute((const void*)attn_decode1_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_decode1_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      (void)hipFuncSetAttribute((const void*)attn_decode1_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      attr_set = true;
    }
    if (G == 4) hipLaunchKernelGGL(attn_decode1_kernel<4>, dim3(a.Hkv), dim3(1024), lds, st, a);
    else if (G == 2) hipLaunchKernelGGL(attn_decode1_kernel<2>, dim3(a.Hkv), dim3(1024), lds, st, a);
    else if (G == 1) hipLaunchKernelGGL(attn_decode1_kernel<1>, dim3(a.Hkv), dim3(1024), lds, st, a);
    else { usdm_set_error("usdm_attn_decode: group size %d unsupported (1,2,4)", G); return 2; }
    USDM_LAUNCH_CHECK();
    return 0;
  }
  const int nbatch = a.batch > 1 ? a.batch : 1;
  dim3 grid(a.Hkv, a.NS, nbatch);
  const bool pipe = nbatch > 1 && cdiv(span, a.NS) > 64;      // long splits (the many-sequence step): next batch of keys prefetched
  if (pipe && G == 4) hipLaunchKernelGGL((attn_decode_kernel<4, true>), grid, dim3(256), 0, st, a);
  else if (pipe && G == 2) hipLaunchKernelGGL((attn_decode_kernel<2, true>), grid, dim3(256), 0, st, a);
  else if (pipe && G == 1) hipLaunchKernelGGL((attn_decode_kernel<1, true>), grid, dim3(256), 0, st, a);
  else if (G == 4) hipLaunchKernelGGL(attn_decode_kernel<4>, grid, dim3(256), 0, st, a);
  else if (G == 2) hipLaunchKernelGGL(attn_decode_kernel<2>, grid, dim3(256), 0, st, a);
  else if (G == 1) hipLaunchKernelGGL(attn_decode_kernel<1>, grid, dim3(256), 0, st, a);
  else { usdm_set_error("usdm_attn_decode: group size %d unsupported (1,2,4)", G); return 2; }
  USDM_LAUNCH_CHECK();
  if (!a.counters && !a.defer_merge) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(a.Hq, nbatch), dim3(128), 0, st, a.pm, a.pl, a.po, a.NS, (bf16_t*)a.out, a.out_bs, a.skip);
    USDM_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int usdm_residual_add(void* h, const float* delta, int32_t n, usdm_stream_t stream) {
  USDM_CHECK_ARG(h && delta && n > 0, "usdm_residual_add: bad args");
  hipLaunchKernelGGL(residual_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)h, delta, n);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_gemv_args(void) { return (int)sizeof(usdm_gemv_args); }
extern "C" int usdm_sizeof_decode_state(void) { return (int)sizeof(usdm_decode_state); }
extern "C" int usdm_sizeof_rope_args(void) { return (int)sizeof(usdm_rope_args); }
extern "C" int usdm_sizeof_attn_decode_args(void) { return (int)sizeof(usdm_attn_decode_args); }
