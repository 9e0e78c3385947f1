// Chained decode GEMVs in one persistent launch (include/usdm_hip_experimental.h, usdm_gemv_chain).
//
// Why: a batch-1 decode GEMV is pure weight streaming, and at every launch boundary HBM drains and ramps up again (~2.6 us of
// dispatch + drain and ~4.6 us until the first ring of loads has arrived, profiles/r01_decode_ablation.txt).  The weights do not
// depend on the activations, only the input vector does.  So consecutive projections run in ONE resident grid: a wave that has
// finished its rows of phase p requests the first ring of its rows of phase p+1 BEFORE it waits at the grid barrier for the
// phase-p vector to be complete, and the barrier latency is spent with loads in flight.
//
// Grid: 512 workgroups x 7 waves (two per CU, 64 ring registers per lane), all resident.  A workgroup tile is 14 consecutive
// outputs (7 waves x 2): for SwiGLU a wave streams 2 gate + 2 up rows (ring 4 x 4 loads), else 2 rows (ring 2 x 8 loads).
// Hand-offs between workgroups follow the write-through recipe (cdna_hip_programming.md G16): outputs leave as 4-byte agent-scope
// (sc1) stores, every storing wave drains them, ONE lane adds to the arrival counter; consumers poll the counter (bounded) and read
// the handed-off vectors only with agent-scope loads.
#include "common.h"
#include "../../include/usdm_hip_experimental.h"

namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
constexpr int CW = 7;             // waves per workgroup
constexpr int CTH = CW * 64;      // 448 threads
constexpr int CGRID = 512;        // resident workgroups (2 per CU)
constexpr int TILE = 2 * CW;      // outputs per workgroup tile

__device__ __forceinline__ int ntiles_of(const usdm_gemv_args& a) {
  const int nout = a.act == USDM_ACT_SWIGLU ? a.N / 2 : a.N;
  return (nout + TILE - 1) / TILE;
}

__device__ __forceinline__ float cdot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = w[i], b = x[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  return acc;
}

// 8 consecutive bf16 of a vector that another workgroup may have written in this launch: agent-scope (sc1) loads only
__device__ __forceinline__ u32x4 ld_shared8(const bf16_t* p) {
  const unsigned long long lo = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}

template <bool GLU>
struct Shape {
  static constexpr int NR = GLU ? 4 : 2;      // weight rows a wave streams together
  static constexpr int UNR = GLU ? 4 : 8;     // ring depth: NR * UNR = 16 loads in flight per lane
};

// weight row pointers of this wave for workgroup tile `tile` of phase `a`
// (32-bit byte offsets from the uniform base a.W: the loads then use the scalar-base + vector-offset form, 4 VGPRs instead of 8)
template <bool GLU>
__device__ __forceinline__ void row_ptrs(const usdm_gemv_args& a, int tile, int wave, int lane, unsigned (&wp)[4]) {
  const int ob = tile * TILE + wave * 2;
#pragma unroll
  for (int j = 0; j < Shape<GLU>::NR; ++j) {
    int r;
    if (GLU) {   // packed layout: blocks of 32 rows = 16 gate + 16 up
      const int o = ob + (j & 1);
      r = (o >> 4) * 32 + (o & 15) + (j >= 2 ? 16 : 0);
    } else {
      r = ob + j;
    }
    r = r < a.N ? r : a.N - 1;
    wp[j] = (unsigned)(((int64_t)r * a.ldw + lane * 8) * 2);
  }
}

template <bool GLU>
__device__ __forceinline__ void issue_ring(const char* W, const unsigned (&wp)[4], u32x4 (&ring)[16]) {
  constexpr int NR = Shape<GLU>::NR, UNR = Shape<GLU>::UNR;
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int j = 0; j < NR; ++j) ring[j * UNR + u] = __builtin_nontemporal_load((const u32x4*)(W + wp[j] + u * 1024));   // (nit >= UNR: launcher)
}

// K loop + epilogue of one workgroup tile whose first ring is already in flight
template <bool GLU>
__device__ __forceinline__ void run_tile(const usdm_gemv_args& a, int tile, const unsigned (&wp)[4], u32x4 (&ring)[16],
                                         const bf16_t* xs, float* outs, int tid, int lane, int wave) {
  constexpr int NR = Shape<GLU>::NR, UNR = Shape<GLU>::UNR;
  const int nit = a.K >> 9;
  const char* W = (const char*)a.W;
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
          acc[j] = cdot8(ring[j * UNR + u], xv, acc[j]);
          if (it + UNR < nit) ring[j * UNR + u] = __builtin_nontemporal_load((const u32x4*)(W + wp[j] + (it + UNR) * 1024));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) acc[j] = wave_sum(acc[j]);
  // epilogue: the wave's 2 outputs -> LDS -> 4-byte write-through stores (7 dwords per tile)
  const int nout = GLU ? a.N / 2 : a.N;
  const int ob = tile * TILE + wave * 2;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = ob + j;
      float r = 0.f;
      if (o < nout) {
        if (GLU) {
          const float g = acc[j], u = acc[j + 2];
          if (a.round_bf16) {
            const float gt = round_bf(g), up = round_bf(u);
            r = round_bf(round_bf(gt / (1.0f + __expf(-gt))) * up);
          } else {
            r = (g / (1.0f + __expf(-g))) * u;
          }
        } else {
          r = acc[j];
          if (a.round_bf16) r = round_bf(r);
          if (a.residual) {
            // the residual stream may have been written by another workgroup earlier in this launch: agent-scope load
            const unsigned w = __hip_atomic_load((const unsigned*)((const bf16_t*)a.residual + (o & ~1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r += bf2f((o & 1) ? (bf16_t)(w >> 16) : (bf16_t)(w & 0xffff));
            if (a.round_bf16) r = round_bf(r);
          }
        }
      }
      outs[wave * 2 + j] = r;
    }
  }
  __syncthreads();
  if (tid < CW) {
    const int o = tile * TILE + tid * 2;
    if (o < nout) {   // nout is even and tiles start at even outputs: a pair never straddles the end
      const unsigned v = pack_bf2(outs[tid * 2], outs[tid * 2 + 1]);
      __hip_atomic_store((unsigned*)((bf16_t*)a.y16 + o), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();   // outs is reused by the next tile
}

// stage the phase's input vector into LDS (optionally RMS-normalised with HF rounding), reading it with agent-scope loads
__device__ __forceinline__ void stage_x(const usdm_gemv_args& a, bf16_t* xs, float* red, int tid, int lane, int wave) {
  const int K = a.K;
  const bf16_t* xg = (const bf16_t*)a.x;
  if (a.norm_w) {
    float ss = 0.f;
#pragma unroll 1
    for (int i = tid * 8; i < K; i += CTH * 8) {
      const u32x4 v = ld_shared8(xg + i);
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
    for (int w = 0; w < CW; ++w) tot += red[w];
    const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll 1
    for (int i = tid * 8; i < K; i += CTH * 8) {
      const u32x4 v = ld_shared8(xg + i);
      const float4 g0 = *(const float4*)(a.norm_w + i), g1 = *(const float4*)(a.norm_w + i + 4);
      const float gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf2f(v[e] & 0xffff), hi = bf2f(v[e] >> 16);
        o[e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
      }
      *(u32x4*)(xs + i) = o;
    }
  } else {
#pragma unroll 2
    for (int i = tid * 8; i < K; i += CTH * 8) *(u32x4*)(xs + i) = ld_shared8(xg + i);
  }
  __syncthreads();
}

// one phase, its kind known at compile time (the kernel is instantiated per chain pattern: a runtime GLU / plain switch inside one
// function made the register allocator spill ~200 registers)
template <bool GLU>
__device__ __forceinline__ void run_phase(const usdm_gemv_chain_args& c, int p, unsigned gen, bool& failed, unsigned long long tmo,
                                          bf16_t* xs, float* red, float* outs, int tid, int lane, int wave) {
  const usdm_gemv_args& a = c.ph[p];
  const int ntiles = ntiles_of(a);
  u32x4 ring[16];
  unsigned wp[4];
  // ---- first ring of this phase: requested BEFORE the wait for its input vector (weights do not depend on activations)
  if ((int)blockIdx.x < ntiles) {
    row_ptrs<GLU>(a, blockIdx.x, wave, lane, wp);
    issue_ring<GLU>((const char*)a.W, wp, ring);
  }
  if (p > 0) {
    // ---- grid barrier p-1: the whole output of phase p-1 must be complete
    if (wave == 0) {
      const unsigned target = (gen + 1u) * (unsigned)CGRID;
      bool ok = failed;
      const unsigned long long t0 = wall_clock64();
      for (unsigned spins = 0; !ok; ++spins) {
        const unsigned v = __hip_atomic_load(c.sync + 2 + (p - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (int)(v - target) >= 0;
        if (!ok) {
          if ((spins & 31) == 31 && wall_clock64() - t0 > tmo) {
            if (lane == 0) __hip_atomic_fetch_or(c.sync + 1, (unsigned)USDM_CHAIN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
  }
  stage_x(a, xs, red, tid, lane, wave);
  // ---- this workgroup's tiles of phase p
  for (int tile = blockIdx.x; tile < ntiles; tile += CGRID) {
    if (tile != (int)blockIdx.x) {     // (the first tile's ring is already in flight)
      row_ptrs<GLU>(a, tile, wave, lane, wp);
      issue_ring<GLU>((const char*)a.W, wp, ring);
    }
    run_tile<GLU>(a, tile, wp, ring, xs, outs, tid, lane, wave);
  }
  if (p + 1 < c.nph) {
    // ---- publish: drain this workgroup's stores, then ONE lane arrives (the next phase requests its ring, then waits)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(c.sync + 2 + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // the last arriver of the LAST barrier opens the next generation (every workgroup has read `gen` long ago)
      if (p + 2 == c.nph && old + 1u == (gen + 1u) * (unsigned)CGRID)
        __hip_atomic_store(c.sync + 0, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// MASK bit p = phase p is a SwiGLU projection
template <int NPH, unsigned MASK>
__global__ __launch_bounds__(CTH, 4) void gemv_chain_kernel(const usdm_gemv_chain_args c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;                       // [Kmax] bf16 input vector of the current phase
  __shared__ float red[8];
  __shared__ float outs[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int skipv = c.ph[0].skip ? *c.ph[0].skip : 0;
  const unsigned gen = __hip_atomic_load(c.sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool failed = __hip_atomic_load(c.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  const unsigned long long tmo = (unsigned long long)c.timeout_ms * 100000ull;
  if (skipv) return;   // the sequence ended (usdm_decode_state.done): uniform over the grid, nobody arrives, nobody waits
  if constexpr (NPH > 0) { if constexpr (MASK & 1u) run_phase<true>(c, 0, gen, failed, tmo, xs, red, outs, tid, lane, wave); else run_phase<false>(c, 0, gen, failed, tmo, xs, red, outs, tid, lane, wave); }
  if constexpr (NPH > 1) { if constexpr (MASK & 2u) run_phase<true>(c, 1, gen, failed, tmo, xs, red, outs, tid, lane, wave); else run_phase<false>(c, 1, gen, failed, tmo, xs, red, outs, tid, lane, wave); }
  if constexpr (NPH > 2) { if constexpr (MASK & 4u) run_phase<true>(c, 2, gen, failed, tmo, xs, red, outs, tid, lane, wave); else run_phase<false>(c, 2, gen, failed, tmo, xs, red, outs, tid, lane, wave); }
  if constexpr (NPH > 3) { if constexpr (MASK & 8u) run_phase<true>(c, 3, gen, failed, tmo, xs, red, outs, tid, lane, wave); else run_phase<false>(c, 3, gen, failed, tmo, xs, red, outs, tid, lane, wave); }
}
}  // namespace

extern "C" int usdm_gemv_chain(const usdm_gemv_chain_args* pc, usdm_stream_t stream) {
  USDM_CHECK_ARG(pc && pc->nph >= 1 && pc->nph <= USDM_CHAIN_MAX_PHASES && pc->sync && pc->timeout_ms > 0, "usdm_gemv_chain: nph (1..4) / sync / timeout_ms");
  int kmax = 0;
  for (int p = 0; p < pc->nph; ++p) {
    const usdm_gemv_args& a = pc->ph[p];
    const bool glu = a.act == USDM_ACT_SWIGLU;
    USDM_CHECK_ARG(a.W && a.x && a.y16 && a.N > 0, "usdm_gemv_chain: phase %d needs W, x and a bf16 output", p);
    USDM_CHECK_ARG(a.K >= 4096 && a.K % 512 == 0 && a.K <= 16384 && a.ldw % 8 == 0 && a.ldw >= a.K, "usdm_gemv_chain: phase %d: K must be a multiple of 512 in [4096, 16384]", p);
    USDM_CHECK_ARG(a.act == USDM_ACT_NONE || glu, "usdm_gemv_chain: phase %d: activation", p);
    USDM_CHECK_ARG((int64_t)a.N * a.ldw * 2 < 0xFFFF0000ll, "usdm_gemv_chain: phase %d: weight matrix exceeds the 4 GiB offset range", p);
    USDM_CHECK_ARG(!glu || (a.N % 32 == 0 && !a.residual), "usdm_gemv_chain: phase %d: swiglu needs N %% 32 == 0 and no residual", p);
    USDM_CHECK_ARG(glu || a.N % 2 == 0, "usdm_gemv_chain: phase %d: N must be even", p);
    USDM_CHECK_ARG(!a.part_val && !a.ban && !a.y32 && !a.x_delta && !a.x_out && !a.p2p_mode && !a.mrg_po,
                   "usdm_gemv_chain: phase %d: lm_head / f32 output / x_delta / p2p / merge modes are not chainable", p);
    USDM_CHECK_ARG(((uintptr_t)a.y16 % 4) == 0 && ((uintptr_t)a.x % 16) == 0 && (!a.residual || ((uintptr_t)a.residual % 4) == 0), "usdm_gemv_chain: phase %d: alignment", p);
    USDM_CHECK_ARG(p == 0 || a.skip == pc->ph[0].skip, "usdm_gemv_chain: every phase must share phase 0's skip word");
    if (a.K > kmax) kmax = a.K;
  }
  unsigned mask = 0;
  for (int p = 0; p < pc->nph; ++p)
    if (pc->ph[p].act == USDM_ACT_SWIGLU) mask |= 1u << p;
  const dim3 g(CGRID), b(CTH);
  const size_t lds = (size_t)kmax * 2;
  hipStream_t st = (hipStream_t)stream;
  // the chain patterns of the decode step (others: add an instantiation)
  if (pc->nph == 2 && mask == 1u) hipLaunchKernelGGL((gemv_chain_kernel<2, 1u>), g, b, lds, st, *pc);            // gate/up -> down
  else if (pc->nph == 3 && mask == 2u) hipLaunchKernelGGL((gemv_chain_kernel<3, 2u>), g, b, lds, st, *pc);       // o -> gate/up -> down
  else if (pc->nph == 4 && mask == 2u) hipLaunchKernelGGL((gemv_chain_kernel<4, 2u>), g, b, lds, st, *pc);       // o -> gate/up -> down -> qkv
  else if (pc->nph == 2 && mask == 0u) hipLaunchKernelGGL((gemv_chain_kernel<2, 0u>), g, b, lds, st, *pc);       // plain -> plain
  else if (pc->nph == 1 && mask == 0u) hipLaunchKernelGGL((gemv_chain_kernel<1, 0u>), g, b, lds, st, *pc);
  else if (pc->nph == 1 && mask == 1u) hipLaunchKernelGGL((gemv_chain_kernel<1, 1u>), g, b, lds, st, *pc);
  else { usdm_set_error("usdm_gemv_chain: no instantiation for %d phases with SwiGLU mask %u", pc->nph, mask); return 2; }
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_gemv_chain_args(void) { return (int)sizeof(usdm_gemv_chain_args); }
