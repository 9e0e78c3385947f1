// Loader / consumer engine for chained decode projections (include/usdm_hip_experimental.h, usdm_gemv_engine).
//
// One 8-wave workgroup per CU (256 workgroups, 157 KB of LDS each, so exactly one is resident per CU); NL loader waves and NC
// consumer waves (4 + 4):
//   LOADER wave l       owns the stream slots s = l (mod NL); it streams this CU's share of every phase's weight rows, in order, into a ring of 7 x 16 KiB LDS slots by
//                       LDS-DMA (buffer_load_dwordx4 ... lds, non-temporal: every byte is read once by one CU).  It keeps two
//                       slots in flight behind a counted s_waitcnt vmcnt and publishes a slot (FULL word in LDS) once its DMAs
//                       have landed.  It never looks at activations, so it runs ahead across phase boundaries as far as the ring
//                       allows: HBM keeps streaming while the consumers wait for the previous phase's vector.
//   CONSUMER waves      take jobs round-robin.  A job is the ring slots of one output PAIR (2 plain rows = 1 slot at K = 4096,
//                       2 x (gate, up) = 2 slots for SwiGLU, 2 rows = 4 slots at K = 14336); each slot is multiplied against the
//                       phase's input vector in LDS as soon as it is FULL and handed back (FREE word) once its bytes are in
//                       registers.  The pair is published as plain bf16 (for later launches) and as ONE 8-byte granule
//                       {tag = epoch, 2 x bf16} - a single aligned agent-scope store, the data is its own flag.
//   the first consumer also GATHERS: before its first job of a phase it sweeps the previous phase's granules from all CUs into this CU's LDS
//                       copy of the input vector (re-reading granules whose tag is not the epoch yet, bounded), applies the fused
//                       RMSNorm, and sets the phase's READY word.
// Nothing in here is a grid barrier; all inter-workgroup traffic is granules, all intra-workgroup hand-offs are LDS words.
// Per output row the arithmetic is usdm_gemv's, bit for bit (lane l owns 16-byte pieces l, l + 64, ... of K and accumulates them
// in order; the RMSNorm partial sums follow the thread partition of the usdm_gemv variant that would run the projection).
#include "common.h"
#include "../../include/usdm_hip_experimental.h"

namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
constexpr int NSLOT = 7, SLOTB = 16384;
constexpr int XR0 = NSLOT * SLOTB, XR1 = XR0 + 8192, XR2 = XR1 + 8192;   // input-vector regions: 8 KB, 8 KB, 28 KB
constexpr int CTL = XR2 + 28672;                                         // control words + CU-local residual values
constexpr int ELDS = CTL + 1024;                                         // 160768 bytes
constexpr int NCU = 256;
#ifndef USDM_ENG_NL
#define USDM_ENG_NL 4   // measured best split of 8 waves (profiles/r02_decode_ablation.txt section 4)
#endif
#ifndef USDM_ENG_NC
#define USDM_ENG_NC 4
#endif
constexpr int NL = USDM_ENG_NL, NC = USDM_ENG_NC;   // loader waves (stream slot s belongs to loader s % NL) and consumer waves
// own slots a loader may leave in flight behind the one it publishes; every loader holds at most LDEPTH + 1 unpublished ring slots
// and NL * (LDEPTH + 1) must not exceed the ring (a loader waiting for a FREE slot that another loader has not published yet
// would otherwise wait for ever)
constexpr int LDEPTH = (NL * 3 <= 7) ? 2 : ((NL * 2 <= 7) ? 1 : 0);
// control word indices (unsigned, in LDS)
constexpr int W_FULL = 0, W_FREE = 8, W_READY = 16, W_ABORT = 24, W_HLOC = 32;   // hloc: 64 floats from word 32

struct Ph {   // per-phase geometry, derived from usdm_gemv_args (uniform)
  bool glu, kindB;
  int nit, pps, spj, nout, upc, njobs, nslots, xreg;
};

__device__ __forceinline__ Ph make_ph(const usdm_gemv_args& a, int xreg) {
  Ph p;
  p.glu = a.act == USDM_ACT_SWIGLU;
  p.nit = a.K >> 9;
  p.kindB = p.nit > 16;
  p.pps = p.kindB ? p.nit / 2 : 16;           // 1-KiB pieces per slot
  p.spj = p.kindB ? 4 : (p.glu ? 2 : 1);      // slots per job (one output pair)
  p.nout = p.glu ? a.N / 2 : a.N;
  p.upc = p.nout / NCU;                       // outputs of this CU
  p.njobs = p.upc / 2;
  p.nslots = p.njobs * p.spj;
  p.xreg = xreg;
  return p;
}

__device__ __forceinline__ float edot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = w[i], b = x[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  return acc;
}

// control words live in LDS and are touched only through address-space-3 volatile accesses (ds_read / ds_write: a generic
// volatile pointer would compile to flat_* instructions, which also count on vmcnt and return out of order)
typedef volatile __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned lds_ld(lds_u32* p) { return *p; }

// bounded wait on an LDS word: returns false (and raises the workgroup's abort word + the global error word) on timeout
template <typename F>
__device__ __forceinline__ bool lds_wait(lds_u32* ctl, unsigned* gerr, unsigned long long tmo, F&& done) {
  if (done()) return true;
  const unsigned long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    __builtin_amdgcn_s_sleep(1);
    if (done()) return true;
    if (lds_ld(ctl + W_ABORT)) return false;
    if ((spins & 63) == 63 && wall_clock64() - t0 > tmo) {
      ctl[W_ABORT] = 1u;
      __hip_atomic_fetch_or(gerr, (unsigned)USDM_CHAIN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

__device__ __forceinline__ void vm_wait(int n) {   // s_waitcnt vmcnt(n), n uniform in [0, 48]
  switch (n) {
#define VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    VMC(0) VMC(1) VMC(2) VMC(3) VMC(4) VMC(5) VMC(6) VMC(7) VMC(8) VMC(9) VMC(10) VMC(11) VMC(12) VMC(13) VMC(14) VMC(15) VMC(16)
    VMC(17) VMC(18) VMC(19) VMC(20) VMC(21) VMC(22) VMC(23) VMC(24) VMC(25) VMC(26) VMC(27) VMC(28) VMC(29) VMC(30) VMC(31) VMC(32)
    VMC(33) VMC(34) VMC(35) VMC(36) VMC(37) VMC(38) VMC(39) VMC(40) VMC(41) VMC(42) VMC(43) VMC(44) VMC(45) VMC(46) VMC(47) VMC(48)
#undef VMC
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

#ifdef USDM_ENG_TRACE
__device__ unsigned long long g_eng_trace[64];
#define ETR_T0() const unsigned long long etr_t0 = clock64()
#define ETR_ADD(i) do { if (blockIdx.x == 0 && lane == 0) g_eng_trace[i] += clock64() - etr_t0; } while (0)
#else
#define ETR_T0() do { } while (0)
#define ETR_ADD(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------- loader
__device__ __forceinline__ void loader(const usdm_gemv_chain_args& c, char* smem, lds_u32* ctl, unsigned long long tmo, int lw, int lane) {
  const int cu = blockIdx.x;
  int s = 0;                 // stream slot counter of this CU over all phases
  int pend1 = 0, pend2 = 0;  // pieces of this loader's two youngest issued slots
  int mine1 = -1, mine2 = -1, mine3 = -1;   // this loader's three youngest issued stream slots (newest first)
  int xsmall = 0;
  for (int p = 0; p < c.nph; ++p) {
    const usdm_gemv_args& a = c.ph[p];
    const Ph ph = make_ph(a, 0);
    (void)xsmall;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, 0xFFFFFF00u, 0x00020000);
    const unsigned ldwb = (unsigned)(a.ldw * 2);
    for (int js = 0; js < ph.nslots; ++js, ++s) {
      if (s % NL != lw) continue;
      const int i = s % NSLOT;
      // the ring slot must have been handed back by its consumer (stream slot s - NSLOT)
      if (s >= NSLOT) {
        const unsigned need = (unsigned)(s - NSLOT + 1);
        ETR_T0();
        if (!lds_wait(ctl, c.sync + 1, tmo, [&]() { return lds_ld(ctl + W_FREE + i) >= need; })) return;
        ETR_ADD(lw * 4 + 0);          // loader: waiting for a free ring slot
      }
      ETR_T0();
      // rows of this slot
      unsigned r0, r1 = 0;
      int it0 = 0;
      if (ph.kindB) {          // one row over two slots
        r0 = (unsigned)(cu * ph.upc + (js >> 1));
        it0 = (js & 1) * ph.pps;
      } else if (ph.glu) {     // gate and up row of one unit (packed layout: blocks of 32 rows = 16 gate + 16 up)
        const int u = cu * ph.upc + js;
        r0 = (unsigned)((u >> 4) * 32 + (u & 15));
        r1 = r0 + 16;
      } else {                 // two consecutive rows
        r0 = (unsigned)(cu * ph.upc + 2 * js);
        r1 = r0 + 1;
      }
      char* dst = smem + i * SLOTB;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < ph.pps) {
          unsigned off;
          if (ph.kindB) off = r0 * ldwb + (unsigned)((it0 + k) * 1024) + (unsigned)(lane * 16);
          else off = (k < 8 ? r0 : r1) * ldwb + (unsigned)((k & 7) * 1024) + (unsigned)(lane * 16);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, off, 0, 0, 2 /* nt */);
        }
      }
      ETR_ADD(lw * 4 + 1);            // loader: address arithmetic + DMA issue
      // publish the slot LDEPTH issues back: everything of this loader's own but its LDEPTH youngest slots has landed
      {
        ETR_T0();
        if (LDEPTH == 0) {
          vm_wait(0);
          if (lane == 0) ctl[W_FULL + i] = (unsigned)(s + 1);
        } else if (LDEPTH == 1) {
          if (mine1 >= 0) { vm_wait(ph.pps); if (lane == 0) ctl[W_FULL + mine1 % NSLOT] = (unsigned)(mine1 + 1); }
        } else {
          if (mine2 >= 0) { vm_wait(ph.pps + pend1); if (lane == 0) ctl[W_FULL + mine2 % NSLOT] = (unsigned)(mine2 + 1); }
        }
        ETR_ADD(lw * 4 + 2);          // loader: waiting for DMAs to land
      }
      mine3 = mine2; mine2 = mine1; mine1 = s;
      pend2 = pend1; pend1 = ph.pps;
    }
  }
  (void)mine3; (void)pend2;
  if (LDEPTH >= 2 && mine2 >= 0) { vm_wait(pend1); if (lane == 0) ctl[W_FULL + mine2 % NSLOT] = (unsigned)(mine2 + 1); }
  if (LDEPTH >= 1 && mine1 >= 0) { vm_wait(0); if (lane == 0) ctl[W_FULL + mine1 % NSLOT] = (unsigned)(mine1 + 1); }
}

// ---------------------------------------------------------------------------------------------------------------- gather
// wave 1: bring the input vector of phase p into LDS region xs (bf16 [K]) and, if the phase has a fused RMSNorm, normalise it
__device__ __forceinline__ void gather(const usdm_gemv_chain_args& c, int p, int src_phase, bf16_t* xs, lds_u32* ctl, unsigned epoch_src,
                       unsigned long long tmo, bool& failed, int lane) {
  const usdm_gemv_args& a = c.ph[p];
  const int K = a.K;
  if (src_phase < 0) {   // produced by an earlier launch: plain loads
    const bf16_t* xg = (const bf16_t*)a.x;
    for (int i = lane * 8; i < K; i += 512) *(u32x4*)(xs + i) = *(const u32x4*)(xg + i);
  } else {               // produced in this launch by every CU: sweep the granules until every tag is the epoch
    const unsigned long long* g = (const unsigned long long*)c.gran + (int64_t)src_phase * 8192;
    const int ng = K / 2;
    for (int base = 0; base < ng; base += 64 * 16) {   // 16 granules per lane and pass (8 KB)
      unsigned long long v[16];
      unsigned okm = 0;       // bit k: granule k of this lane has arrived (a bit mask: a bool array here gets promoted to LDS)
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (failed || (base + k * 64 + lane >= ng)) okm |= 1u << k;
      const unsigned long long t0 = wall_clock64();
      for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (!(okm & (1u << k))) v[k] = __hip_atomic_load(g + base + k * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (!(okm & (1u << k)) && (unsigned)(v[k] >> 32) == epoch_src) {
            okm |= 1u << k;
            *(unsigned*)(xs + 2 * (base + k * 64 + lane)) = (unsigned)v[k];     // one ds_write_b32 per granule
          }
        }
        const bool all = okm == 0xffffu;
        if (__all(all)) break;
        if ((spins & 15) == 15 && (lds_ld(ctl + W_ABORT) || wall_clock64() - t0 > tmo)) {
          if (lane == 0) {
            ctl[W_ABORT] = 1u;
            __hip_atomic_fetch_or(c.sync + 1, (unsigned)USDM_CHAIN_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (a.norm_w) {
    // sum of squares in the thread partition of the usdm_gemv variant that would run this projection: thread t owns the
    // 16-byte pieces t, t + NTH, ...; per wave a DPP tree; waves added in order
    const int nth = c.norm_nth[p], nv = nth >> 6;
    float tot = 0.f;
    for (int v = 0; v < nv; ++v) {
      float ss = 0.f;
      for (int i = (v * 64 + lane) * 8; i < K; i += nth * 8) {
        const u32x4 q = *(const u32x4*)(xs + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf2f(q[e] & 0xffff), hi = bf2f(q[e] >> 16);
          ss += lo * lo + hi * hi;
        }
      }
      tot += wave_sum(ss);
    }
    const float rstd = rsqrtf(tot / (float)K + a.eps);
    for (int i = lane * 8; i < K; i += 512) {
      const u32x4 q = *(const u32x4*)(xs + i);
      const float4 g0 = *(const float4*)(a.norm_w + i), g1 = *(const float4*)(a.norm_w + i + 4);
      const float gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf2f(q[e] & 0xffff), hi = bf2f(q[e] >> 16);
        o[e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
      }
      *(u32x4*)(xs + i) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (lane == 0) ctl[W_READY + p] = 1u;
}

// ---------------------------------------------------------------------------------------------------------------- consumer
__device__ __forceinline__ void consumer(const usdm_gemv_chain_args& c, char* smem, lds_u32* ctl, unsigned gen, unsigned long long tmo,
                         bool failed, int cw, int lane) {
  const int cu = blockIdx.x;
  float* hloc = (float*)(smem + CTL + 4 * W_HLOC);
  int sbase = 0, nsmall = 0;
  for (int p = 0; p < c.nph; ++p) {
    const usdm_gemv_args& a = c.ph[p];
    const int xreg = (a.K * 2 > 8192) ? XR2 : ((nsmall++ & 1) ? XR1 : XR0);
    const Ph ph = make_ph(a, xreg);
    bf16_t* xs = (bf16_t*)(smem + xreg);
    // where does the input vector come from: the output of an earlier phase of this launch (granules) or an earlier launch
    int src = -1;
    for (int q = 0; q < p; ++q)
      if (c.ph[q].y16 == a.x) src = q;
    // residual of this CU's own rows: an earlier phase of this launch (kept in LDS) or an earlier launch (global)
    int rsrc = -1;
    if (a.residual)
      for (int q = 0; q < p; ++q)
        if (c.ph[q].y16 == a.residual) rsrc = q;
    bool publish = false, keep = false;   // is this phase's output the input / the residual of a later phase?
    for (int q = p + 1; q < c.nph; ++q) {
      if (c.ph[q].x == a.y16) publish = true;
      if (c.ph[q].residual == a.y16) keep = true;
    }
    if (cw == 0) gather(c, p, src, xs, ctl, gen * 4u + (unsigned)src + 1u, tmo, failed, lane);
    else if (!lds_wait(ctl, c.sync + 1, tmo, [&]() { return lds_ld(ctl + W_READY + p) != 0u; })) return;
    const unsigned epoch = gen * 4u + (unsigned)p + 1u;
    for (int job = cw; job < ph.njobs; job += NC) {
      const int u0 = cu * ph.upc + 2 * job;     // outputs u0, u0 + 1
      float r[2];
      if (!ph.kindB) {
        // K = 4096: a slot holds two rows (8 pieces each); SwiGLU: one slot per output (gate, up), else both outputs in one slot
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !ph.glu) break;
          const int s = sbase + job * ph.spj + h;
          const int i = s % NSLOT;
          {
            ETR_T0();
            if (!lds_wait(ctl, c.sync + 1, tmo, [&]() { return lds_ld(ctl + W_FULL + i) == (unsigned)(s + 1); })) return;
            ETR_ADD(32 + cw * 4 + 0);   // consumer: waiting for a full slot
          }
          ETR_T0();
          const char* slot = smem + i * SLOTB;
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const u32x4 xv = *(const u32x4*)(xs + (it * 64 + lane) * 8);
            const u32x4 w0 = *(const u32x4*)(slot + it * 1024 + lane * 16);
            const u32x4 w1 = *(const u32x4*)(slot + (8 + it) * 1024 + lane * 16);
            a0 = edot8(w0, xv, a0);
            a1 = edot8(w1, xv, a1);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) ctl[W_FREE + i] = (unsigned)(s + 1);      // the slot's bytes are in registers: hand it back
          a0 = wave_sum(a0); a1 = wave_sum(a1);
          ETR_ADD(32 + cw * 4 + 1);     // consumer: reads + dots + reduction of one slot
          if (ph.glu) {
            if (a.round_bf16) {
              const float gt = round_bf(a0), up = round_bf(a1);
              r[h] = round_bf(round_bf(gt / (1.0f + __expf(-gt))) * up);
            } else {
              r[h] = (a0 / (1.0f + __expf(-a0))) * a1;
            }
          } else {
            r[0] = a0; r[1] = a1;
          }
        }
      } else {
        // deep K: a row spans two slots; two rows (four slots) per job
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float acc = 0.f;
#pragma unroll 1
          for (int hf = 0; hf < 2; ++hf) {
            const int s = sbase + job * 4 + h * 2 + hf;
            const int i = s % NSLOT;
            if (!lds_wait(ctl, c.sync + 1, tmo, [&]() { return lds_ld(ctl + W_FULL + i) == (unsigned)(s + 1); })) return;
            const char* slot = smem + i * SLOTB;
            for (int k = 0; k < ph.pps; ++k) {
              const int it = hf * ph.pps + k;
              const u32x4 xv = *(const u32x4*)(xs + (it * 64 + lane) * 8);
              const u32x4 w = *(const u32x4*)(slot + k * 1024 + lane * 16);
              acc = edot8(w, xv, acc);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) ctl[W_FREE + i] = (unsigned)(s + 1);
          }
          r[h] = wave_sum(acc);
        }
      }
      if (!ph.glu) {
        // plain projection epilogue: bf16 rounding, residual add (HF rounding points)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v = r[h];
          if (a.round_bf16) v = round_bf(v);
          if (a.residual) {
            const float res = rsrc >= 0 ? hloc[2 * job + h] : bf2f(((const bf16_t*)a.residual)[u0 + h]);
            v += res;
            if (a.round_bf16) v = round_bf(v);
          }
          r[h] = v;
        }
      }
      if (lane == 0) {
        const unsigned pair = pack_bf2(r[0], r[1]);
        *(unsigned*)((bf16_t*)a.y16 + u0) = pair;                                   // plain copy for later launches
        if (publish)
          __hip_atomic_store((unsigned long long*)c.gran + (int64_t)p * 8192 + (u0 >> 1), ((unsigned long long)epoch << 32) | pair,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (keep) {                                                                  // this CU's own rows, for a later residual add
          hloc[2 * job] = bf2f((bf16_t)(pair & 0xffff));
          hloc[2 * job + 1] = bf2f((bf16_t)(pair >> 16));
        }
      }
    }
    sbase += ph.nslots;
  }
}

__global__ __launch_bounds__((NL + NC) * 64) void gemv_engine_kernel(const usdm_gemv_chain_args c) {
  __shared__ __attribute__((aligned(16))) char smem[ELDS];   // (static: the >64 KiB dynamic-LDS attribute is refused for this kernel)
  lds_u32* ctl = (lds_u32*)(__attribute__((address_space(3))) char*)(smem + CTL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int skipv = c.ph[0].skip ? *c.ph[0].skip : 0;
  const unsigned gen = __hip_atomic_load(c.sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool failed = __hip_atomic_load(c.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (skipv) return;     // uniform over the grid: the sequence ended (usdm_decode_state.done)
  if (tid < 32) ctl[tid] = 0u;   // FULL / FREE / READY / ABORT words
  __syncthreads();
  const unsigned long long tmo = (unsigned long long)c.timeout_ms * 100000ull;
  if (wave < NL) loader(c, smem, ctl, tmo, wave, lane);
  else consumer(c, smem, ctl, gen, tmo, failed, wave - NL, lane);
  // the gathering wave of workgroup 0 opens the next generation: it can only get here after every workgroup has published
  // (hence started and read `gen`) whenever the chain has a hand-off; single-phase launches do not use the generation
  if (blockIdx.x == 0 && wave == NL && lane == 0 && c.nph > 1)
    __hip_atomic_store(c.sync + 0, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

#ifdef USDM_ENG_TRACE
extern "C" int usdm_dbg_eng_trace(unsigned long long* host, int reset) {
  if (reset) { unsigned long long z[64] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_eng_trace), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_eng_trace), sizeof(unsigned long long) * 64);
}
#endif

extern "C" int usdm_gemv_engine(const usdm_gemv_chain_args* pc, usdm_stream_t stream) {
  USDM_CHECK_ARG(pc && pc->nph >= 1 && pc->nph <= USDM_CHAIN_MAX_PHASES && pc->sync && pc->gran && pc->timeout_ms > 0,
                 "usdm_gemv_engine: nph (1..4) / sync / gran / timeout_ms");
  for (int p = 0; p < pc->nph; ++p) {
    const usdm_gemv_args& a = pc->ph[p];
    const bool glu = a.act == USDM_ACT_SWIGLU;
    const int nout = glu ? a.N / 2 : a.N, nit = a.K / 512;
    USDM_CHECK_ARG(a.W && a.x && a.y16 && a.N > 0, "usdm_gemv_engine: phase %d needs W, x and a bf16 output", p);
    USDM_CHECK_ARG(a.act == USDM_ACT_NONE || glu, "usdm_gemv_engine: phase %d: activation", p);
    USDM_CHECK_ARG(a.K % 512 == 0 && (nit == 8 || (!glu && nit > 16 && nit <= 28 && nit % 2 == 0)) && a.ldw % 8 == 0 && a.ldw >= a.K,
                   "usdm_gemv_engine: phase %d: K must be 4096, or (plain) an even multiple of 512 in (8192, 14336]", p);
    USDM_CHECK_ARG(nout % 512 == 0 && nout <= 16384 && (!glu || (a.N % 32 == 0 && !a.residual)), "usdm_gemv_engine: phase %d: outputs must be a multiple of 512 (<= 16384)", p);
    USDM_CHECK_ARG((int64_t)a.N * a.ldw * 2 < 0xFFFFFF00ll, "usdm_gemv_engine: phase %d: weight matrix exceeds the buffer range", p);
    USDM_CHECK_ARG(!a.part_val && !a.ban && !a.y32 && !a.x_delta && !a.x_out && !a.p2p_mode && !a.mrg_po,
                   "usdm_gemv_engine: phase %d: lm_head / f32 output / x_delta / p2p / merge modes are not chainable", p);
    USDM_CHECK_ARG(((uintptr_t)a.y16 % 4) == 0 && ((uintptr_t)a.x % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "usdm_gemv_engine: phase %d: alignment", p);
    USDM_CHECK_ARG(p == 0 || a.skip == pc->ph[0].skip, "usdm_gemv_engine: every phase must share phase 0's skip word");
    if (a.residual) {
      bool local = false;
      for (int q = 0; q < p; ++q)
        if (pc->ph[q].y16 == a.residual) {
          const int nq = pc->ph[q].act == USDM_ACT_SWIGLU ? pc->ph[q].N / 2 : pc->ph[q].N;
          USDM_CHECK_ARG(nq == nout, "usdm_gemv_engine: phase %d: an in-launch residual must have this phase's row count", p);
          local = true;
        }
      (void)local;
    }
  }
  usdm_gemv_chain_args c = *pc;
  for (int p = 0; p < c.nph; ++p) c.norm_nth[p] = usdm_gemv_threads(&c.ph[p]);
  hipLaunchKernelGGL(gemv_engine_kernel, dim3(NCU), dim3((NL + NC) * 64), 0, (hipStream_t)stream, c);
  USDM_LAUNCH_CHECK();
  return 0;
}
