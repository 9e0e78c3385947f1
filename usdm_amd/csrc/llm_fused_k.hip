// Decode step of the 7B: context-split attention + the merge of its partials + the o_proj GEMV in ONE launch (round 3).
//
// Separate launches (usdm_attn_decode: split kernel 6.6 us + combine 4.6 us, then usdm_gemv o_proj 7.7 us) are three latency chains
// in a row, and the o_proj launch spends most of its time waiting for its first weight bytes (ring issue 1.8 us, arrival 4.6 us,
// K loop 0.9 us: profiles/r01_decode_ablation.txt 6).  Here every workgroup requests ITS o_proj weight rows first - the whole rows:
// 2 rows x 8 KB per wave, 33.5 MB over the chip - and the attention runs while they travel:
//   grid  = Hkv * NS workgroups (7B: 8 x 32 = 256 = one per CU, all co-resident), 512 threads = 8 waves
//   (1)   waves 4..7: 32 non-temporal 16-B loads per lane = four whole weight rows per wave (16 rows per workgroup)
//   (2)   waves 0..3 of workgroup (kh, sp): the attention partial of kv head kh over split sp - the code of attn_decode_kernel -
//         published as 8-byte granules {tag, f32}
//   (3)   workgroups 0 .. Hq-1: gather the NS partials of head blockIdx.x (bounded re-reads), combine them with the arithmetic of
//         attn_combine_kernel, publish the 128 outputs as granules {tag, 2 x bf16}
//   (4)   all: gather the K/2 output granules into LDS, multiply the rows held in registers, residual epilogue of usdm_gemv
// tag = *epoch + 1 with *epoch advanced once per decode step by usdm_epoch_inc (monotonic: a replayed hipGraph, a new sequence and a
// stale buffer can never agree on a tag); buffers are per layer.  Nothing waits without a bound (timeout_ms of the 100 MHz clock):
// on expiry the workgroup ORs 1 into *err and carries on with zeros.  No workgroup waits for a workgroup that is not resident: the
// grid is at most one workgroup per CU (checked by the launcher against the device).
// Bit-identical to the separate launches: same per-split arithmetic, same combine order, same K partition and summation order
// of the GEMV (lane l of a wave owns elements 8 (64 i + l) .. +7 of a row in both forms).
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
constexpr int FA_KMAX = 512;   // max keys per split (as attn_decode_kernel)
constexpr int FA_PG = 130;     // granules per (head, split): 128 outputs, max, sum

__device__ __forceinline__ float fdot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = w[i], b = x[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  return acc;
}
__device__ __forceinline__ void frope_pair(float x1, float x2, float c, float s, float& o1, float& o2) {
  o1 = round_bf(round_bf(x1 * c) + round_bf(-x2 * s));
  o2 = round_bf(round_bf(x2 * c) + round_bf(x1 * s));
}
__device__ __forceinline__ unsigned long long gload(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gstore(unsigned long long* p, unsigned tag, unsigned payload) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// N granules per thread: all requested at once, the missing ones re-read until their tag arrives or the deadline passes
template <int N, typename PF>
__device__ __forceinline__ void ggather(unsigned long long (&g)[N], int n, PF&& ptr, unsigned tag, unsigned long long t0, unsigned long long tmo, bool& late) {
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = i < n ? gload(ptr(i)) : ((unsigned long long)tag << 32);
  unsigned spins = 0;
  for (;;) {
    bool all = true;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((unsigned)(g[i] >> 32) != tag) {
        g[i] = gload(ptr(i));
        all = all && ((unsigned)(g[i] >> 32) == tag);
      }
    if (all || late) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 31) == 31 && wall_clock64() - t0 > tmo) late = true;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
    if ((unsigned)(g[i] >> 32) != tag) g[i] = 0ull;          // timed out: zeros (the error word is raised by the caller)
}

#ifdef USDM_FAO_TRACE
// debugging aid (tools/fao_trace.py): 100 MHz wall-clock stamps per workgroup, wave 0 lane 0 (slots 0..7) and wave 4 lane 0 (8..15)
__device__ unsigned long long g_fao_trace[256 * 16];
#define FTR(i) do { if (lane == 0 && (wave8 == 0 || wave8 == 4) && blockIdx.x < 256) g_fao_trace[blockIdx.x * 16 + (wave8 ? 8 : 0) + (i)] = wall_clock64(); } while (0)
#else
#define FTR(i) do { } while (0)
#endif

template <int G>
__global__ __launch_bounds__(512) void attn_oproj_kernel(const usdm_attn_oproj_args A) {
  const usdm_attn_decode_args& a = A.attn;
  const usdm_gemv_args& gv = A.gemv;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;                     // [Kpad] bf16: the attention output, LDS copy
  __shared__ float qs[G][128];
  __shared__ float knew[128], vnew[128];
  __shared__ float sc[G][FA_KMAX];
  __shared__ float red[8][G][128];
  __shared__ float lsum[G], lmax[G];
  __shared__ float cw[65];
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  // Roles.  vmcnt retires IN ORDER within a wave: a wave that has 33.5 MB / 2048 of weight loads in flight cannot consume a later
  // (or, through the compiler's conservative wait counts, even an earlier) small load until they have all landed.  So the weight
  // rows are requested by waves 4..7 (4 rows each, 32 loads per lane) and the attention is computed by waves 0..3, whose own
  // loads are all that is in their queues.  Every barrier below is executed by all 512 threads at the same place.
  const bool loader = __builtin_amdgcn_readfirstlane(wave8) >= 4;
  const int wave = wave8 & 3;
  const int skipv = a.skip ? *a.skip : 0;
  const unsigned tag = *A.epoch + 1u;
  const int kh = (int)blockIdx.x % a.Hkv, sp = (int)blockIdx.x / a.Hkv, NS = a.NS;
  const int pos = a.pos[0];
  if ((unsigned)pos >= (unsigned)a.ctx_max) return;   // (every workgroup takes this exit together)
  const unsigned long long tmo = (unsigned long long)(A.timeout_ms > 0 ? A.timeout_ms : 200) * 100000ull;
  const unsigned long long t0 = wall_clock64();
  bool late = false;
  FTR(0);

  const int K = gv.K, Kpad = (K + 511) & ~511, nit = Kpad >> 9;      // nit <= 8: a row is at most 8 x 16 B per lane
  const int ob = (int)blockIdx.x * 16 + wave * 4;                    // first of the 4 output rows of a loader wave
  u32x4 ring[4][8];
  // attention-side registers (compute waves)
  const int ctx = pos + 1;
  const int lo = (a.window > 0 && ctx > a.window) ? ctx - a.window : 0;
  const int chunk = (ctx - lo + NS - 1) / NS;
  const int k0 = lo + sp * chunk, k1 = min(ctx, k0 + chunk);
  const int nk = max(0, k1 - k0);
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  bf16_t* kcache_b = (bf16_t*)a.kcache;
  bf16_t* vcache_b = (bf16_t*)a.vcache;
  const bf16_t* Kc = kcache_b + (int64_t)kh * a.ctx_max * 128;
  const bf16_t* Vc = vcache_b + (int64_t)kh * a.ctx_max * 128;
  constexpr int SW = 4, PW = 4;
  const int j8 = lane & 7, gk = (wave << 3) + (lane >> 3);
  const int d4 = (tid & 31) * 4, kl = (tid & 255) >> 5;
  u32x4 r0[SW], r1[SW];
  u32x2 rv[PW];
  if (loader) {
    // ---- (1) the 16 o_proj weight rows of this workgroup, entire rows; every load is issued (rows / pieces past the end are
    // redirected to a valid address and multiplied by the zero padding of x) so that the count in flight is a constant
    const bool tail_ok = ((nit - 1) << 9) + lane * 8 < K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = ob + j < gv.N ? ob + j : gv.N - 1;
      const u32x4* wr = (const u32x4*)((const bf16_t*)gv.W + (int64_t)r * gv.ldw);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = u < nit && !(u == nit - 1 && !tail_ok);
        ring[j][u] = __builtin_nontemporal_load(in ? wr + u * 64 + lane : wr);
      }
    }
  } else if (nk > 0) {
#pragma unroll
    for (int w = 0; w < SW; ++w) {
      const int kk = min(32 * w + gk, nk - 1);
      const bf16_t* kp = Kc + (int64_t)(k0 + kk) * 128 + j8 * 16;
      r0[w] = *(const u32x4*)kp;
      r1[w] = *(const u32x4*)(kp + 8);
    }
#pragma unroll
    for (int w = 0; w < PW; ++w) {
      const int kk = min(8 * w + kl, nk - 1);
      rv[w] = *(const u32x2*)(Vc + (int64_t)(k0 + kk) * 128 + d4);
    }
  }
  FTR(1);
  if (skipv) return;   // the sequence has ended (usdm_decode_state.done): every workgroup leaves here, nothing is appended

  // ---- (2) attention partial of (kh, sp) on waves 0..3: the code of attn_decode_kernel
  if (!loader) {
    for (int i = tid; i < (G + 1) * 64; i += 256) {
      const int hsel = i >> 6, d = i & 63;
      const float c = bf2f(a.cos[(int64_t)pos * 64 + d]), sn = bf2f(a.sin[(int64_t)pos * 64 + d]);
      const bf16_t* src = hsel < G ? qkv + (kh * G + hsel) * 128 : qkv + (a.Hq + kh) * 128;
      float o1, o2;
      frope_pair(bf2f(src[d]), bf2f(src[d + 64]), c, sn, o1, o2);
      if (hsel < G) { qs[hsel][d] = o1; qs[hsel][d + 64] = o2; }
      else { knew[d] = o1; knew[d + 64] = o2; }
    }
    if (tid < 128) vnew[tid] = bf2f(qkv[(a.Hq + a.Hkv + kh) * 128 + tid]);
  }
  __syncthreads();
  if (!loader) {
    if (sp == 0 && tid < 128) {  // designated writer of the new cache row
      kcache_b[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(knew[tid]);
      vcache_b[((int64_t)kh * a.ctx_max + pos) * 128 + tid] = f2bf(vnew[tid]);
    }
    float qr[G][16];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 16; ++e) qr[h][e] = qs[h][j8 * 16 + e];
    for (int base = 0; base < nk; base += 32 * SW) {
      if (base > 0) {
#pragma unroll
        for (int w = 0; w < SW; ++w) {
          const int kk = min(base + 32 * w + gk, nk - 1);
          const bf16_t* kp = Kc + (int64_t)(k0 + kk) * 128 + j8 * 16;
          r0[w] = *(const u32x4*)kp;
          r1[w] = *(const u32x4*)(kp + 8);
        }
      }
#pragma unroll
      for (int w = 0; w < SW; ++w) {
        const int kk = base + 32 * w + gk;
        if (kk >= nk) continue;
        const bool is_new = (k0 + kk) == pos;
        float kv[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kv[2 * e] = bf2f(r0[w][e] & 0xffff); kv[2 * e + 1] = bf2f(r0[w][e] >> 16);
          kv[8 + 2 * e] = bf2f(r1[w][e] & 0xffff); kv[8 + 2 * e + 1] = bf2f(r1[w][e] >> 16);
        }
        if (is_new) {
#pragma unroll
          for (int e = 0; e < 16; ++e) kv[e] = knew[j8 * 16 + e];
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
          float sdot = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) sdot = fmaf(qr[h][e], kv[e], sdot);
          sdot += __shfl_xor(sdot, 1, 64); sdot += __shfl_xor(sdot, 2, 64); sdot += __shfl_xor(sdot, 4, 64);
          if (j8 == 0) sc[h][kk] = sdot * a.scale;
        }
      }
    }
  }
  __syncthreads();
  if (!loader) {
    for (int h = wave; h < G; h += 4) {   // softmax statistics per head (wave h <-> head h)
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
  }
  __syncthreads();
  if (!loader) {
    float acc[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[h][e] = 0.f;
    for (int base = 0; base < nk; base += 8 * PW) {
      if (base > 0) {
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
    }
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[kl][h][d4 + e] = acc[h][e];
  }
  __syncthreads();
  if (!loader) {
    for (int i = tid; i < G * 128; i += 256) {
      const int h = i >> 7, d = i & 127;
      float s = 0.f;
#pragma unroll
      for (int kl2 = 0; kl2 < 8; ++kl2) s += red[kl2][h][d];
      const int hq = kh * G + h;
      unsigned long long* pg = A.part_gran + ((int64_t)hq * NS + sp) * FA_PG;
      gstore(pg + d, tag, __float_as_uint(s));
      if (d == 0) {
        gstore(pg + 128, tag, __float_as_uint(nk > 0 ? lmax[h] : -1e30f));
        gstore(pg + 129, tag, __float_as_uint(nk > 0 ? lsum[h] : 0.f));
      }
    }
  }

  FTR(2);
  // ---- (3) workgroups 0 .. Hq-1 combine one head each (attn_combine_kernel's arithmetic on threads 0..127)
  if ((int)blockIdx.x < a.Hq) {
    const int hq = blockIdx.x, d = tid;
    const unsigned long long* pg = A.part_gran + (int64_t)hq * NS * FA_PG;
    unsigned long long gp[32];
    if (d < 128) ggather<32>(gp, NS, [&](int s) { return pg + (int64_t)s * FA_PG + d; }, tag, t0, tmo, late);
    if (d < 64) {
      unsigned long long gm[2];
      const int sd = d < NS ? d : 0;
      ggather<2>(gm, 2, [&](int i) { return pg + (int64_t)sd * FA_PG + 128 + i; }, tag, t0, tmo, late);
      const float mv = d < NS ? __uint_as_float((unsigned)gm[0]) : -1e30f;
      const float m = wave_max(mv);
      const float e = d < NS ? __expf(mv - m) : 0.f;
      const float l = wave_sum(d < NS ? __uint_as_float((unsigned)gm[1]) * e : 0.f);
      cw[d] = e;
      if (d == 0) cw[64] = 1.0f / l;
    }
    __syncthreads();
    if (d < 128) {
      float o = 0.f;
      int s = 0;
#pragma unroll
      for (int s4 = 0; s4 < 32; s4 += 4) {
        if (s4 + 4 <= NS) {
          o += (__uint_as_float((unsigned)gp[s4]) * cw[s4] + __uint_as_float((unsigned)gp[s4 + 1]) * cw[s4 + 1]) +
               (__uint_as_float((unsigned)gp[s4 + 2]) * cw[s4 + 2] + __uint_as_float((unsigned)gp[s4 + 3]) * cw[s4 + 3]);
          s = s4 + 4;
        }
      }
#pragma unroll
      for (int s1 = 0; s1 < 32; ++s1)
        if (s1 >= s && s1 < NS) o += __uint_as_float((unsigned)gp[s1]) * cw[s1];
      const unsigned v = f2bf(o * cw[64]);
      const unsigned hi = __shfl_down(v, 1, 64);
      if (!(d & 1)) gstore(A.x_gran + hq * 64 + (d >> 1), tag, v | (hi << 16));
    }
  }

  FTR(3);
  // ---- (4) waves 0..3 gather the attention output (K/2 granules) into LDS; waves 4..7 multiply their rows; usdm_gemv's epilogue
  if (!loader) {
    const int ng = K >> 1;
    for (int base = 0; base < (Kpad >> 1); base += 256 * 8) {
      unsigned long long gx[8];
      ggather<8>(gx, 8, [&](int i) { const int q = base + i * 256 + tid; return A.x_gran + (q < ng ? q : 0); }, tag, t0, tmo, late);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = base + i * 256 + tid;
        if (q < (Kpad >> 1)) *(unsigned*)(xs + 2 * q) = q < ng ? (unsigned)gx[i] : 0u;
      }
    }
  }
  if (late && A.err) atomicOr((int*)A.err, 1);
  FTR(4);
  __syncthreads();
  FTR(5);
  if (!loader) return;
  float acc2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < nit) {
      const u32x4 xv = *(const u32x4*)(xs + (u * 64 + lane) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[j] = fdot8(ring[j][u], xv, acc2[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc2[j] = wave_sum(acc2[j]);
  FTR(6);
  if (lane != 0) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = ob + j;
    if (n >= gv.N) continue;
    float v = acc2[j];
    if (gv.round_bf16) v = round_bf(v);
    if (gv.residual) {
      v += bf2f(((const bf16_t*)gv.residual)[n]);
      if (gv.round_bf16) v = round_bf(v);
    }
    if (gv.y16) ((bf16_t*)gv.y16)[n] = f2bf(v);
    if (gv.y32) gv.y32[n] = v;
  }
}

__global__ void epoch_inc_kernel(unsigned* epoch) { *epoch += 1u; }
}  // namespace

#ifdef USDM_FAO_TRACE
extern "C" int usdm_dbg_fao_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fao_trace), sizeof(unsigned long long) * n);
}
#endif

extern "C" int usdm_epoch_inc(uint32_t* epoch, usdm_stream_t stream) {
  USDM_CHECK_ARG(epoch != nullptr, "usdm_epoch_inc: null");
  hipLaunchKernelGGL(epoch_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, epoch);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_attn_oproj(const usdm_attn_oproj_args* pA, usdm_stream_t stream) {
  USDM_CHECK_ARG(pA != nullptr, "usdm_attn_oproj: null args");
  const usdm_attn_oproj_args& A = *pA;
  const usdm_attn_decode_args& a = A.attn;
  const usdm_gemv_args& g = A.gemv;
  USDM_CHECK_ARG(a.qkv && a.pos && a.kcache && a.vcache && a.cos && a.sin && A.part_gran && A.x_gran && A.epoch, "usdm_attn_oproj: null pointers");
  USDM_CHECK_ARG(a.Hkv > 0 && a.Hq % a.Hkv == 0 && a.NS > 1 && a.NS <= 32 && a.batch <= 1 && !a.counters, "usdm_attn_oproj: heads / 2 <= NS <= 32 / single sequence");
  const int G = a.Hq / a.Hkv;
  const int span = (a.window > 0 && a.window < a.ctx_max) ? a.window : a.ctx_max;
  USDM_CHECK_ARG(cdiv(span, a.NS) <= FA_KMAX, "usdm_attn_oproj: visible keys / NS exceeds %d keys per split", FA_KMAX);
  USDM_CHECK_ARG(g.W && g.y16 && g.K == a.Hq * 128 && g.K <= 4096 && g.K % 8 == 0 && g.ldw >= g.K && g.ldw % 8 == 0 && g.act == USDM_ACT_NONE && !g.norm_w &&
                     !g.x_delta && !g.part_val && !g.p2p_mode && !g.mrg_po && !g.cmb_gran,
                 "usdm_attn_oproj: the projection must be plain, K = Hq * 128 <= 4096");
  USDM_CHECK_ARG(g.N == 16 * a.Hkv * a.NS, "usdm_attn_oproj: N (%d) must be 16 * Hkv * NS (%d): one workgroup = one (kv head, split) = 16 rows", g.N, 16 * a.Hkv * a.NS);
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
  }
  USDM_CHECK_ARG(n_cu > 0 && a.Hkv * a.NS <= n_cu, "usdm_attn_oproj: %d workgroups must be co-resident, the device has %d CUs", a.Hkv * a.NS, n_cu);
  const size_t lds = (size_t)((g.K + 511) & ~511) * 2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.Hkv * a.NS), block(512);
  if (G == 4) hipLaunchKernelGGL(attn_oproj_kernel<4>, grid, block, lds, st, A);
  else if (G == 2) hipLaunchKernelGGL(attn_oproj_kernel<2>, grid, block, lds, st, A);
  else if (G == 1) hipLaunchKernelGGL(attn_oproj_kernel<1>, grid, block, lds, st, A);
  else { usdm_set_error("usdm_attn_oproj: group size %d unsupported (1,2,4)", G); return 2; }
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_attn_oproj_args(void) { return (int)sizeof(usdm_attn_oproj_args); }
