// Batched decode for 5..16 sequences (round 4): the weight-streaming projection as a skinny GEMM on the matrix cores.
//
// Reference call site: the serving path hands vLLM an arbitrary request list (src/inference_vllm.py:109-125); here a decode step
// of up to 16 sequences streams every weight byte ONCE (14.3 GB per step, HBM-bound exactly like the batch-1 step) and does the
// 16 x more arithmetic on v_mfma_f32_16x16x32_bf16, where it is free: one MFMA (16 cycles of one SIMD) per KiB of weights against
// ~100 cycles of HBM time per KiB and CU.  The VALU form (llm_batch_k.hip, <= 4 sequences) pays 4 v_dot2 per 16 bytes and sequence.
//
// Layout of the product ("swap-AB"): the WEIGHTS are the A operand - a tile is 16 weight rows, lane (r = lane % 16, g = lane / 16)
// loads the 16 bytes W[row r][k0 + 8 g .. + 8) straight from HBM into the MFMA fragment layout (a wave instruction = 16 rows x 64
// contiguous bytes; two consecutive K chunks complete every 128-byte line; non-temporal) - and the activations [16 sequences][K] are
// the B operand (lane (b, g) holds x[b][k0 + 8 g .. + 8)).  D[row][sequence]: lane (b, g) ends with 4 consecutive output features
// of sequence b.  No LDS on the weight path, no transposes.
//
// Decomposition: ONE 8-wave workgroup per CU; the 8 waves split K (wave w owns K chunks [w cpw, (w + 1) cpw) of 32), so the
// activation slice a wave needs is 16 fragments that it HOLDS in registers for every tile (K <= 4096: all RMSNorm-fed projections and
// o_proj; the RMSNorm with HF's rounding points is applied to the held fragments in the prologue, under the first weight loads), or
// streams beside the weights (down_proj, K = 14336, one tile per workgroup).  Tiles (16 rows; 12 where that balances the 256 CUs
// better; SwiGLU: 8 gate + 8 up rows of the same 8 features) are dealt round-robin to the workgroups; a wave keeps the loads of two
// tiles (16 x 16 B per lane) in flight.  The 16 partial sums per output are added in wave order through LDS (deterministic), then
// the batch-1 kernels' epilogues: bf16 rounding points of HF, residual add, SwiGLU, ban-masked arg-max partials + logits.
//
// Per sequence the result is NOT bit-identical to usdm_gemv: the K partition and the MFMA's internal summation order differ (f32
// rounding of the accumulation; the RMSNorm sum of squares is partitioned differently as well).  Parity is against the oracle
// (tests/test_batch_gpu.py: near-tie rule), not against the batch-1 kernel.
#include "common.h"
#include "../../include/usdm_hip.h"
#include <type_traits>

namespace {
constexpr int MW = 8;     // waves per workgroup (two per SIMD: 256 registers each, spent on loads in flight)
constexpr int CH = 16;    // K chunks of 32 a wave can hold activations for (K <= MW * CH * 32 = 4096)
constexpr int MTG_DEF = 16;   // tiles per reduction group (LDS: MTG x MW waves x 1 KiB)
constexpr int MTG_TRL = 8;    // ... of the transposed-load variant (its LDS also holds 8 KiB of transposition buffer per wave)
#ifndef USDM_MFMA_NO_ASM
#define USDM_MFMA_NO_ASM 0   // 1: the compiler-scheduled stream everywhere (debugging)
#endif
constexpr int GAM_FLOATS = 4096;
constexpr int TAIL_BYTES = GAM_FLOATS * 4 + MW * 16 * 4 + 64 * 4 + 2 * MTG_DEF * 16 * 4;       // gam, ssum, tl, sv, si
constexpr int LDS_BYTES = MTG_DEF * MW * 1024 + TAIL_BYTES;
constexpr int TB_OFF = MTG_TRL * MW * 1024 + TAIL_BYTES;                                        // transposition buffers of the TRL variant
constexpr int LDS_BYTES_TRL = TB_OFF + MW * 8192;

#ifdef USDM_MFMA_TRACE
// debugging aid (tools/gemv_mfma_trace.py): per-workgroup phase timestamps of wave 0 (100 MHz wall clock) + hardware ids
__device__ unsigned long long g_mfma_trace[512 * 8];
#define TRM(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_mfma_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TRM(i) do { } while (0)
#endif

struct MfmaDev {
  usdm_gemv_batch_args ba;
  int ntiles, rt, cpw, nchunks, grid, nout;
  int ksplit, kwg;      // K split over workgroups: slices of ksplit_k elements, kwg workgroups per slice (1, grid: not split)
};
constexpr int KS_K = 2048;  // K per workgroup of the split form: 8 waves x 8 chunks, one 8-load unit per tile and wave
constexpr int KS_MAX = 8;   // slices (K <= 16384)

__device__ __forceinline__ f32x4 mfma16(u32x4 w, u32x4 x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// CPWT: K chunks per wave as a compile-time constant (16: K = 4096; 56: K = 14336), 0 = read it from the launch (any K).  With a
// constant every load of the stream loop is unconditional straight-line code; behind per-chunk branches the compiler's wait-count
// pass gives up at the joins and drains the whole ring (vmcnt(0)) in front of every MFMA - measured: 3.0 instead of ~5 TB/s.
// Hand-counted loads for the straight-line part of the HOLD stream (cdna_hip_programming.md 5.7, form (ii)): the compiler neither
// sees these loads nor waits for them; every consumer below waits for exactly its own load (in-order completion: "all but the 23
// youngest") inside the same asm statement that multiplies it.
__device__ __forceinline__ void ld_nt_asm(u32x4& dst, const u32x4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int WAIT, bool FIRST>
__device__ __forceinline__ void wait_mfma_asm(f32x4& acc, const u32x4& w, const u32x4& x) {
  if constexpr (FIRST)
    asm volatile("s_waitcnt vmcnt(%3)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x), "n"(WAIT));
  else
    asm volatile("s_waitcnt vmcnt(%3)\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x), "n"(WAIT));
}

// TRL ("transposed loads", K = 4096 only): the weights are loaded ROW-CONTIGUOUSLY - a load instruction reads 512 bytes of each of
// two rows, the pattern the batch-1 kernel streams at 5.9 TB/s - and re-cut into MFMA fragments through a private 8 KiB LDS buffer
// per wave (ds_write_b128 by rows, ds_read_b128 by fragments, XOR swizzle piece ^ row: conflict-free both ways).  The fragment-shaped
// loads of the other variants (16 rows x 64 bytes per instruction) measured 4.4 TB/s at best: 64-byte requests and a DRAM page
// visit per 64 - 128 bytes.  LDS traffic: 2 x 56 MB per CU and step against 56 MB of HBM traffic at a tenth of the LDS rate.
template <bool HOLD, int CPWT, bool TRL>
__global__ __launch_bounds__(MW * 64) void gemv_mfma_kernel(const MfmaDev d) {
  const usdm_gemv_args& a = d.ba.g;
  const int cpw = CPWT > 0 ? CPWT : d.cpw;
  constexpr int MTG = TRL ? MTG_TRL : MTG_DEF;             // tiles per reduction group
  constexpr int RED_BYTES = MTG * MW * 64 * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TRM(0);
#ifdef USDM_MFMA_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 512)
    g_mfma_trace[blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
#endif
  f32x4* red = (f32x4*)smem;                               // [MTG][MW][64] partial D fragments
  float* gam = (float*)(smem + RED_BYTES);                 // [K <= 4096] RMSNorm weight
  float* ssum = gam + GAM_FLOATS;                          // [MW][16] partial sums of squares
  int* tl = (int*)(ssum + MW * 16);                        // [MTG] tile ids of the group being reduced
  float* sv = (float*)(tl + 64);                           // lm_head: [MW][16] best value / index per reducing wave and sequence
  int* si = (int*)(sv + MTG_DEF * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int nb = d.ba.nb, K = a.K;
  const bool glu = a.act == USDM_ACT_SWIGLU;
  const bool lmh = a.part_val != nullptr;
  const bf16_t* Wb = (const bf16_t*)a.W;

  // ---- tiles of this workgroup: t = blockIdx.x + i * grid, i < ncand; lm_head: tiles whose 16 ids are all banned are not streamed
  // (K split over workgroups: workgroup = (slice ksi, member tile0 of the kwg workgroups that share the slice))
  const int ksi = d.ksplit > 1 ? (int)blockIdx.x % d.ksplit : 0;
  const int tile0 = d.ksplit > 1 ? (int)blockIdx.x / d.ksplit : (int)blockIdx.x, tstep = d.kwg;
  const int kofs = ksi * KS_K;
  const int ncand = tile0 < d.ntiles ? (d.ntiles - tile0 + tstep - 1) / tstep : 0;
  unsigned long long mask = ncand >= 64 ? ~0ull : ((1ull << ncand) - 1ull);
  if (lmh && a.ban) {
    bool act = false;
    if (lane < ncand) {
      const int t = tile0 + lane * tstep;
      if (t * 16 + 16 <= a.N && (((uintptr_t)a.ban) & 15) == 0) {      // one 16-byte load per tile (sixteen dependent byte loads cost ~10 us)
        const u32x4 bv = *(const u32x4*)(a.ban + t * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) act |= ((bv[e] & 0xff) == 0) | (((bv[e] >> 8) & 0xff) == 0) | (((bv[e] >> 16) & 0xff) == 0) | ((bv[e] >> 24) == 0);
      } else {
        for (int r = 0; r < 16 && t * 16 + r < a.N; ++r) act |= (a.ban[t * 16 + r] == 0);
      }
    }
    mask = __ballot(act);
    if (a.y32) {      // the logits of tiles that are not streamed: -inf for every sequence (what the sampling kernel expects of banned ids)
      for (int i = 0; i < ncand; ++i) {
        if ((mask >> i) & 1ull) continue;
        const int t = tile0 + i * tstep;
        for (int e = tid; e < 16 * nb; e += MW * 64) {
          const int n = t * 16 + (e & 15);
          if (n < a.N) a.y32[(int64_t)(e >> 4) * d.ba.y_bs + n] = -INFINITY;
        }
      }
    }
  }
  unsigned long long rem_ld = mask, rem_cp = mask;
  auto next_tile = [&](unsigned long long& m) -> int {
    if (!m) return -1;
    const int i = __builtin_ctzll(m);
    m &= m - 1;
    return tile0 + i * tstep;
  };
  // Per-lane base of this wave's K slice of tile t (A-row r16), and of the activation slice.  Rows / sequences that do not exist are
  // CLAMPED to existing ones instead of masked: they only feed output rows / columns that the epilogue never stores, and every load
  // below stays unconditional (uniform control flow, one 64-bit base + immediates per tile).
  const int kc0 = wave * cpw;                             // first K chunk of this wave (host: K %% (MW * 32) == 0)
  auto wbase = [&](int t) -> const u32x4* {
    int row;
    if (glu) {
      const int f = t < 0 ? 0 : min(t * 8 + (r16 & 7), d.nout - 1);
      row = (f >> 4) * 32 + (f & 15) + (r16 >= 8 ? 16 : 0);
    } else {
      row = t < 0 ? 0 : min(t * d.rt + r16, a.N - 1);
    }
    if (t < 0 && K >= 2048) {
      // No such tile: the stream loop's loads stay unconditional (uniform wait counts), but they must cost nothing - 64 bytes of the
      // activation vector per instruction (one request instead of sixteen), a different line for every wave so that no L2 channel
      // becomes a hot spot (weight row 0 for everyone did).  Row 0 of x holds K * 2 bytes; chunk offsets reach 1 KiB further.
      const int line = (int)((blockIdx.x * MW + wave) % (unsigned)(K / 32 - 16));
      return (const u32x4*)a.x + line * 4 + (lane & 3);
    }
    return (const u32x4*)(Wb + (int64_t)row * a.ldw + (int64_t)kc0 * 32 + 8 * g);
  };
  auto wload = [&](const u32x4* base, int c) -> u32x4 {
    return __builtin_nontemporal_load(base + c * 4);
  };
  const u32x4* xbase = (const u32x4*)((const bf16_t*)a.x + (int64_t)min(r16, nb - 1) * d.ba.x_bs + kofs + (int64_t)kc0 * 32 + 8 * g);
  auto xload = [&](int c) -> u32x4 { return xbase[c * 4]; };

  // ---- loads: the activation slice first (they return first: L2 hits), then the weights of the first two tiles
  u32x4 xf[HOLD ? CH : 1];
  float4 gv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if constexpr (HOLD) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (c < cpw) xf[c] = xload(c);
    if (a.norm_w) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if ((tid + q * MW * 64) * 4 < K) gv[q] = *(const float4*)(a.norm_w + (tid + q * MW * 64) * 4);
    }
  }
  // HOLD: a ring of RS = 24 weight fragments = the loads of one and a half tiles (24 KiB per wave, 192 KiB per CU in flight; 32 slots
  // next to the 16 held activation fragments do not fit 256 registers).  Item j = (tile j / 16, chunk j % 16) lives in slot j % 24 and is
  // refilled with item j + 24 as soon as it has been multiplied; the stream loop below is unrolled over three tiles so that slot and
  // chunk indices are constants.  Otherwise (STREAM): CH (weight, activation) pairs.
  constexpr int RS = 24;
  u32x4 ring[TRL ? 1 : (HOLD ? RS : CH)];
  u32x4 rx[HOLD ? 1 : CH];
  // ---- TRL: NR = 3 units of 8 row-contiguous loads (16 rows x 512 bytes = 8 chunks of this wave's K slice) = one and a half tiles in
  // flight (24 KiB per wave; four units next to the 16 held activation fragments spill)
  // (streamed activations: 16 loads per unit, two units = 32 KiB per wave; the K-split form holds only 8 activation fragments: 4 units)
  constexpr int NR = HOLD ? (CPWT == 8 ? 4 : 3) : 2;
  u32x4 U[TRL ? NR : 1][8];
  u32x4 X[(TRL && !HOLD) ? NR : 1][8];                     // streamed activations of the same units (K > 4096)
  const int lrow = lane >> 5, lp = lane & 31;              // row (of a pair) and 16-byte piece this lane loads
  auto unit_ptr = [&](int t, int sub, int i) -> const u32x4* {
    if (t < 0) return (const u32x4*)a.x + ((blockIdx.x * MW + wave) & 7) * 4 + (lane & 3) + i * 32;      // placeholder: 64 B of x (see wbase)
    const int r = 2 * i + lrow;                            // A-row of the tile
    int row;
    if (glu) {
      const int f = min(t * 8 + (r & 7), d.nout - 1);
      row = (f >> 4) * 32 + (f & 15) + (r >= 8 ? 16 : 0);
    } else {
      row = min(t * d.rt + r, a.N - 1);
    }
    return (const u32x4*)(Wb + (int64_t)row * a.ldw + kofs + (int64_t)wave * (cpw * 32) + sub * 256 + lp * 8);
  };
  int t0 = TRL ? -1 : next_tile(rem_ld), t1 = -1, t2 = -1;  // tile being multiplied, the next two (loads in flight / to be issued)
  const u32x4* wp0 = wbase(t0);
  const u32x4* wp1 = wp0;
  const u32x4* wp2 = wp0;
  constexpr bool ASM_STREAM = HOLD && CPWT == CH && !USDM_MFMA_NO_ASM && !TRL;     // hand-counted loads (see the stream loop)
  int ua = -1, ub = -1, uc = -1;                            // TRL: the tile being multiplied and the next two
  // TRL with streamed activations (K = 14336): units are numbered through the tiles, 7 per tile; (lt, ls) = the next unit to load
  int lt = -1, ls = 0;
  constexpr int UPT = CPWT / 8;                             // units per tile and wave
  auto ld_unit_x = [&](auto SLOT, int t, int sub) __attribute__((always_inline)) {
    constexpr int q = decltype(SLOT)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_nt_asm(U[q][i], unit_ptr(t, sub, i));
#pragma unroll
    for (int c = 0; c < 8; ++c) ld_nt_asm(X[q][c], xbase + (sub * 8 + c) * 4);
  };
  auto adv_unit = [&]() { if (++ls == UPT) { ls = 0; lt = next_tile(rem_ld); } };
  if constexpr (TRL && !HOLD) {
    lt = next_tile(rem_ld);
    ld_unit_x(std::integral_constant<int, 0>{}, lt, ls); adv_unit();
    ld_unit_x(std::integral_constant<int, 1>{}, lt, ls); adv_unit();
  }
  int tq[4] = {-1, -1, -1, -1};                             // K-split form: the tiles whose single unit sits in slots 0 .. 3
  if constexpr (TRL && HOLD && CPWT == 8) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tq[q] = next_tile(rem_ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) ld_nt_asm(U[q][i], unit_ptr(tq[q], 0, i));
    }
  }
  if constexpr (TRL && HOLD && CPWT != 8) {
    ua = next_tile(rem_ld); ub = next_tile(rem_ld); uc = next_tile(rem_ld);
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_nt_asm(U[0][i], unit_ptr(ua, 0, i));      // units 0, 1, 2 = (tile a, half 0), (a, 1), (b, 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_nt_asm(U[1][i], unit_ptr(ua, 1, i));
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_nt_asm(U[2][i], unit_ptr(ub, 0, i));
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (c < cpw && !TRL) {
      if constexpr (ASM_STREAM) ld_nt_asm(ring[c], wp0 + c * 4);
      else ring[c] = wload(wp0, c);
      if constexpr (!HOLD) rx[c] = xload(c);
    }
  }
  if constexpr (HOLD && !TRL) {
    t1 = next_tile(rem_ld);
    wp1 = wbase(t1);
#pragma unroll
    for (int c = 0; c < RS - CH; ++c) {
      if (c < cpw) {
        if constexpr (ASM_STREAM) ld_nt_asm(ring[CH + c], wp1 + c * 4);
        else ring[CH + c] = wload(wp1, c);
      }
    }
    t2 = next_tile(rem_ld);
    wp2 = wbase(t2);
  }

  TRM(1);
  // ---- RMSNorm of the held activation slice (HF: bf16(bf16(x * rstd) * weight)), under the weight loads
  if constexpr (HOLD) {
    if (a.norm_w) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (c < cpw) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bf2f(xf[c][e] & 0xffff), hi = bf2f(xf[c][e] >> 16);
            ss = fmaf(hi, hi, fmaf(lo, lo, ss));      // (explicit: the build contracts nothing; this prologue is ~700 vector instructions per wave)
          }
        }
      if (r16 >= nb) ss = 0.f;
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (g == 0) ssum[wave * 16 + r16] = ss;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if ((tid + q * MW * 64) * 4 < K) *(float4*)(gam + (tid + q * MW * 64) * 4) = gv[q];
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < MW; ++w) tot += ssum[w * 16 + r16];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (c < cpw) {
          const float4 g0 = *(const float4*)(gam + (kc0 + c) * 32 + 8 * g), g1 = *(const float4*)(gam + (kc0 + c) * 32 + 8 * g + 4);
          const float gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bf2f(xf[c][e] & 0xffff), hi = bf2f(xf[c][e] >> 16);
            xf[c][e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
          }
        }
      }
    }
  }

#ifdef USDM_MFMA_TRACE
  if constexpr (HOLD) asm volatile("s_nop 0" : "+v"(xf[0]), "+v"(xf[(CPWT > 0 ? CPWT : CH) - 1]));      // the held activations have landed / are normalised
#endif
  TRM(2);
  // ---- epilogue of one fully reduced tile: lane (sequence b = r16, row group g) holds output rows 4 g .. 4 g + 3
  float bestv = -INFINITY;
  int besti = 0x7fffffff;
  auto epilogue = [&](int t, f32x4 v) __attribute__((always_inline)) {
    const int b = r16;
    if (lmh) {
      float lv = -INFINITY;
      int li = 0x7fffffff;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = t * 16 + 4 * g + e;
        o[e] = -INFINITY;
        if (n < a.N && !(a.ban && a.ban[n])) {
          o[e] = round_bf(v[e]);
          if (o[e] > lv) { lv = o[e]; li = n; }
        }
      }
      if (a.y32 && b < nb) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t * 16 + 4 * g + e < a.N) a.y32[(int64_t)b * d.ba.y_bs + t * 16 + 4 * g + e] = o[e];
      }
#pragma unroll
      for (int s = 16; s <= 32; s <<= 1) {
        const float ov = __shfl_xor(lv, s, 64);
        const int oi = __shfl_xor(li, s, 64);
        if (ov > lv || (ov == lv && oi < li)) { lv = ov; li = oi; }
      }
      if (lv > bestv || (lv == bestv && li < besti)) { bestv = lv; besti = li; }
      return;
    }
    if (glu) {
      float up[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) up[e] = __shfl_xor(v[e], 32, 64);     // rows 8 .. 15 of the tile = the up rows of features 0 .. 7
      if (g < 2 && b < nb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o = t * 8 + 4 * g + e;
          if (o >= d.nout) continue;
          float r;
          if (a.round_bf16) {
            const float gt = round_bf(v[e]), u = round_bf(up[e]);
            r = round_bf(round_bf(gt / (1.0f + __expf(-gt))) * u);
          } else {
            r = (v[e] / (1.0f + __expf(-v[e]))) * up[e];
          }
          if (a.y16) ((bf16_t*)a.y16)[(int64_t)b * d.ba.y_bs + o] = f2bf(r);
          if (a.y32) a.y32[(int64_t)b * d.ba.y_bs + o] = r;
        }
      }
      return;
    }
    if (b >= nb) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = 4 * g + e, n = t * d.rt + rr;
      if (rr >= d.rt || n >= a.N) continue;
      float x = v[e];
      if (a.round_bf16) x = round_bf(x);
      if (a.residual) {
        x += bf2f(((const bf16_t*)a.residual)[(int64_t)b * d.ba.res_bs + n]);
        if (a.round_bf16) x = round_bf(x);
      }
      if (a.y16) ((bf16_t*)a.y16)[(int64_t)b * d.ba.y_bs + n] = f2bf(x);
      if (a.y32) a.y32[(int64_t)b * d.ba.y_bs + n] = x;
    }
  };
  // the tiles of a group are summed over the 16 waves in wave order (wave w < ng takes tile slot w) and finished
  int done = 0;
  // K split over workgroups: this workgroup's partial tile goes to ks_part[t][slice] as write-through stores (the merging workgroup
  // may sit on another XCD, whose L2 is not coherent with ours); once they are acknowledged, one relaxed increment of the tile's
  // counter; the wave that arrives last reads all slices back (agent-scope loads, all in flight together) and sums them in SLICE
  // order - the result does not depend on who was last.  No fences (an L2 write-back / invalidate per workgroup costs ~30 us).
  auto ks_merge = [&](int t, f32x4& s) __attribute__((always_inline)) -> bool {
    float* pp = d.ba.ks_part + (int64_t)t * d.ksplit * 256 + lane * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(pp + ksi * 256), "v"(s) : "memory");
    int old = 0;
    if (lane == 0) {
      old = __hip_atomic_fetch_add(d.ba.ks_cnt + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == d.ksplit - 1) __hip_atomic_store(d.ba.ks_cnt + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != d.ksplit - 1) return false;
    // all KS_MAX loads and their wait in ONE statement (slices that do not exist re-read the last one): nothing the compiler emits
    // can sit between a load and its wait
    f32x4 p[KS_MAX];
    const float* q[KS_MAX];
#pragma unroll
    for (int k = 0; k < KS_MAX; ++k) q[k] = pp + min(k, d.ksplit - 1) * 256;
    static_assert(KS_MAX == 8, "eight loads below");
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
        : "memory");
    s = p[0];
#pragma unroll
    for (int k = 1; k < KS_MAX; ++k)
      if (k < d.ksplit) s += p[k];
    return true;
  };
  auto flush_group = [&](bool more) __attribute__((always_inline)) {
    const int ng = ((done - 1) % MTG) + 1;                  // (the K-split form keeps all its <= MTG tiles for ONE flush after the stream)
    __syncthreads();
    TRM(5);
    for (int slot = wave; slot < ng; slot += MW) {
      f32x4 s = red[(slot * MW) * 64 + lane];
#pragma unroll
      for (int w = 1; w < MW; ++w) s += red[(slot * MW + w) * 64 + lane];
      if constexpr (TRL && HOLD && CPWT == 8) {
        if (d.ksplit > 1 && !ks_merge(tl[slot], s)) continue;
      }
      epilogue(tl[slot], s);
    }
    if (more) __syncthreads();
  };
  auto finish_tile = [&](int t, f32x4 acc) __attribute__((always_inline)) {
    const int slot = done % MTG;
    red[(slot * MW + wave) * 64 + lane] = acc;
    if (tid == 0) tl[slot] = t;
    ++done;
#ifdef USDM_MFMA_TRACE
    if (done == 1) TRM(3);
#endif
    if (done % MTG == 0) flush_group(rem_cp != 0ull);
  };

  // ---- stream
  if constexpr (TRL && !HOLD) {
    char* tb = smem + TB_OFF + wave * 8192;
    int ct = next_tile(rem_cp), cs = 0;                     // the unit being multiplied
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // one unit: its 16 loads have landed (the 32 younger ones stay in flight), rows -> LDS, fragments <- LDS, 8 MFMAs against the
    // unit's own activation fragments, then the slot is refilled with the unit three further on
    auto unit = [&](auto SLOT) __attribute__((always_inline)) -> bool {
      constexpr int q = decltype(SLOT)::value;
      if (ct < 0) return false;
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(U[q][0]), "+v"(U[q][1]), "+v"(U[q][2]), "+v"(U[q][3]), "+v"(U[q][4]), "+v"(U[q][5]), "+v"(U[q][6]), "+v"(U[q][7])
                   : "n"((NR - 1) * 16) : "memory");
      asm volatile("" : "+v"(X[q][0]), "+v"(X[q][1]), "+v"(X[q][2]), "+v"(X[q][3]), "+v"(X[q][4]), "+v"(X[q][5]), "+v"(X[q][6]), "+v"(X[q][7]));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 2 * i + lrow;
        *(u32x4*)(tb + r * 512 + ((lp ^ r) << 4)) = U[q][i];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 f = *(const u32x4*)(tb + r16 * 512 + (((4 * c + g) ^ r16) << 4));
        acc = mfma16(f, X[q][c], acc);
      }
      ld_unit_x(SLOT, lt, ls); adv_unit();
      if (++cs == UPT) {
        cs = 0;
        const int t = ct;
        ct = next_tile(rem_cp);
        finish_tile(t, acc);
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      return true;
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    while (unit(Q0{}) && unit(Q1{})) {}
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(U[0][0]), "v"(U[0][1]), "v"(U[0][2]), "v"(U[0][3]), "v"(U[0][4]), "v"(U[0][5]), "v"(U[0][6]), "v"(U[0][7]),
                 "v"(U[1][0]), "v"(U[1][1]), "v"(U[1][2]), "v"(U[1][3]), "v"(U[1][4]), "v"(U[1][5]), "v"(U[1][6]), "v"(U[1][7]) : "memory");
    asm volatile("" ::"v"(X[0][0]), "v"(X[0][1]), "v"(X[0][2]), "v"(X[0][3]), "v"(X[0][4]), "v"(X[0][5]), "v"(X[0][6]), "v"(X[0][7]),
                 "v"(X[1][0]), "v"(X[1][1]), "v"(X[1][2]), "v"(X[1][3]), "v"(X[1][4]), "v"(X[1][5]), "v"(X[1][6]), "v"(X[1][7]) : "memory");
  } else if constexpr (TRL && CPWT == 8) {
    // K split over workgroups: ONE unit per tile and wave (16 rows x 512 bytes of this wave's 256 K), four tiles in flight; tile n
    // sits in slot n % 4 and is refilled with tile n + 4 as soon as its rows are in LDS
    char* tb = smem + TB_OFF + wave * 8192;
    // A round of four units is straight-line code: a unit whose tile does not exist (only behind the last tile: tiles are dealt to
    // the slots in order) still waits, multiplies its placeholder lines and refills - only the hand-over of the result is conditional.
    auto unit = [&](auto SLOT) __attribute__((always_inline)) {
      constexpr int q = decltype(SLOT)::value;
      const int t = tq[q];
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(U[q][0]), "+v"(U[q][1]), "+v"(U[q][2]), "+v"(U[q][3]), "+v"(U[q][4]), "+v"(U[q][5]), "+v"(U[q][6]), "+v"(U[q][7])
                   : "n"((NR - 1) * 8) : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 2 * i + lrow;
        *(u32x4*)(tb + r * 512 + ((lp ^ r) << 4)) = U[q][i];
      }
      tq[q] = next_tile(rem_ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) ld_nt_asm(U[q][i], unit_ptr(tq[q], 0, i));
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 f = *(const u32x4*)(tb + r16 * 512 + (((4 * c + g) ^ r16) << 4));
        acc = mfma16(f, xf[c], acc);
      }
      if (t >= 0) {
        red[(done * MW + wave) * 64 + lane] = acc;           // at most MTG tiles per workgroup (host): reduced and merged after the stream,
        if (tid == 0) tl[done] = t;                          // when the load registers are free for the merge's partials
        ++done;
#ifdef USDM_MFMA_TRACE
        if (done == 1) TRM(3);
#endif
      }
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
    while (tq[0] >= 0) { unit(Q0{}); unit(Q1{}); unit(Q2{}); unit(Q3{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(U[0][0]), "v"(U[0][1]), "v"(U[0][2]), "v"(U[0][3]), "v"(U[0][4]), "v"(U[0][5]), "v"(U[0][6]), "v"(U[0][7]),
                 "v"(U[1][0]), "v"(U[1][1]), "v"(U[1][2]), "v"(U[1][3]), "v"(U[1][4]), "v"(U[1][5]), "v"(U[1][6]), "v"(U[1][7]),
                 "v"(U[2][0]), "v"(U[2][1]), "v"(U[2][2]), "v"(U[2][3]), "v"(U[2][4]), "v"(U[2][5]), "v"(U[2][6]), "v"(U[2][7]) : "memory");
    asm volatile("" ::"v"(U[3][0]), "v"(U[3][1]), "v"(U[3][2]), "v"(U[3][3]), "v"(U[3][4]), "v"(U[3][5]), "v"(U[3][6]), "v"(U[3][7]) : "memory");
  } else if constexpr (TRL) {
    char* tb = smem + TB_OFF + wave * 8192;                 // this wave's transposition buffer: [16 rows][32 pieces of 16 B], piece ^ row
    // one unit: wait for its 8 loads (the 24 younger ones stay in flight), rows -> LDS, refill the registers with the same unit of the
    // tile after next, fragments <- LDS, 8 MFMAs.  LDS operations of one wave execute in order: no wait between the writes and the
    // reads, nor between these reads and the next unit's writes.
    // Unit n = (tile n / 2, half n % 2) sits in slot n % 3 and is refilled with unit n + 3 = (next tile, half 1) behind half 0,
    // (tile after next, half 0) behind half 1.
    auto unit = [&](auto SLOT, auto SUB, f32x4& acc, int tnext) __attribute__((always_inline)) {
      constexpr int q = decltype(SLOT)::value, sub = decltype(SUB)::value;
      asm volatile("s_waitcnt vmcnt(%8)" : "+v"(U[q][0]), "+v"(U[q][1]), "+v"(U[q][2]), "+v"(U[q][3]), "+v"(U[q][4]), "+v"(U[q][5]), "+v"(U[q][6]), "+v"(U[q][7])
                   : "n"((NR - 1) * 8) : "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 2 * i + lrow;
        *(u32x4*)(tb + r * 512 + ((lp ^ r) << 4)) = U[q][i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) ld_nt_asm(U[q][i], unit_ptr(tnext, 1 - sub, i));
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 f = *(const u32x4*)(tb + r16 * 512 + (((4 * c + g) ^ r16) << 4));
        acc = mfma16(f, xf[sub * 8 + c], acc);
      }
    };
    using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>; using Q2 = std::integral_constant<int, 2>;
    // one tile at ring phase P (its halves sit in slots 2 P % 3 and (2 P + 1) % 3); false after the last tile
    auto ttile = [&](auto S0, auto S1) __attribute__((always_inline)) -> bool {
      if (ua < 0) return false;
      (void)next_tile(rem_cp);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      unit(S0, Q0{}, acc, ub);
      unit(S1, Q1{}, acc, uc);
      finish_tile(ua, acc);
      ua = ub; ub = uc; uc = next_tile(rem_ld);
      return true;
    };
    while (ttile(Q0{}, Q1{}) && ttile(Q2{}, Q0{}) && ttile(Q1{}, Q2{})) {}
    // drain: placeholder loads for tiles that do not exist may still be in flight into U; naming all 24 registers keeps the compiler
    // from reusing any of them before the wait has executed
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(U[0][0]), "v"(U[0][1]), "v"(U[0][2]), "v"(U[0][3]), "v"(U[0][4]), "v"(U[0][5]), "v"(U[0][6]), "v"(U[0][7]),
                 "v"(U[1][0]), "v"(U[1][1]), "v"(U[1][2]), "v"(U[1][3]), "v"(U[1][4]), "v"(U[1][5]), "v"(U[1][6]), "v"(U[1][7]),
                 "v"(U[2][0]), "v"(U[2][1]), "v"(U[2][2]), "v"(U[2][3]), "v"(U[2][4]), "v"(U[2][5]), "v"(U[2][6]), "v"(U[2][7]) : "memory");
  } else if constexpr (HOLD) {
    // one tile at ring phase P (its chunk c sits in slot (16 P + c) % 24); returns false after the last tile
    auto tile_phase = [&](auto P, auto STATIC) __attribute__((always_inline)) -> bool {
      constexpr int ph = decltype(P)::value;
      constexpr bool stat = decltype(STATIC)::value;
      if (t0 < 0) return false;
      (void)next_tile(rem_cp);                               // (rem_cp = the tiles after this one: finish_tile's "more")
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int sl = (CH * ph + c) % RS;
        if (c < cpw) {
          // item + 24 = (next tile, chunk c + 8) for c < 8, (tile after next, chunk c - 8) otherwise
          const u32x4* np = c < RS - CH ? wp1 + (c + (RS - CH)) * 4 : wp2 + (c - (RS - CH)) * 4;
          if constexpr (stat) {
            if (c == 0) wait_mfma_asm<RS - 1, true>(acc, ring[sl], xf[c]);
            else wait_mfma_asm<RS - 1, false>(acc, ring[sl], xf[c]);
            ld_nt_asm(ring[sl], np);
          } else {
            acc = mfma16(ring[sl], xf[c], acc);
            if ((c < RS - CH ? c + (RS - CH) : c - (RS - CH)) < cpw) ring[sl] = __builtin_nontemporal_load(np);
            __builtin_amdgcn_sched_barrier(0);               // keep (multiply, refill) pairs in program order
          }
        }
      }
      if constexpr (stat) {                                  // straight-line part: at most MTG tiles, no flush in between
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));     // the last MFMA's result before the compiler's own code reads it
        red[(done * MW + wave) * 64 + lane] = acc;
        if (tid == 0) tl[done] = t0;
        ++done;
      } else {
        finish_tile(t0, acc);
      }
      t0 = t1; t1 = t2; wp1 = wp2;
      t2 = next_tile(rem_ld);
      wp2 = wbase(t2);
      return true;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    if constexpr (ASM_STREAM) {
      // The first 12 tiles (every production launch has at most 11 per workgroup) as STRAIGHT-LINE code with hand-counted loads.
      // Left to the compiler, the wait-count pass puts vmcnt(7) in front of the first eight MFMAs of every tile - in the looped AND in
      // the straight-line form, with the (multiply, refill) pairs pinned in program order - although 23 younger loads are in flight:
      // the ring drains to a third every tile (measured: 3.0 TB/s against 5.9 for the batch-1 kernel).  The ring's first fill above is
      // hand-counted as well (a compiler-issued fill would be waited for with counts that ignore the asm loads: a full drain).
      static_assert(MTG >= 12 || TRL, "the straight-line part must fit one reduction group");
      using T = std::true_type;
      do {
        if (!tile_phase(I0{}, T{})) break; if (!tile_phase(I1{}, T{})) break; if (!tile_phase(I2{}, T{})) break;
        if (!tile_phase(I0{}, T{})) break; if (!tile_phase(I1{}, T{})) break; if (!tile_phase(I2{}, T{})) break;
        if (!tile_phase(I0{}, T{})) break; if (!tile_phase(I1{}, T{})) break; if (!tile_phase(I2{}, T{})) break;
        if (!tile_phase(I0{}, T{})) break; if (!tile_phase(I1{}, T{})) break; if (!tile_phase(I2{}, T{})) break;
      } while (false);
      // drain: loads for tiles that do not exist (row 0) may still be in flight into ring registers; naming all 24 as inputs keeps
      // the compiler from reusing any of them before the wait has executed
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(ring[0]), "v"(ring[1]), "v"(ring[2]), "v"(ring[3]), "v"(ring[4]), "v"(ring[5]), "v"(ring[6]), "v"(ring[7]),
                   "v"(ring[8]), "v"(ring[9]), "v"(ring[10]), "v"(ring[11]), "v"(ring[12]), "v"(ring[13]), "v"(ring[14]), "v"(ring[15]),
                   "v"(ring[16]), "v"(ring[17]), "v"(ring[18]), "v"(ring[19]), "v"(ring[20]), "v"(ring[21]), "v"(ring[22]), "v"(ring[23])
                   : "memory");
    }
    using F = std::false_type;
    while (tile_phase(I0{}, F{}) && tile_phase(I1{}, F{}) && tile_phase(I2{}, F{})) {}
  } else {
    // activations streamed beside the weights (K slices longer than CH chunks: down_proj): a ring of CH (weight, activation) pairs,
    // each refilled CH chunks ahead as soon as it has been multiplied
    int t = next_tile(rem_cp);
    const u32x4* wp = wp0;
    while (t >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = 0; c0 < cpw; c0 += CH) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (c0 + u < cpw) {
            acc = mfma16(ring[u], rx[u], acc);
            const int cn = c0 + u + CH;
            if (cn < cpw) { ring[u] = wload(wp, cn); rx[u] = xload(cn); }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      finish_tile(t, acc);
      t = next_tile(rem_cp);
      wp = wbase(t);
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (c < cpw) { ring[c] = wload(wp, c); rx[c] = xload(c); }
    }
  }
  TRM(4);
  if (done % MTG || (TRL && HOLD && CPWT == 8 && done)) flush_group(false);
#ifdef USDM_MFMA_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TRM(6);
#endif

  // ---- lm_head: per-workgroup arg-max partial of every sequence (ties -> lowest id); unused partial slots keep "no candidate"
  if (lmh) {
    __syncthreads();
    if (g == 0) { sv[wave * 16 + r16] = bestv; si[wave * 16 + r16] = besti; }
    __syncthreads();
    if (tid < nb) {
      float bv = sv[tid];
      int bi = si[tid];
      for (int w = 1; w < MW; ++w) {
        const float v = sv[w * 16 + tid];
        const int i = si[w * 16 + tid];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
      }
      a.part_val[(int64_t)tid * d.ba.part_bs + blockIdx.x] = bv;
      a.part_idx[(int64_t)tid * d.ba.part_bs + blockIdx.x] = bi == 0x7fffffff ? bi : bi + a.idx_offset;
      for (int j = blockIdx.x + d.grid; j < d.ba.part_bs; j += d.grid) {
        a.part_val[(int64_t)tid * d.ba.part_bs + j] = -INFINITY;
        a.part_idx[(int64_t)tid * d.ba.part_bs + j] = 0x7fffffff;
      }
    }
  }
}
}  // namespace

#ifdef USDM_MFMA_TRACE
extern "C" int usdm_dbg_mfma_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mfma_trace), sizeof(unsigned long long) * n);
}
#endif

extern "C" int64_t usdm_gemv_batch_ks_floats(int32_t N, int32_t K) {
  if (K <= MW * CH * 32 || K % KS_K != 0 || K / KS_K > KS_MAX || N <= 0) return 0;
  return (int64_t)cdiv(N, 16) * (K / KS_K) * 256;
}

// called by usdm_gemv_batch (llm_batch_k.hip) for 5..16 sequences, or when the caller forces the matrix-core form
int usdm_gemv_mfma_launch(const usdm_gemv_batch_args* pa, hipStream_t st) {
  const usdm_gemv_args& a = pa->g;
  USDM_CHECK_ARG(pa->nb >= 1 && pa->nb <= 16, "usdm_gemv_batch (matrix-core form): 1..16 sequences per step");
  USDM_CHECK_ARG(a.K % (MW * 32) == 0 && a.ldw % 8 == 0 && a.ldw >= a.K && pa->x_bs % 8 == 0,
                 "usdm_gemv_batch (matrix-core form): K %% %d, ldw %% 8, x stride %% 8", MW * 32);
  const bool glu = a.act == USDM_ACT_SWIGLU, lmh = a.part_val != nullptr;
  MfmaDev d;
  d.ba = *pa;
  d.nchunks = a.K / 32;
  d.cpw = cdiv(d.nchunks, MW);
  const bool hold = d.cpw <= CH;
  USDM_CHECK_ARG(!a.norm_w || (hold && a.K <= GAM_FLOATS && a.K % 4 == 0), "usdm_gemv_batch (matrix-core form): the fused RMSNorm needs K <= 4096");
  USDM_CHECK_ARG(!lmh || (a.part_idx && !glu), "usdm_gemv_batch: lm_head partial buffers");
  d.nout = glu ? a.N / 2 : a.N;
  d.rt = 16;
  d.ksplit = 1;
  const int64_t ksf = usdm_gemv_batch_ks_floats(a.N, a.K);
  if (ksf && pa->ks_part && pa->ks_cnt && !glu && !lmh && !a.norm_w && pa->form != 3 && pa->form != 5) {
    USDM_CHECK_ARG(pa->ks_part_floats >= ksf && (((uintptr_t)pa->ks_part) & 15) == 0, "usdm_gemv_batch: ks_part must hold %lld floats (16-byte aligned)", (long long)ksf);
    d.ksplit = a.K / KS_K;
    if (cdiv(cdiv(a.N, 16), min(256 / d.ksplit, cdiv(a.N, 16))) > MTG_TRL) d.ksplit = 1;      // one reduction group per workgroup
  }
  if (glu) {
    d.ntiles = cdiv(d.nout, 8);
  } else if (lmh) {
    d.ntiles = cdiv(a.N, 16);
  } else {
    // rows per tile: 16, or 12 / 8 where that shortens the longest workgroup (N = 6144: 384 tiles of 16 = 2 rounds of 16 rows, 512
    // tiles of 12 = 2 rounds of 12)
    int best = 1 << 30;
    for (int rt : {16, 12, 8}) {
      const int nt = cdiv(a.N, rt), per = cdiv(nt, nt < 256 ? nt : 256) * rt;
      if (per < best) { best = per; d.rt = rt; }
    }
    d.ntiles = cdiv(a.N, d.rt);
  }
  d.grid = d.ntiles < 256 ? d.ntiles : 256;
  d.kwg = d.grid;
  if (d.ksplit > 1) {      // 16-row tiles; 256 / ksplit workgroups per slice, each with every kwg-th tile
    d.rt = 16;
    d.ntiles = cdiv(a.N, 16);
    d.kwg = min(256 / d.ksplit, d.ntiles);
    d.grid = d.kwg * d.ksplit;
    d.cpw = KS_K / 32 / MW;
  }
  USDM_CHECK_ARG(cdiv(d.ntiles, d.kwg) <= 64, "usdm_gemv_batch (matrix-core form): N too large (more than 64 tiles per workgroup)");
  USDM_CHECK_ARG(!lmh || pa->part_bs >= d.grid, "usdm_gemv_batch: part_bs must hold one partial per workgroup (%d)", d.grid);
  // K = 4096 (every RMSNorm-fed projection and o_proj of the 7B): row-contiguous loads re-cut through LDS; form 3 forces the
  // fragment-shaped loads there (A/B: tools/gemv_mfma_bench.py)
  const bool trl = (hold ? d.cpw == 16 : d.cpw == 56) && pa->form != 3;
  if (d.ksplit > 1) {
    static bool ks_attr = false;
    if (!ks_attr) { (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_TRL); ks_attr = true; }
    hipLaunchKernelGGL((gemv_mfma_kernel<true, 8, true>), dim3(d.grid), dim3(MW * 64), LDS_BYTES_TRL, st, d);
    USDM_LAUNCH_CHECK();
    return 0;
  }
  void (*kfn)(const MfmaDev) = trl ? (hold ? gemv_mfma_kernel<true, 16, true> : gemv_mfma_kernel<false, 56, true>)
                             : hold ? (d.cpw == 16 ? gemv_mfma_kernel<true, 16, false> : gemv_mfma_kernel<true, 0, false>)
                                    : (d.cpw == 56 ? gemv_mfma_kernel<false, 56, false> : gemv_mfma_kernel<false, 0, false>);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_TRL);
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<false, 56, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_TRL);
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<false, 56, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemv_mfma_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(kfn, dim3(d.grid), dim3(MW * 64), trl ? LDS_BYTES_TRL : LDS_BYTES, st, d);
  USDM_LAUNCH_CHECK();
  return 0;
}
