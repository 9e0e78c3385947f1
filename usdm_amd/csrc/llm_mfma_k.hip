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
// Decomposition: ONE 16-wave workgroup per CU; the 16 waves split K (wave w owns K chunks [w cpw, (w + 1) cpw) of 32), so the
// activation slice a wave needs is 8 fragments that it HOLDS in registers for every tile (K <= 4096: all RMSNorm-fed projections and
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

namespace {
constexpr int MW = 16;    // waves per workgroup
constexpr int MTG = 8;    // tiles per reduction group (LDS: MTG x 16 waves x 1 KiB)
constexpr int RED_BYTES = MTG * MW * 64 * 16;
constexpr int GAM_FLOATS = 4096;
constexpr int LDS_BYTES = RED_BYTES + GAM_FLOATS * 4 + MW * 16 * 4 + 64 * 4 + 2 * 8 * 16 * 4;

struct MfmaDev {
  usdm_gemv_batch_args ba;
  int ntiles, rt, cpw, nchunks, grid, nout;
};

__device__ __forceinline__ f32x4 mfma16(u32x4 w, u32x4 x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

template <bool HOLD>
__global__ __launch_bounds__(MW * 64) void gemv_mfma_kernel(const MfmaDev d) {
  const usdm_gemv_args& a = d.ba.g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                               // [MTG][MW][64] partial D fragments
  float* gam = (float*)(smem + RED_BYTES);                 // [K <= 4096] RMSNorm weight
  float* ssum = gam + GAM_FLOATS;                          // [MW][16] partial sums of squares
  int* tl = (int*)(ssum + MW * 16);                        // [MTG] tile ids of the group being reduced
  float* sv = (float*)(tl + 64);                           // lm_head: [8][16] best value / index per reducing wave and sequence
  int* si = (int*)(sv + 8 * 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  const int nb = d.ba.nb, K = a.K;
  const bool glu = a.act == USDM_ACT_SWIGLU;
  const bool lmh = a.part_val != nullptr;
  const bf16_t* Wb = (const bf16_t*)a.W;

  // ---- tiles of this workgroup: t = blockIdx.x + i * grid, i < ncand; lm_head: tiles whose 16 ids are all banned are not streamed
  const int ncand = (d.ntiles - (int)blockIdx.x + d.grid - 1) / d.grid;
  unsigned long long mask = ncand >= 64 ? ~0ull : ((1ull << ncand) - 1ull);
  if (lmh && a.ban) {
    bool act = false;
    if (lane < ncand) {
      const int t = blockIdx.x + lane * d.grid;
      for (int r = 0; r < 16 && t * 16 + r < a.N; ++r) act |= (a.ban[t * 16 + r] == 0);
    }
    mask = __ballot(act);
  }
  unsigned long long rem_ld = mask, rem_cp = mask;
  auto next_tile = [&](unsigned long long& m) -> int {
    if (!m) return -1;
    const int i = __builtin_ctzll(m);
    m &= m - 1;
    return (int)blockIdx.x + i * d.grid;
  };
  // weight row of A-row r16 of tile t (-1: none), and the 16-byte fragment of K chunk c of this wave's slice
  auto wrow = [&](int t) -> int {
    if (t < 0) return -1;
    if (glu) {
      const int f = t * 8 + (r16 & 7);
      return f < d.nout ? (f >> 4) * 32 + (f & 15) + (r16 >= 8 ? 16 : 0) : -1;
    }
    const int n = t * d.rt + r16;
    return (r16 < d.rt && n < a.N) ? n : -1;
  };
  const int kc0 = wave * d.cpw;                             // first K chunk of this wave
  auto wload = [&](int row, int c) -> u32x4 {
    if (row < 0 || kc0 + c >= d.nchunks) return u32x4{0u, 0u, 0u, 0u};
    return __builtin_nontemporal_load((const u32x4*)(Wb + (int64_t)row * a.ldw + (int64_t)(kc0 + c) * 32 + 8 * g));
  };
  auto xload = [&](int c) -> u32x4 {
    if (r16 >= nb || kc0 + c >= d.nchunks) return u32x4{0u, 0u, 0u, 0u};
    return *(const u32x4*)((const bf16_t*)a.x + (int64_t)r16 * d.ba.x_bs + (int64_t)(kc0 + c) * 32 + 8 * g);
  };

  // ---- loads: the activation slice first (they return first: L2 hits), then the weights of the first two tiles
  u32x4 xf[8];
  float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (HOLD) {
#pragma unroll
    for (int c = 0; c < 8; ++c) xf[c] = xload(c);
    if (a.norm_w && tid * 4 < K) gv = *(const float4*)(a.norm_w + tid * 4);
  }
  u32x4 ring[8];                                            // the 8 weight fragments of one tile; slot c is refilled with the NEXT
  u32x4 rx[HOLD ? 1 : 8];                                   // tile's chunk c as soon as it has been multiplied (8 KiB per wave in flight)
  const int tA = next_tile(rem_ld), rowA = wrow(tA);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    ring[c] = wload(rowA, c);
    if constexpr (!HOLD) rx[c] = xload(c);
  }

  // ---- RMSNorm of the held activation slice (HF: bf16(bf16(x * rstd) * weight)), under the weight loads
  if constexpr (HOLD) {
    if (a.norm_w) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf2f(xf[c][e] & 0xffff), hi = bf2f(xf[c][e] >> 16);
          ss += lo * lo + hi * hi;
        }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (g == 0) ssum[wave * 16 + r16] = ss;
      if (tid * 4 < K) *(float4*)(gam + tid * 4) = gv;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < MW; ++w) tot += ssum[w * 16 + r16];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (kc0 + c < d.nchunks) {
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

  // ---- epilogue of one fully reduced tile: lane (sequence b = r16, row group g) holds output rows 4 g .. 4 g + 3
  float bestv = -INFINITY;
  int besti = 0x7fffffff;
  auto epilogue = [&](int t, f32x4 v) {
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
  auto flush_group = [&](bool more) {
    const int ng = ((done - 1) % MTG) + 1;
    __syncthreads();
    if (wave < ng) {
      f32x4 s = red[(wave * MW) * 64 + lane];
#pragma unroll
      for (int w = 1; w < MW; ++w) s += red[(wave * MW + w) * 64 + lane];
      epilogue(tl[wave], s);
    }
    if (more) __syncthreads();
  };
  auto finish_tile = [&](int t, f32x4 acc) {
    const int slot = done % MTG;
    red[(slot * MW + wave) * 64 + lane] = acc;
    if (tid == 0) tl[slot] = t;
    ++done;
    if (done % MTG == 0) flush_group(rem_cp != 0ull);
  };

  // ---- stream
  if constexpr (HOLD) {
    while (true) {
      const int t = next_tile(rem_cp);
      if (t < 0) break;
      const int tn = next_tile(rem_ld), rown = wrow(tn);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c) { acc = mfma16(ring[c], xf[c], acc); if (tn >= 0) ring[c] = wload(rown, c); }
      finish_tile(t, acc);
    }
  } else {
    // activations streamed beside the weights (K slices longer than 8 chunks: down_proj): a ring of 8 (weight, activation) pairs,
    // each refilled 8 chunks ahead as soon as it has been multiplied
    int t = next_tile(rem_cp);
    int row = rowA;
    while (t >= 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = 0; c0 < d.cpw; c0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (c0 + u < d.cpw) {
            acc = mfma16(ring[u], rx[u], acc);
            const int cn = c0 + u + 8;
            if (cn < d.cpw) { ring[u] = wload(row, cn); rx[u] = xload(cn); }
          }
        }
      }
      finish_tile(t, acc);
      t = next_tile(rem_cp);
      row = wrow(t);
      if (t >= 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { ring[c] = wload(row, c); rx[c] = xload(c); }
      }
    }
  }
  if (done % MTG) flush_group(false);

  // ---- lm_head: per-workgroup arg-max partial of every sequence (ties -> lowest id); unused partial slots keep "no candidate"
  if (lmh) {
    __syncthreads();
    if (wave < 8 && g == 0) { sv[wave * 16 + r16] = bestv; si[wave * 16 + r16] = besti; }
    __syncthreads();
    if (tid < nb) {
      float bv = sv[tid];
      int bi = si[tid];
      for (int w = 1; w < 8; ++w) {
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

// called by usdm_gemv_batch (llm_batch_k.hip) for 5..16 sequences, or when the caller forces the matrix-core form
int usdm_gemv_mfma_launch(const usdm_gemv_batch_args* pa, hipStream_t st) {
  const usdm_gemv_args& a = pa->g;
  USDM_CHECK_ARG(pa->nb >= 1 && pa->nb <= 16, "usdm_gemv_batch (matrix-core form): 1..16 sequences per step");
  USDM_CHECK_ARG(a.K % 32 == 0 && a.ldw % 8 == 0 && a.ldw >= a.K && pa->x_bs % 8 == 0, "usdm_gemv_batch (matrix-core form): K %% 32, ldw %% 8, x stride %% 8");
  const bool glu = a.act == USDM_ACT_SWIGLU, lmh = a.part_val != nullptr;
  MfmaDev d;
  d.ba = *pa;
  d.nchunks = a.K / 32;
  d.cpw = cdiv(d.nchunks, MW);
  const bool hold = d.cpw <= 8;
  USDM_CHECK_ARG(!a.norm_w || (hold && a.K <= GAM_FLOATS && a.K % 4 == 0), "usdm_gemv_batch (matrix-core form): the fused RMSNorm needs K <= 4096");
  USDM_CHECK_ARG(!lmh || (a.part_idx && !glu), "usdm_gemv_batch: lm_head partial buffers");
  d.nout = glu ? a.N / 2 : a.N;
  d.rt = 16;
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
  USDM_CHECK_ARG(cdiv(d.ntiles, d.grid) <= 64, "usdm_gemv_batch (matrix-core form): N too large (more than 64 tiles per workgroup)");
  USDM_CHECK_ARG(!lmh || pa->part_bs >= d.grid, "usdm_gemv_batch: part_bs must hold one partial per workgroup (%d)", d.grid);
  auto kh = gemv_mfma_kernel<true>;
  auto ks = gemv_mfma_kernel<false>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  if (hold) hipLaunchKernelGGL(kh, dim3(d.grid), dim3(MW * 64), LDS_BYTES, st, d);
  else hipLaunchKernelGGL(ks, dim3(d.grid), dim3(MW * 64), LDS_BYTES, st, d);
  USDM_LAUNCH_CHECK();
  return 0;
}
