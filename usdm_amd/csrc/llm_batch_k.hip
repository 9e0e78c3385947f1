// Batched decode (several utterances decoded in lockstep): the weight-streaming GEMV with NB input
// vectors.  Same streaming structure as gemv_kernel (llm_k.hip): every weight byte is still read exactly once per STEP,
// now amortised over NB tokens; per item the arithmetic (lane partition of K, accumulation order, rounding points) is
// identical to the batch-1 kernel, so a batched step reproduces NB independent steps bit for bit.
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ float dot8b(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = w[i], b = x[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  }
  return acc;
}

// (NB = 4 needs 140 VGPRs in the gate/up variant = 3 workgroups per SIMD instead of 4, i.e. a third round of workgroups for
// the 1792-workgroup launch: 57 us instead of 40.  Forcing 128 VGPRs spills and was measured slower: 898 vs 967 tok/s.)
template <int RW, bool GLU, int NWV, int NB>
__global__ __launch_bounds__(NWV * 64) void gemv_batch_kernel(const usdm_gemv_batch_args ba) {
  const usdm_gemv_args& a = ba.g;
  constexpr int NTH = NWV * 64;
  constexpr int NR = GLU ? 2 * RW : RW;
  constexpr int UNR = (NR >= 8) ? 2 : (NR >= 4) ? 4 : (NR == 3 ? 5 : 8);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;  // [NB][Kpad] bf16, zero padded
  __shared__ float red[NB][NWV];
  __shared__ float sv[NWV];
  __shared__ int si[NWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int Kpad = (K + 511) & ~511;
  const int nit = Kpad >> 9;

  const int rows_per_block = NWV * RW;
  const int ob = blockIdx.x * rows_per_block + wave * RW;
  const u32x4* wp[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    int r;
    if (GLU) {
      const int o = ob + (j % RW);
      r = (o >> 4) * 32 + (o & 15) + (j >= RW ? 16 : 0);
    } else {
      r = ob + j;
    }
    r = r < a.N ? r : a.N - 1;
    wp[j] = (const u32x4*)((const bf16_t*)a.W + (int64_t)r * a.ldw) + lane;
  }
  const bool tail_ok = ((nit - 1) << 9) + lane * 8 < K;
  // lm_head mode: rows of banned ids are not streamed (see gemv_kernel)
  bool active = true;
  if (a.part_val && a.ban) {
    const int wb = blockIdx.x * rows_per_block;
    bool wg_active = false;
    for (int r = wb; r < wb + rows_per_block && r < a.N; ++r) wg_active |= (a.ban[r] == 0);
    if (!wg_active) {
      if (tid < NB) {
        a.part_val[(int64_t)tid * ba.part_bs + blockIdx.x] = -INFINITY;
        a.part_idx[(int64_t)tid * ba.part_bs + blockIdx.x] = 0x7fffffff;
      }
      if (a.y32 && tid < rows_per_block && wb + tid < a.N)
        for (int b = 0; b < NB; ++b) a.y32[(int64_t)b * ba.y_bs + wb + tid] = -INFINITY;
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
    const u32x4* p = (it == nit - 1 && !tail_ok) ? wp[j] - lane : wp[j] + it * 64;
    return __builtin_nontemporal_load(p);
  };
  u32x4 ring[NR][UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u)
#pragma unroll
    for (int j = 0; j < NR; ++j)
      if (u < nit) ring[j][u] = wload(j, u);

  // ---- stage the NB input vectors (optionally RMS-normalised, HF rounding) while the first ring is in flight
  if (a.norm_w) {
    float ss[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ss[b] = 0.f;
      const bf16_t* xg = (const bf16_t*)a.x + (int64_t)b * ba.x_bs;
      for (int i = tid * 8; i < K; i += NTH * 8) {
        const u32x4 v = *(const u32x4*)(xg + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = bf2f(v[e] & 0xffff), hi = bf2f(v[e] >> 16);
          ss[b] += lo * lo + hi * hi;
        }
      }
      ss[b] = wave_sum(ss[b]);
      if (lane == 0) red[b][wave] = ss[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) tot += red[b][w];
      const float rstd = rsqrtf(tot / (float)K + a.eps);
      const bf16_t* xg = (const bf16_t*)a.x + (int64_t)b * ba.x_bs;
      for (int i = tid * 8; i < Kpad; i += NTH * 8) {
        u32x4 o = {0, 0, 0, 0};
        if (i < K) {
          const u32x4 v = *(const u32x4*)(xg + i);
          const float4 g0 = *(const float4*)(a.norm_w + i), g1 = *(const float4*)(a.norm_w + i + 4);
          const float gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = bf2f(v[e] & 0xffff), hi = bf2f(v[e] >> 16);
            o[e] = pack_bf2(round_bf(round_bf(lo * rstd) * gw[2 * e]), round_bf(round_bf(hi * rstd) * gw[2 * e + 1]));
          }
        }
        *(u32x4*)(xs + (int64_t)b * Kpad + i) = o;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bf16_t* xg = (const bf16_t*)a.x + (int64_t)b * ba.x_bs;
      for (int i = tid * 8; i < Kpad; i += NTH * 8) {
        u32x4 v = {0, 0, 0, 0};
        if (i < K) v = *(const u32x4*)(xg + i);
        *(u32x4*)(xs + (int64_t)b * Kpad + i) = v;
      }
    }
  }
  __syncthreads();

  float acc[NR][NB];
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[j][b] = 0.f;
  for (int it0 = 0; it0 < nit; it0 += UNR) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int it = it0 + u;
      if (it < nit) {
        u32x4 xv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) xv[b] = *(const u32x4*)(xs + (int64_t)b * Kpad + (it * 64 + lane) * 8);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[j][b] = dot8b(ring[j][u], xv[b], acc[j][b]);
          if (it + UNR < nit) ring[j][u] = wload(j, it + UNR);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[j][b] = wave_sum(acc[j][b]);

  if (a.part_val) {  // lm_head: bf16-rounded logits, ban mask, per-block arg-max per item (ties -> lowest id)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ob + j;
        if (n < a.N && !(a.ban && a.ban[n])) {
          const float v = round_bf(acc[j][b]);
          if (a.y32 && lane == 0) a.y32[(int64_t)b * ba.y_bs + n] = v;
          if (v > bv) { bv = v; bi = n; }
        } else if (n < a.N && a.y32 && lane == 0) {
          a.y32[(int64_t)b * ba.y_bs + n] = -INFINITY;
        }
      }
      if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < NWV; ++w)
          if (sv[w] > bv) { bv = sv[w]; bi = si[w]; }
        a.part_val[(int64_t)b * ba.part_bs + blockIdx.x] = bv;
        a.part_idx[(int64_t)b * ba.part_bs + blockIdx.x] = bi == 0x7fffffff ? bi : bi + a.idx_offset;
      }
      __syncthreads();
    }
    return;
  }
  if (lane != 0) return;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (GLU) {
#pragma unroll
      for (int j = 0; j < RW; ++j) {
        const int o = ob + j;
        if (2 * o >= a.N) continue;
        const float g = acc[j][b], u = acc[j + RW][b];
        float r;
        if (a.round_bf16) {
          const float gt = round_bf(g), up = round_bf(u);
          r = round_bf(round_bf(gt / (1.0f + __expf(-gt))) * up);
        } else {
          r = (g / (1.0f + __expf(-g))) * u;
        }
        if (a.y16) ((bf16_t*)a.y16)[(int64_t)b * ba.y_bs + o] = f2bf(r);
        if (a.y32) a.y32[(int64_t)b * ba.y_bs + o] = r;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = ob + j;
        if (n >= a.N) continue;
        float v = acc[j][b];
        if (a.round_bf16) v = round_bf(v);
        if (a.residual) {
          v += bf2f(((const bf16_t*)a.residual)[(int64_t)b * ba.res_bs + n]);
          if (a.round_bf16) v = round_bf(v);
        }
        if (a.y16) ((bf16_t*)a.y16)[(int64_t)b * ba.y_bs + n] = f2bf(v);
        if (a.y32) a.y32[(int64_t)b * ba.y_bs + n] = v;
      }
    }
  }
}

static int pick_rw(int nout, bool glu) {   // same balance rule as the batch-1 launcher
  const int ncand = glu ? 2 : 4;
  const int cands[4] = {glu ? 2 : 4, glu ? 1 : 3, 2, 1};
  int best = cands[ncand - 1];
  double best_score = -1.0;
  for (int c = 0; c < ncand; ++c) {
    const int rw = cands[c];
    const int blocks = cdiv(nout, 4 * rw);
    const double eff = (blocks / 256.0) / (double)((blocks + 255) / 256);
    if (blocks >= 1024 && eff >= 0.9) return rw;
    const double score = eff * (blocks >= 512 ? 1.0 : 0.5 + blocks / 1024.0);
    if (score > best_score) { best_score = score; best = rw; }
  }
  return best;
}

template <int NB>
int launch_nb(const usdm_gemv_batch_args& ba, hipStream_t st) {
  const usdm_gemv_args& a = ba.g;
  const bool glu = a.act == USDM_ACT_SWIGLU;
  const int nout = glu ? a.N / 2 : a.N;
  const int Kpad = (a.K + 511) & ~511;
  const size_t lds = (size_t)Kpad * 2 * NB;
#define USDM_GB(RW, GLUV, NWV, GRID)                                                                              \
  do {                                                                                                             \
    auto kfn = gemv_batch_kernel<RW, GLUV, NWV, NB>;                                                               \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
    hipLaunchKernelGGL(kfn, dim3(GRID), dim3(NWV * 64), lds, st, ba);                                              \
  } while (0)
  if (!glu && !a.part_val && nout % 256 == 0 && (nout / 256 == 16 || nout / 256 == 24)) {
    if (nout / 256 == 16) USDM_GB(1, false, 16, 256);
    else USDM_GB(2, false, 12, 256);
  } else {
    const int rw = a.part_val ? 4 : pick_rw(nout, glu);
    const int grid = cdiv(nout, 4 * rw);
    if (glu) {
      if (rw == 2) USDM_GB(2, true, 4, grid);
      else USDM_GB(1, true, 4, grid);
    } else {
      if (rw == 4) USDM_GB(4, false, 4, grid);
      else if (rw == 3) USDM_GB(3, false, 4, grid);
      else if (rw == 2) USDM_GB(2, false, 4, grid);
      else USDM_GB(1, false, 4, grid);
    }
  }
#undef USDM_GB
  USDM_LAUNCH_CHECK();
  return 0;
}
}  // namespace

int usdm_gemv_mfma_launch(const usdm_gemv_batch_args* pa, hipStream_t st);   // llm_mfma_k.hip: the matrix-core form, 1..16 sequences

extern "C" int usdm_gemv_batch(const usdm_gemv_batch_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->g.W && pa->g.x, "usdm_gemv_batch: null args");
  const usdm_gemv_args& a = pa->g;
  USDM_CHECK_ARG(pa->g.N > 0 && pa->g.K > 0, "usdm_gemv_batch: bad N/K");
  USDM_CHECK_ARG(a.y16 || a.y32 || a.part_val, "usdm_gemv_batch: no output");
  USDM_CHECK_ARG(!a.x_delta && !a.x_out, "usdm_gemv_batch: x_delta / x_out are batch-1 (tensor-parallel) only");
  if (pa->form == 1 || pa->form == 3 || pa->form == 5 || (pa->form == 0 && pa->nb > 4)) return usdm_gemv_mfma_launch(pa, (hipStream_t)stream);
  USDM_CHECK_ARG(pa->nb >= 1 && pa->nb <= 4, "usdm_gemv_batch: the VALU form takes 1..4 sequences per step (form = 1 or nb > 4: matrix cores, <= 16)");
  USDM_CHECK_ARG(a.N > 0 && a.K > 0 && a.K % 8 == 0 && a.ldw % 8 == 0 && a.ldw >= a.K, "usdm_gemv_batch: bad N/K/ldw");
  USDM_CHECK_ARG(a.K <= 16384, "usdm_gemv_batch: K too large for the LDS-resident input vectors");
  const bool glu = a.act == USDM_ACT_SWIGLU;
  USDM_CHECK_ARG(!glu || a.N % 32 == 0, "usdm_gemv_batch: swiglu needs N %% 32 == 0");
  USDM_CHECK_ARG(a.y16 || a.y32 || a.part_val, "usdm_gemv_batch: no output");
  USDM_CHECK_ARG(!a.part_val || (a.part_idx && !glu && pa->part_bs >= cdiv(a.N, 16)), "usdm_gemv_batch: lm_head partial buffers");
  USDM_CHECK_ARG(!a.x_delta && !a.x_out, "usdm_gemv_batch: x_delta / x_out are batch-1 (tensor-parallel) only");
  USDM_CHECK_ARG(pa->x_bs % 8 == 0, "usdm_gemv_batch: x stride must keep 16-B alignment");
  hipStream_t st = (hipStream_t)stream;
  switch (pa->nb) {
    case 1: return launch_nb<1>(*pa, st);
    case 2: return launch_nb<2>(*pa, st);
    case 3: return launch_nb<3>(*pa, st);
    default: return launch_nb<4>(*pa, st);
  }
}
extern "C" int usdm_sizeof_gemv_batch_args(void) { return (int)sizeof(usdm_gemv_batch_args); }
