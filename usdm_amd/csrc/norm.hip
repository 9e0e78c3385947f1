// LayerNorm / RMSNorm over the channel axis of channels-last activations: one wave per row.
// Replaces nn.LayerNorm at networks.py:245-247,297 (Voicebox), the per-conv / per-layer LayerNorms
// of the XLS-R encoder and HF MistralRMSNorm (third-party).
// HBM-bound: each row is read once (kept in registers), written once per requested output.
#include "common.h"
#include "../../include/usdm_hip.h"

// outputs as write-through stores (see st_wt in common.h; NFE -0.8 %, profiles/r04_gemm_ablation.txt E)
#ifndef USDM_NORM_WT
#define USDM_NORM_WT 1   // 0: plain stores (A/B builds)
#endif
#if USDM_NORM_WT
#define NST(p, v) st_wt(p, v)
#else
#define NST(p, v) (*(p) = (v))
#endif
namespace {
constexpr int MAXP = 20;  // float4 pieces per lane: C <= 5120

template <int NP>
__global__ __launch_bounds__(256) void norm_kernel(const usdm_norm_args a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int npieces = a.C >> 2;
  float4 v[NP];
  // gamma / beta do not depend on the row statistics: requested together with the row so that the kernel pays one memory
  // latency, not two.  beta: NP <= 5 only (the wide instantiations would spill); gamma: up to NP = 16 (round 4: the 7B's RMSNorm
  // rows of 4096 loaded gamma piece by piece behind the statistics - 16 dependent L2 latencies, 15 - 17 us per launch of a prefill)
  constexpr bool PRE = NP <= 16, PREB = NP <= 5;
  float4 gpre[PRE ? NP : 1], bpre[PREB ? NP : 1];
  if constexpr (PRE) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int idc = min(p * 64 + lane, npieces - 1) * 4;      // (unconditional: pieces past the row are never used)
      gpre[p] = *(const float4*)(a.gamma + idc);
    }
    if constexpr (PREB) {
      if (a.beta) {
#pragma unroll
        for (int p = 0; p < NP; ++p) bpre[p] = *(const float4*)(a.beta + min(p * 64 + lane, npieces - 1) * 4);
      }
    }
  }
  // Loads first, arithmetic second (round 4): with the dtype / residual / addend tests inside the per-piece loop every piece paid its
  // own memory latencies one after the other (the wait-count pass drains at each join) - 14 us for a 38 x 4096 RMSNorm.  The
  // per-element order of the additions is unchanged (x, + res, rounding, + res2[0], + res2[1] ...), so results are bit-identical.
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint2 zero2 = make_uint2(0u, 0u);
  auto cvt = [](uint2 r) { return make_float4(bf2f(r.x & 0xffff), bf2f(r.x >> 16), bf2f(r.y & 0xffff), bf2f(r.y >> 16)); };
  if constexpr (NP <= 5) {
    // narrow rows (Voicebox, XLS-R): EVERY source - x, the residual, up to three split-K addends - is requested before the first
    // addition, whatever the dtypes (the branches below hold loads only: nothing waits at their joins)
    constexpr int ME = 3;
    float4 x4[NP], q4[NP], e4[ME][NP];
    uint2 x2[NP], q2[NP];
    const bool xf = a.x_dtype == USDM_F32, rf = a.res_dtype == USDM_F32;
    const int ne = a.n_res2 < ME ? a.n_res2 : ME;
    // per-lane validity is applied AFTER the loads (pieces past the row re-read its last piece): a predicated load would have
    // to be merged with its zero at the join, i.e. waited for on the spot
    int ic[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) ic[p] = min(p * 64 + lane, npieces - 1) * 4;
    // (x4 / x2, q4 / q2, e4: only the set that was loaded is read below - the other stays unwritten ON PURPOSE: initialising it would
    // put a merge of {zero, loaded} at the join of the branch, and the wait-count pass makes the wave wait for the load there)
    if (xf) {
#pragma unroll
      for (int p = 0; p < NP; ++p) x4[p] = *(const float4*)((const float*)a.x + (int64_t)row * a.ldx + ic[p]);
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) x2[p] = *(const uint2*)((const bf16_t*)a.x + (int64_t)row * a.ldx + ic[p]);
    }
    if (a.res) {
      if (rf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) q4[p] = *(const float4*)((const float*)a.res + (int64_t)row * a.ldr + ic[p]);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) q2[p] = *(const uint2*)((const bf16_t*)a.res + (int64_t)row * a.ldr + ic[p]);
      }
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      if (e < ne) {
        const float* rp = a.res2 + (int64_t)e * a.res2_stride + (int64_t)row * a.ldr;
#pragma unroll
        for (int p = 0; p < NP; ++p) e4[e][p] = *(const float4*)(rp + ic[p]);
      }
    }
    // ---- arithmetic, in the order the piece-by-piece form used: x, + res, rounding, + addend 0, + addend 1 ...
#pragma unroll
    for (int p = 0; p < NP; ++p) v[p] = (p * 64 + lane < npieces) ? (xf ? x4[p] : cvt(x2[p])) : zero4;
    if (a.res) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float4 r4 = (p * 64 + lane < npieces) ? (rf ? q4[p] : cvt(q2[p])) : zero4;
        v[p].x += r4.x; v[p].y += r4.y; v[p].z += r4.z; v[p].w += r4.w;
      }
      if (a.round_bf16) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { v[p].x = round_bf(v[p].x); v[p].y = round_bf(v[p].y); v[p].z = round_bf(v[p].z); v[p].w = round_bf(v[p].w); }
      }
    }
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      if (e < ne) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const float4 r4 = (p * 64 + lane < npieces) ? e4[e][p] : zero4;
          v[p].x += r4.x; v[p].y += r4.y; v[p].z += r4.z; v[p].w += r4.w;
        }
      }
    }
    for (int e2 = ME; e2 < a.n_res2; ++e2) {
      const float* rp = a.res2 + (int64_t)e2 * a.res2_stride + (int64_t)row * a.ldr;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p * 64 + lane < npieces) {
          const float4 r4 = *(const float4*)(rp + (p * 64 + lane) * 4);
          v[p].x += r4.x; v[p].y += r4.y; v[p].z += r4.z; v[p].w += r4.w;
        }
      }
    }
  } else {
    // wide rows (the 7B's 4096): x for all pieces first; residual / addends piece by piece (a second full register set would spill)
    if (a.x_dtype == USDM_F32) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int idx = p * 64 + lane;
        v[p] = idx < npieces ? *(const float4*)((const float*)a.x + (int64_t)row * a.ldx + idx * 4) : zero4;
      }
    } else {
      uint2 r[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int idx = p * 64 + lane;
        r[p] = idx < npieces ? *(const uint2*)((const bf16_t*)a.x + (int64_t)row * a.ldx + idx * 4) : zero2;
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) v[p] = cvt(r[p]);
    }
    if (a.res) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int idx = p * 64 + lane;
        if (idx < npieces) {
          const float4 r4 = a.res_dtype == USDM_F32 ? *(const float4*)((const float*)a.res + (int64_t)row * a.ldr + idx * 4)
                                                    : cvt(*(const uint2*)((const bf16_t*)a.res + (int64_t)row * a.ldr + idx * 4));
          v[p].x += r4.x; v[p].y += r4.y; v[p].z += r4.z; v[p].w += r4.w;
        }
      }
      if (a.round_bf16) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { v[p].x = round_bf(v[p].x); v[p].y = round_bf(v[p].y); v[p].z = round_bf(v[p].z); v[p].w = round_bf(v[p].w); }
      }
    }
    for (int e2 = 0; e2 < a.n_res2; ++e2) {
      const float* rp = a.res2 + (int64_t)e2 * a.res2_stride + (int64_t)row * a.ldr;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int idx = p * 64 + lane;
        if (idx < npieces) {
          const float4 r4 = *(const float4*)(rp + idx * 4);
          v[p].x += r4.x; v[p].y += r4.y; v[p].z += r4.z; v[p].w += r4.w;
        }
      }
    }
  }
  // (the padding-row test is made where it is first needed: AFTER the row's loads were issued: at the join of this branch the wait-count pass waits for everything requested so far)
  int vlen = 0x7fffffff, srow = 0;
  if (a.valid_len) {
    const int b = row / a.rows_per_batch;
    srow = row - b * a.rows_per_batch;
    vlen = a.valid_len[b];
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = p * 64 + lane;
    if (idx < npieces) {
      if (a.premask && srow >= vlen) v[p] = zero4;
      if (a.sum32) NST((float4*)((float*)a.sum32 + (int64_t)row * a.lds + idx * 4), v[p]);
      if (a.sum16) {
        uint2 o; o.x = pack_bf2(v[p].x, v[p].y); o.y = pack_bf2(v[p].z, v[p].w);
        NST((uint2*)((bf16_t*)a.sum16 + (int64_t)row * a.lds + idx * 4), o);
      }
    } else {
      v[p] = zero4;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) s += (v[p].x + v[p].y) + (v[p].z + v[p].w);
  const float invC = 1.0f / (float)a.C;
  float mean = 0.f;
  if (!a.rms) mean = wave_sum(s) * invC;
  float q = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = p * 64 + lane;
    if (idx < npieces) {
      const float dx = v[p].x - mean, dy = v[p].y - mean, dz = v[p].z - mean, dw = v[p].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float var = wave_sum(q) * invC;
  const float rstd = rsqrtf(var + a.eps);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = p * 64 + lane;
    if (idx >= npieces) continue;
    float4 gm, bt = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (PRE) {
      gm = gpre[p];
      if constexpr (PREB) { if (a.beta) bt = bpre[p]; }
      else if (a.beta) bt = *(const float4*)(a.beta + idx * 4);
    } else {
      gm = *(const float4*)(a.gamma + idx * 4);
      if (a.beta) bt = *(const float4*)(a.beta + idx * 4);
    }
    float y[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
    const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = (y[e] - mean) * rstd;
      if (a.round_bf16) t = round_bf(round_bf(t) * g4[e]);  // HF: weight * hidden.to(bf16)
      else t = t * g4[e] + b4[e];
      if (a.act == USDM_ACT_GELU) t = gelu_erf(t);
      if (srow >= vlen) t = 0.f;
      y[e] = t;
    }
    if (a.out32) NST((float4*)((float*)a.out32 + (int64_t)row * a.ldo + idx * 4), make_float4(y[0], y[1], y[2], y[3]));
    if (a.out16) {
      uint2 o; o.x = pack_bf2(y[0], y[1]); o.y = pack_bf2(y[2], y[3]);
      NST((uint2*)((bf16_t*)a.out16 + (int64_t)row * a.ldo + idx * 4), o);
    }
  }
}
}  // namespace

extern "C" int usdm_norm(const usdm_norm_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->x && pa->gamma, "usdm_norm: null args");
  const usdm_norm_args& a = *pa;
  USDM_CHECK_ARG(a.C > 0 && a.C % 4 == 0 && a.C <= MAXP * 256, "usdm_norm: C=%d must be a multiple of 4 and <= %d", a.C, MAXP * 256);
  USDM_CHECK_ARG(a.rows > 0, "usdm_norm: rows");
  USDM_CHECK_ARG(a.out32 || a.out16, "usdm_norm: no output");
  USDM_CHECK_ARG(!a.valid_len || a.rows_per_batch > 0, "usdm_norm: rows_per_batch");
  USDM_CHECK_ARG(a.n_res2 >= 0 && a.n_res2 <= 8 && (a.n_res2 == 0 || (a.res2 && a.res2_stride % 4 == 0 && !a.round_bf16)), "usdm_norm: res2");
  USDM_CHECK_ARG(a.ldx % 4 == 0 && a.ldo % 4 == 0 && a.ldr % 4 == 0 && a.lds % 4 == 0, "usdm_norm: strides must be multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(a.rows, 4)), block(256);
  const int np = cdiv(a.C, 256);
  if (np <= 2) hipLaunchKernelGGL(norm_kernel<2>, grid, block, 0, st, a);
  else if (np == 4) hipLaunchKernelGGL(norm_kernel<4>, grid, block, 0, st, a);   // C = 1024 (Voicebox): no idle fifth piece
  else if (np <= 5) hipLaunchKernelGGL(norm_kernel<5>, grid, block, 0, st, a);
  else if (np <= 16) hipLaunchKernelGGL(norm_kernel<16>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(norm_kernel<MAXP>, grid, block, 0, st, a);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_norm_args(void) { return (int)sizeof(usdm_norm_args); }
