// Flash-style attention forward on bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 online softmax.
//
// Replaces, without materialising any [S,S] tensor:
//   * Voicebox Attention.forward + the ALiBi/padding bias built in Transformer.forward
//     (networks.py:162-210, 319-341): bidirectional, bias = -slope_h*|i-j| with key column 0 = 0
//   * HF Mistral causal GQA attention in prefill (third-party)
//
// Orientation (CDNA4-specific): the wave computes S^T = K.Q^T, so one lane owns one QUERY column
// and the softmax statistics are lane-local; the exponentiated accumulator tile is then fed
// straight back as the B operand of O^T += V^T.P^T (no LDS round trip, no cross-lane traffic).
// V is therefore consumed as V^T [d][key], which the producing GEMM epilogue writes directly.
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KT = 64;  // keys per tile

template <int DH, int MODE>  // MODE 0: bidirectional + ALiBi + key-length mask ; 1: causal
__global__ __launch_bounds__(256) void attn_kernel(const usdm_attn_args a) {
  constexpr int DS = DH / 16;  // d-steps of QK^T
  constexpr int DT = DH / 32;  // 32-row tiles of O^T
  constexpr int KROW = DH * 2; // bytes per K row
  __shared__ __attribute__((aligned(16))) char smem[KT * KROW + DH * KT * 2];
  char* sK = smem;
  char* sV = smem + KT * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (a.Hq / a.Hkv);
  const int q0 = qb * 128 + wave * 32;
  const int kv_len = a.kv_len ? a.kv_len[b] : a.Skv;

  const bf16_t* Q = (const bf16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
  const bf16_t* K = (const bf16_t*)a.k + (int64_t)b * a.k_bs + (int64_t)hk * a.k_hs;
  const bf16_t* V = (const bf16_t*)a.vt + (int64_t)b * a.v_bs + (int64_t)hk * a.v_hs;

  // Q fragments (B operand): lane holds Q[q0+lq][16s + 8*lh .. +7]
  bf16x8 qf[DS];
  {
    int qr = q0 + lq;
    if (qr > a.Sq - 1) qr = a.Sq - 1;
    const bf16_t* qp = Q + (int64_t)qr * a.q_rs + 8 * lh;
#pragma unroll
    for (int s = 0; s < DS; ++s) qf[s] = __builtin_bit_cast(bf16x8, *(const u32x4*)(qp + 16 * s));
  }

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int qpos = a.q_pos0 + q0 + lq;  // absolute position of this lane's query
  const float slope = (MODE == 0 && a.slopes) ? a.slopes[h] : 0.f;
  const float sc = a.scale * 1.4426950408889634f;  // scores kept in log2 domain
  const float slope2 = slope * 1.4426950408889634f;

  int kend = kv_len;
  if (MODE == 1) {
    const int last_q = a.q_pos0 + min(qb * 128 + 127, a.Sq - 1);
    kend = min(kv_len, last_q + 1);
  }
  const int ntiles = (kend + KT - 1) / KT;

  // loader mapping: K tile = KT rows x (DH/8) 16-B pieces ; V^T tile = DH rows x 8 pieces
  constexpr int KP = KT * (DH / 8) / 256;  // pieces per thread
  constexpr int VP = DH * 8 / 256;
  u32x4 rk[KP], rv[VP];
  auto load_tile = [&](int kt) {
    const int k0 = kt * KT;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + 256 * i;
      const int row = p / (DH / 8), c = p % (DH / 8);
      rk[i] = *(const u32x4*)(K + (int64_t)(k0 + row) * a.k_rs + c * 8);
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int p = tid + 256 * i;
      const int d = p >> 3, c = p & 7;
      rv[i] = *(const u32x4*)(V + (int64_t)d * a.v_ds + k0 + c * 8);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + 256 * i;
      const int row = p / (DH / 8), c = p % (DH / 8);
      const int cs = (DH == 64) ? (c ^ ((row >> 1) & 7)) : (c ^ (row & 15));
      *(u32x4*)(sK + row * KROW + cs * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int p = tid + 256 * i;
      const int d = p >> 3, c = p & 7;
      const int f = (d >> 1) & 15;
      u32x2 lo = {rv[i][0], rv[i][1]}, hi = {rv[i][2], rv[i][3]};
      *(u32x2*)(sV + d * 128 + (((2 * c) ^ f) << 3)) = lo;
      *(u32x2*)(sV + d * 128 + (((2 * c + 1) ^ f) << 3)) = hi;
    }
  };

  if (ntiles > 0) load_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < ntiles) load_tile(kt + 1);

    // ---- S^T = K . Q^T for the two 32-key sub-tiles
    f32x16 sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
      const int row = 32 * u + lq;
#pragma unroll
      for (int s = 0; s < DS; ++s) {
        const int c = 2 * s + lh;
        const int cs = (DH == 64) ? (c ^ ((row >> 1) & 7)) : (c ^ (row & 15));
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const u32x4*)(sK + row * KROW + cs * 16));
        sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[u], 0, 0, 0);
      }
    }
    // ---- bias, mask, online softmax (log2 domain)
    float mloc = -1e30f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kpos = kt * KT + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float s = sacc[u][r] * sc;
        bool ok = kpos < kv_len;
        if (MODE == 0) {
          const int dlt = qpos > kpos ? qpos - kpos : kpos - qpos;
          if (kpos != 0 || !a.alibi_col0_zero) s -= slope2 * (float)dlt;
        } else {
          ok = ok && (kpos <= qpos);
        }
        s = ok ? s : -1e30f;
        sacc[u][r] = s;
        mloc = fmaxf(mloc, s);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = sacc[u][r];
        const float p = (s > -1e29f) ? exp2f(s - m_new) : 0.f;
        sacc[u][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;

    // ---- O^T += V^T . P^T   (P^T taken from the accumulator registers as the B operand)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)sacc[u][8 * s + j];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const int d = 32 * t + lq;
          const int f = (d >> 1) & 15;
          const int g = 8 * u + 4 * s + lh;
          const u32x2 lo = *(const u32x2*)(sV + d * 128 + ((g ^ f) << 3));
          const u32x2 hi = *(const u32x2*)(sV + d * 128 + (((g + 2) ^ f) << 3));
          const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
          oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[t], 0, 0, 0);
        }
      }
    __syncthreads();
  }

  // ---- normalise and store O[q][h*DH + d] (bf16)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int qr = q0 + lq;
  if (qr < a.Sq) {
    bf16_t* op = (bf16_t*)a.o + (int64_t)b * a.o_bs + (int64_t)qr * a.o_rs + h * DH;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * t + 8 * g + 4 * lh;
        uint2 o;
        o.x = pack_bf2(oacc[t][4 * g + 0] * inv, oacc[t][4 * g + 1] * inv);
        o.y = pack_bf2(oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
        *(uint2*)(op + d) = o;
      }
  }
}
}  // namespace

extern "C" int usdm_attention(const usdm_attn_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->q && pa->k && pa->vt && pa->o, "usdm_attention: null args");
  const usdm_attn_args& a = *pa;
  USDM_CHECK_ARG(a.dh == 64 || a.dh == 128, "usdm_attention: head dim %d unsupported (64/128)", a.dh);
  USDM_CHECK_ARG(a.B > 0 && a.Hq > 0 && a.Hkv > 0 && a.Hq % a.Hkv == 0 && a.Sq > 0 && a.Skv > 0, "usdm_attention: bad sizes");
  USDM_CHECK_ARG(a.Skv_alloc >= cdiv(a.Skv, KT) * KT, "usdm_attention: K/V^T buffers must be allocated (and finite) up to a multiple of %d keys", KT);
  USDM_CHECK_ARG(a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.v_ds % 8 == 0 && a.o_rs % 4 == 0, "usdm_attention: strides break 16-B alignment");
  USDM_CHECK_ARG(a.mode == 0 || a.mode == 1, "usdm_attention: mode");
  dim3 grid(cdiv(a.Sq, 128), a.Hq, a.B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (a.dh == 64 && a.mode == 0) hipLaunchKernelGGL((attn_kernel<64, 0>), grid, block, 0, st, a);
  else if (a.dh == 64) hipLaunchKernelGGL((attn_kernel<64, 1>), grid, block, 0, st, a);
  else if (a.mode == 0) hipLaunchKernelGGL((attn_kernel<128, 0>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((attn_kernel<128, 1>), grid, block, 0, st, a);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_attn_args(void) { return (int)sizeof(usdm_attn_args); }
