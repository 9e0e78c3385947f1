// Flash-style attention forward on bf16 MFMA, fp32 online softmax.
//
// Replaces, without materialising any [S,S] tensor:
//   * Voicebox Attention.forward + the ALiBi/padding bias built in Transformer.forward
//     (networks.py:162-210, 319-341): bidirectional, bias = -slope_h*|i-j| with key column 0 = 0
//   * HF Mistral causal GQA attention in prefill (third-party), with the sliding window of
//     src/model.py:337-371 as a key-range bound (usdm_attn_args.window)
//
// Orientation (CDNA4-specific): the wave computes S^T = K.Q^T, so one lane owns one QUERY column
// and the softmax statistics are lane-local (attn_kernel, v_mfma_f32_32x32x16_bf16) or shared by 4 lanes (attn16_kernel,
// v_mfma_f32_16x16x32_bf16); the exponentiated accumulator tile is then fed straight back as the B operand of
// O^T += V^T.P^T (no LDS round trip, no cross-lane traffic).
// V is therefore consumed as V^T [d][key], which the producing GEMM epilogue writes directly.
//   attn_kernel   : 4 waves x 32 queries per workgroup; every form (d = 64 / 128, bidirectional / causal, GQA)
//   attn16_kernel : 8 waves x 16 queries; the Voicebox form (d = 64, MHA, bidirectional), default since round 3
#include "common.h"
#include <stdlib.h>
#include "../../include/usdm_hip.h"

namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int KT = 64;  // keys per tile

#ifdef USDM_ATTN_TRACE
// debugging aid (tools/attn_trace.py): cycles per phase summed over the key tiles, wave 0 lane 0 of every workgroup
__device__ unsigned long long g_attn_trace[4096 * 8];
#define ATR_T() __builtin_readcyclecounter()
#define ATR_ADD(i, t0) do { const unsigned long long t1_ = ATR_T(); atr[i] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define ATR_T() 0ull
#define ATR_ADD(i, t0) do { } while (0)
#endif

// MODE 0: bidirectional + ALiBi + key-length mask ; 1: causal ; NW waves x 32 queries ; KS key-split groups: group g of NW
// waves walks key tiles g, g+KS, ... of the SAME queries and the groups' (m, l, O) are merged at the end.  KS = 2 puts two
// waves on every SIMD when the grid is about one workgroup per CU (Voicebox: 288 workgroups): one wave's softmax VALU work
// then overlaps the other's LDS-fed MFMAs (per-phase cycle counts in profiles/r01_gemm_ablation.txt).
template <int DH, int MODE, int NW, int KS = 1>
__global__ __launch_bounds__(NW * 64 * KS) void attn_kernel(const usdm_attn_args a) {
  constexpr int NT = NW * 64;   // threads of one key-split group (the loaders below are per group)
  constexpr int QB = NW * 32;   // queries per workgroup
  constexpr int DS = DH / 16;  // d-steps of QK^T
  constexpr int DT = DH / 32;  // 32-row tiles of O^T
  constexpr int KROW = DH * 2; // bytes per K row
  constexpr int STAGE = KT * KROW + DH * KT * 2;  // K tile + V^T tile
  __shared__ __attribute__((aligned(16))) char smem_all[2 * STAGE * KS];
  const int grp = threadIdx.x / NT;                 // key-split group of this wave
  char* smem = smem_all + grp * 2 * STAGE;

  const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  // (head, batch) of this workgroup.  Bidirectional ALiBi attention: heads are walked from the LAST (flattest slope: every
  // key tile counts) to the first (steepest: most far tiles are skipped below), batches innermost, so that when the grid is
  // a bit more than one round of workgroups the short ones are the stragglers' partners, not the long ones.
  const int qb = blockIdx.x;
  const int lin = blockIdx.y + gridDim.y * blockIdx.z;
  // MODE 0 head order (a.head_order; speed only): 0 = flattest first ... steepest last (round 2); 1 = the two steepest heads FIRST, the
  // next two LAST, flat heads in between: a grid of a little more than one workgroup per CU (288 on 256) is dealt round-robin, so
  // the CUs that receive a second workgroup hold the first and the last workgroups of the launch - with order 1 both are short
  // (steep slope: most key tiles skipped), instead of a flat-slope one plus a steep one.
  int h = (int)blockIdx.y;
  if (MODE == 0) {
    const int nh = (int)gridDim.y, p = lin / (int)gridDim.z;           // position of this (head, batch) pair in dispatch order
    if (a.head_order == 0 || nh < 8) h = nh - 1 - p;
    else h = p < 2 ? p : (p >= nh - 2 ? p - (nh - 2) + 2 : nh - 1 - (p - 2));
  }
  const int b = MODE == 0 ? lin % (int)gridDim.z : (int)blockIdx.z;
  const int hk = h / (a.Hq / a.Hkv);
  const int q0 = qb * QB + wave * 32;
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
  // ALiBi term of register pair (r, r+1) of sub-tile u: slope2 * (key offset inside the tile), for the packed fast path
  f32x2 cb[2][8];
  if (MODE == 0) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float c = (float)(32 * u + (r & 3) + 8 * (r >> 2));
        cb[u][r >> 1] = f32x2{slope2 * c, slope2 * (c + 1.0f)};
      }
  }

  int kend = kv_len;
  int kt_lo = 0;                                          // first key tile any query of this workgroup can see
  const int win = MODE == 1 ? a.window : 0;               // sliding window (causal mode): query p sees keys p-win+1 .. p
  if (MODE == 1) {
    const int last_q = a.q_pos0 + min(qb * QB + QB - 1, a.Sq - 1);
    kend = min(kv_len, last_q + 1);
    if (win > 0) kt_lo = max(0, a.q_pos0 + qb * QB - win + 1) / KT;
  }
  const int ntiles = max(0, (kend + KT - 1) / KT - kt_lo);   // tiles kt_lo .. kt_lo + ntiles - 1

  // loader mapping: K tile = KT rows x (DH/8) 16-B pieces ; V^T tile = DH rows x 8 pieces
  constexpr int KP = KT * (DH / 8) / NT;  // pieces per thread
  constexpr int VP = DH * 8 / NT;
  u32x4 rk[KP], rv[VP];
  auto load_tile = [&](int kt) {
    const int k0 = kt * KT;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NT * i;
      const int row = p / (DH / 8), c = p % (DH / 8);
      rk[i] = *(const u32x4*)(K + (int64_t)(k0 + row) * a.k_rs + c * 8);
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int p = tid + NT * i;
      const int d = p >> 3, c = p & 7;
      rv[i] = *(const u32x4*)(V + (int64_t)d * a.v_ds + k0 + c * 8);
    }
  };
  auto store_tile = [&](int stage) {
    char* sK = smem + stage * STAGE;
    char* sV = sK + KT * KROW;
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NT * i;
      const int row = p / (DH / 8), c = p % (DH / 8);
      const int cs = (DH == 64) ? (c ^ ((row >> 1) & 7)) : (c ^ (row & 15));
      *(u32x4*)(sK + row * KROW + cs * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int p = tid + NT * i;
      const int d = p >> 3, c = p & 7;
      const int f = (d >> 1) & 15;
      u32x2 lo = {rv[i][0], rv[i][1]}, hi = {rv[i][2], rv[i][3]};
      *(u32x2*)(sV + d * 128 + (((2 * c) ^ f) << 3)) = lo;
      *(u32x2*)(sV + d * 128 + (((2 * c + 1) ^ f) << 3)) = hi;
    }
  };

  // two LDS stages, one barrier per key tile: tile kt+1 is written to the other stage and tile kt+2 is
  // in flight from HBM/L2 while tile kt is multiplied
  unsigned long long atr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tph = ATR_T();
  const unsigned long long tstart = tph;
  // group-local tile sequence: it-th tile of this group is key tile grp + KS * it
  const int nit = (ntiles + KS - 1) / KS;                  // barrier count is the same for every group
  const int myn = ntiles > grp ? (ntiles - grp + KS - 1) / KS : 0;
  if (myn > 0) { load_tile(kt_lo + grp); store_tile(0); }
  // The Q fragments are consumed here on EVERY path into the loop.  Without this the wait-count pass merges the (myn == 0) path,
  // on which the Q loads are still in flight, into the loop header and puts s_waitcnt vmcnt(0) in front of the loop's first MFMAs:
  // every key tile then waits for the prefetch loads issued at the end of the previous tile, i.e. a full L2 / fabric latency per
  // tile with nothing to hide it (found in round 3 from the ISA; profiles/r03_vb_ablation.txt item 7).
#pragma unroll
  for (int s = 0; s < DS; ++s) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, qf[s])));
  if (myn > 1) load_tile(kt_lo + grp + KS);
  __syncthreads();
  ATR_ADD(0, tph);
  for (int it = 0; it < nit; ++it) {
    if (it >= myn) { __syncthreads(); continue; }          // this group has run out of tiles (wave-uniform)
    const int kt = kt_lo + grp + KS * it;
    const char* sK = smem + (it & 1) * STAGE;
    const char* sV = sK + KT * KROW;
    // a wave whose 32 queries all lie past the end (the last query block of S = 1118: queries 1120..1151) only helps with the
    // loads: its multiplies and exponentials would take vector issue slots from the co-resident workgroup for nothing
    if (q0 < a.Sq) {

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
    ATR_ADD(1, tph);
    // ---- bias, mask, online softmax (log2 domain).  VALU-lean: the key position of register r is a compile-time
    // constant plus a per-lane offset, masks are applied only on the (wave-uniform) tiles that need them.
    const int kbase = kt * KT + 4 * lh;                       // kpos(u,r) = kbase + 32u + (r&3) + 8(r>>2)
    const float fq = (float)(qpos - kbase);                   // qpos - kpos = fq - c(u,r)
    float mloc = -1e30f;
    // every key of this tile on one side of every query of this wave (all but <= 2 tiles of a row block): |q - k| has a
    // fixed sign, so the bias is sg * (A + cb) with A = -slope2 * fq per lane: two packed FMAs per score pair instead of
    // sub / mul|.| / fma per score (the loop is VALU-issue bound, profiles/r01_gemm_ablation.txt)
    const int qw0 = a.q_pos0 + q0;
    const bool k_left = kt * KT + KT - 1 <= qw0, k_right = kt * KT >= qw0 + 31;
    if (MODE == 0 && (k_left || k_right)) {
      const float sg = k_left ? 1.0f : -1.0f;
      const f32x2 sg2 = {sg, sg}, a2 = {-sg * slope2 * fq, -sg * slope2 * fq}, sc2 = {sc, sc};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 bias = __builtin_elementwise_fma(cb[u][r >> 1], sg2, a2);
          const f32x2 sv = __builtin_elementwise_fma(f32x2{sacc[u][r], sacc[u][r + 1]}, sc2, bias);
          sacc[u][r] = sv.x; sacc[u][r + 1] = sv.y;
        }
    } else {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float c = (float)(32 * u + (r & 3) + 8 * (r >> 2));
        float s;
        if (MODE == 0) s = fmaf(sacc[u][r], sc, -slope2 * fabsf(fq - c));
        else s = sacc[u][r] * sc;
        sacc[u][r] = s;
      }
    }
    if (MODE == 0 && kt == 0 && a.alibi_col0_zero) {          // key 0 carries no ALiBi bias (networks.py:327)
      if (lh == 0) sacc[0][0] = fmaf(sacc[0][0], 1.0f, slope2 * fabsf(fq));
    }
    const bool need_mask = (kt * KT + KT > kv_len) || (MODE == 1 && kt * KT + KT - 1 > a.q_pos0 + q0) ||
                           (win > 0 && kt * KT <= a.q_pos0 + q0 + 31 - win);
    if (need_mask) {
      // causal mode masks with -inf: with a window a query's FIRST tiles can be masked entirely, and exp2(-1e30 - m_run) is 1, not
      // 0, while m_run still holds its initial -1e30 (-inf - (-1e30) = -inf -> exactly 0; m_run itself never becomes -inf)
      const float masked = MODE == 1 ? -__builtin_inff() : -1e30f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kpos = kbase + 32 * u + (r & 3) + 8 * (r >> 2);
          const bool ok = (kpos < kv_len) && (MODE == 0 || (kpos <= qpos && (win <= 0 || kpos > qpos - win)));
          sacc[u][r] = ok ? sacc[u][r] : masked;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[u][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // A key tile whose best score lies 40 binades under the running maximum of EVERY query of the wave adds < 2^-34 to a
    // softmax denominator that is >= 1 (no change in f32) and < 2^-40 |v| per key to the output: its exponentials and its
    // P.V MFMAs are skipped.  With ALiBi (networks.py:319-341: slope 2^-(h+1)/2 per key of distance, key 0 unbiased) that is
    // most far tiles of the steep heads; fully masked tiles (bucket padding, scores -1e30) drop out the same way.
    // Causal mode (LLM prefill) skips FULLY MASKED tiles only, which is exact (p = 0, alpha = 1): whether a row's 2^-40-scale
    // terms are dropped would otherwise depend on its wave-mates, and the exact prefix reuse of the LLM promises rows that do
    // not depend on how many tokens were prefilled with them.
    const bool skip_tile = MODE == 0 ? __all(mloc < m_run - 40.0f) : __all(mloc < -1e29f);
    if (!skip_tile) {
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    f32x2 ps2 = {0.f, 0.f};
    const f32x2 mn2 = {m_new, m_new};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 dv = f32x2{sacc[u][r], sacc[u][r + 1]} - mn2;      // packed subtract
        const f32x2 pv = {__builtin_amdgcn_exp2f(dv.x), __builtin_amdgcn_exp2f(dv.y)};   // masked scores (-1e30) underflow to exactly 0
        sacc[u][r] = pv.x; sacc[u][r + 1] = pv.y;
        ps2 += pv;
      }
    const float psum = ps2.x + ps2.y;
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }

    ATR_ADD(2, tph);
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
    }
    }   // (active wave)
    ATR_ADD(3, tph);
    if (it + 1 < myn) store_tile((it + 1) & 1);
    if (it + 2 < myn) load_tile(kt + 2 * KS);
    ATR_ADD(4, tph);
    __syncthreads();
    ATR_ADD(5, tph);
  }
#ifdef USDM_ATTN_TRACE
  if (tid == 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 4096) {
      for (int i = 0; i < 6; ++i) g_attn_trace[wg * 8 + i] = atr[i];
      g_attn_trace[wg * 8 + 6] = ATR_T() - tstart;
      g_attn_trace[wg * 8 + 7] = ntiles;
    }
  }
#endif

  if constexpr (KS > 1) {
    // ---- merge the key-split groups: groups 1.. park (m, l, O^T) in LDS, group 0 folds them in (same wave / lane)
    static_assert(KS == 2, "merge is written for two groups");
    static_assert(NW * 64 * (2 + DT * 16) * 4 <= 2 * STAGE * KS, "merge buffer");
    float* mb = (float*)smem_all + (wave * 64 + lane) * (2 + DT * 16);
    __syncthreads();                                       // every tile buffer is dead
    if (grp == 1) {
      mb[0] = m_run; mb[1] = l_run;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mb[2 + t * 16 + r] = oacc[t][r];
    }
    __syncthreads();
    if (grp == 1) return;
    const float m1 = mb[0], l1 = mb[1];
    const float mm = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] = oacc[t][r] * a0 + mb[2 + t * 16 + r] * a1;
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

// ---- 16-query waves (Voicebox: bidirectional + ALiBi, d = 64) -------------------------------------------------------------------
// The 32-query kernel above keeps ONE wave per SIMD when the grid is about one workgroup per CU (2 x 16 heads x 9 query blocks =
// 288 workgroups on 256 CUs), and its loop is bound by the instruction stream of that single wave (profiles/r03_vb_ablation.txt
// item 2: ~8.5 cycles per instruction with nothing to switch to).  Here the same 128-query workgroup is EIGHT waves of 16 queries:
// S^T tiles are v_mfma_f32_16x16x32_bf16, two waves share a SIMD, and one wave's exponentials run under the other's MFMAs.
//   S^T[key][query]: A = K rows (lane l: row l&15, k = 8 (l>>4) + j), B = Q^T (lane: query l&15, same k), accumulator lane
//   (query l&15, rows 4 (l>>4) + i).  The P.V step needs, per lane, EIGHT consecutive keys of a 32-key block as its B operand
//   (k = 8 (l>>4) + j), but an accumulator holds four rows per 16-row tile: the two S tiles (t = 0, 1) of a 32-key block therefore
//   read K rows PERMUTED, tile t row rho <- key 8 (rho>>2) + 4 t + (rho&3), so that lane group g ends up with keys 8g .. 8g+7 in
//   (tile 0 regs 0..3, tile 1 regs 0..3): P^T goes from the accumulators to the MFMA B operand with no cross-lane traffic.
//   O^T[d][query] += V^T[d][key] . P^T: A = V^T rows as stored (lane: d = 16 dt + (l&15), keys 8 (l>>4) + j of the block).
// LDS image: rows of 128 B (K row = 64 d, V^T row = 64 keys), 16-B piece c of row r at c ^ f(r), f(r) = (r & 2) | ((r >> 1) & 4):
// conflict-free for both b128 read patterns above (checked exhaustively against the ds_read_b128 lane groups).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ __forceinline__ float quad_max16(float v) {    // max over the 4 lanes l, l^16, l^32, l^48 (v_permlane swaps, no LDS)
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_sum16(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// (A half-iteration skew of waves 4..7 against 0..3 - their barrier between softmax and P.V, four-stage ring - was measured too:
// within 1 % at S = 1118, +4 % only at S = 4096; not kept.  profiles/r03_vb_ablation.txt item 7.)
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_kernel(const usdm_attn_args a) {
  constexpr int NT = NW * 64, QB = NW * 16, STAGE = 2 * KT * 128, NP = 512 / NT;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q16 = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x;
  const int lin = blockIdx.y + gridDim.y * blockIdx.z;
  const int nh = (int)gridDim.y, p = lin / (int)gridDim.z;
  int h;
  if (a.head_order == 0 || nh < 8) h = nh - 1 - p;
  else h = p < 2 ? p : (p >= nh - 2 ? p - (nh - 2) + 2 : nh - 1 - (p - 2));
  const int b = lin % (int)gridDim.z;
  const int q0 = qb * QB + wave * 16;
  const int kv_len = a.kv_len ? a.kv_len[b] : a.Skv;
  const bf16_t* Q = (const bf16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
  const bf16_t* K = (const bf16_t*)a.k + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
  const bf16_t* V = (const bf16_t*)a.vt + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;

  bf16x8 qf[2];
  {
    int qr = q0 + q16;
    if (qr > a.Sq - 1) qr = a.Sq - 1;
    const bf16_t* qp = Q + (int64_t)qr * a.q_rs + 8 * g;
    qf[0] = __builtin_bit_cast(bf16x8, *(const u32x4*)qp);
    qf[1] = __builtin_bit_cast(bf16x8, *(const u32x4*)(qp + 32));
  }
  f32x4v oacc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) oacc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;
  const int qpos = a.q_pos0 + q0 + q16;
  const float slope2 = (a.slopes ? a.slopes[h] : 0.f) * 1.4426950408889634f;
  const float sc = a.scale * 1.4426950408889634f;
  // ALiBi term of register pair (i, i+1) of S tile (b32, t): slope2 * (key offset inside the 64-key tile, lane part 8g aside)
  f32x2 cb[2][2][2];
#pragma unroll
  for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const float c = (float)(32 * b32 + 4 * t + i);
        cb[b32][t][i >> 1] = f32x2{slope2 * c, slope2 * (c + 1.0f)};
      }
  const int ntiles = (kv_len + KT - 1) / KT;

  // loaders: one 16-B piece of the K tile and one of the V^T tile per thread
  const int lrow = tid >> 3, lc = tid & 7;                   // piece i of this thread: row lrow + (NT/8) i, 16-B column lc
  const bf16_t* kp = K + (int64_t)lrow * a.k_rs + lc * 8;
  const bf16_t* vp = V + (int64_t)lrow * a.v_ds + lc * 8;
  u32x4 rk[NP], rv[NP];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      rk[i] = *(const u32x4*)(kp + ((int64_t)kt * KT + (NT / 8) * i) * a.k_rs);
      rv[i] = *(const u32x4*)(vp + (int64_t)(NT / 8) * i * a.v_ds + kt * KT);
    }
  };
  auto store_tile = [&](int stage) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = lrow + (NT / 8) * i;
      const int lofs = r * 128 + ((lc ^ ((r & 2) | ((r >> 1) & 4))) << 4);
      *(u32x4*)(smem + stage * STAGE + lofs) = rk[i];
      *(u32x4*)(smem + stage * STAGE + KT * 128 + lofs) = rv[i];
    }
  };
  // fragment addresses of this lane (bytes inside a stage): K row of S tile (b32, t) and V^T row of d-tile dt
  int kofs[2][2][2], vofs[4][2];
#pragma unroll
  for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 32 * b32 + 8 * (q16 >> 2) + 4 * t + (q16 & 3);
      const int f = (row & 2) | ((row >> 1) & 4);
#pragma unroll
      for (int s = 0; s < 2; ++s) kofs[b32][t][s] = row * 128 + (((4 * s + g) ^ f) << 4);
    }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int d = 16 * dt + q16;
    const int f = (d & 2) | ((d >> 1) & 4);
#pragma unroll
    for (int b32 = 0; b32 < 2; ++b32) vofs[dt][b32] = KT * 128 + d * 128 + (((4 * b32 + g) ^ f) << 4);
  }

  unsigned long long atr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tph = ATR_T();
  const unsigned long long tstart = tph;
  if (ntiles > 0) { load_tile(0); store_tile(0); }
  asm volatile("" ::"v"(__builtin_bit_cast(u32x4, qf[0])), "v"(__builtin_bit_cast(u32x4, qf[1])));   // see attn_kernel: keeps vmcnt waits out of the loop head
  if (ntiles > 1) load_tile(1);
  __syncthreads();
  ATR_ADD(0, tph);
  const int qw0 = a.q_pos0 + q0;
  for (int kt = 0; kt < ntiles; ++kt) {
    const char* st = smem + (kt & 1) * STAGE;
    if (q0 < a.Sq) {
      f32x4v sacc[2][2];
#pragma unroll
      for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const u32x4*)(st + kofs[b32][t][s]));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc, 0, 0, 0);
          }
          sacc[b32][t] = acc;
        }
      ATR_ADD(1, tph);
      const int kbase = kt * KT + 8 * g;                       // kpos(b32, t, i) = kbase + 32 b32 + 4 t + i
      const float fq = (float)(qpos - kbase);
      const bool k_left = kt * KT + KT - 1 <= qw0, k_right = kt * KT >= qw0 + 15;
      if (k_left || k_right) {
        const float sg = k_left ? 1.0f : -1.0f;
        const f32x2 sg2 = {sg, sg}, a2 = {-sg * slope2 * fq, -sg * slope2 * fq}, sc2 = {sc, sc};
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const f32x2 bias = __builtin_elementwise_fma(cb[b32][t][i >> 1], sg2, a2);
              const f32x2 sv = __builtin_elementwise_fma(f32x2{sacc[b32][t][i], sacc[b32][t][i + 1]}, sc2, bias);
              sacc[b32][t][i] = sv.x; sacc[b32][t][i + 1] = sv.y;
            }
      } else {
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float c = (float)(32 * b32 + 4 * t + i);
              sacc[b32][t][i] = fmaf(sacc[b32][t][i], sc, -slope2 * fabsf(fq - c));
            }
      }
      if (kt == 0 && a.alibi_col0_zero && g == 0) sacc[0][0][0] += slope2 * fabsf(fq);      // key 0 carries no ALiBi bias
      if (kt * KT + KT > kv_len) {
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              sacc[b32][t][i] = (kbase + 32 * b32 + 4 * t + i < kv_len) ? sacc[b32][t][i] : -1e30f;
      }
      float mloc = -1e30f;
#pragma unroll
      for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) mloc = fmaxf(mloc, sacc[b32][t][i]);
      mloc = quad_max16(mloc);
      if (!__all(mloc < m_run - 40.0f)) {                      // same far-tile rule as attn_kernel (2^-40 under every row's maximum)
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        f32x2 ps2 = {0.f, 0.f};
        const f32x2 mn2 = {m_new, m_new};
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              const f32x2 dv = f32x2{sacc[b32][t][i], sacc[b32][t][i + 1]} - mn2;
              const f32x2 pv = {__builtin_amdgcn_exp2f(dv.x), __builtin_amdgcn_exp2f(dv.y)};
              sacc[b32][t][i] = pv.x; sacc[b32][t][i + 1] = pv.y;
              ps2 += pv;
            }
        l_run = l_run * alpha + (ps2.x + ps2.y);               // partial over this lane's keys; the 4 lanes of a query merge at the end
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
        }
        ATR_ADD(2, tph);
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32) {
          bf16x8 pf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { pf[j] = (__bf16)sacc[b32][0][j]; pf[4 + j] = (__bf16)sacc[b32][1][j]; }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 vf = __builtin_bit_cast(bf16x8, *(const u32x4*)(st + vofs[dt][b32]));
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
          }
        }
      }
    }
    ATR_ADD(3, tph);
    if (kt + 1 < ntiles) store_tile((kt + 1) & 1);
    if (kt + 2 < ntiles) load_tile(kt + 2);
    ATR_ADD(4, tph);
    __syncthreads();
    ATR_ADD(5, tph);
  }
#ifdef USDM_ATTN_TRACE
  if (tid == 0) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 4096) {
      for (int i = 0; i < 6; ++i) g_attn_trace[wg * 8 + i] = atr[i];
      g_attn_trace[wg * 8 + 6] = ATR_T() - tstart;
      g_attn_trace[wg * 8 + 7] = ntiles;
    }
  }
#endif
  const float l_tot = quad_sum16(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int qr = q0 + q16;
  if (qr < a.Sq) {
    bf16_t* op = (bf16_t*)a.o + (int64_t)b * a.o_bs + (int64_t)qr * a.o_rs + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 o;
      o.x = pack_bf2(oacc[dt][0] * inv, oacc[dt][1] * inv);
      o.y = pack_bf2(oacc[dt][2] * inv, oacc[dt][3] * inv);
      *(uint2*)(op + 16 * dt) = o;
    }
  }
}
}  // namespace

#ifdef USDM_ATTN_TRACE
extern "C" int usdm_dbg_attn_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_trace), sizeof(unsigned long long) * n);
}
#endif

extern "C" int usdm_attention(const usdm_attn_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->q && pa->k && pa->vt && pa->o, "usdm_attention: null args");
  const usdm_attn_args& a = *pa;
  USDM_CHECK_ARG(a.dh == 64 || a.dh == 128, "usdm_attention: head dim %d unsupported (64/128)", a.dh);
  USDM_CHECK_ARG(a.B > 0 && a.Hq > 0 && a.Hkv > 0 && a.Hq % a.Hkv == 0 && a.Sq > 0 && a.Skv > 0, "usdm_attention: bad sizes");
  USDM_CHECK_ARG(a.Skv_alloc >= cdiv(a.Skv, KT) * KT, "usdm_attention: K/V^T buffers must be allocated (and finite) up to a multiple of %d keys", KT);
  USDM_CHECK_ARG(a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.v_ds % 8 == 0 && a.o_rs % 4 == 0, "usdm_attention: strides break 16-B alignment");
  USDM_CHECK_ARG(a.mode == 0 || a.mode == 1, "usdm_attention: mode");
  USDM_CHECK_ARG(a.variant == 0 || a.variant == 1, "usdm_attention: variant");
  USDM_CHECK_ARG(a.window >= 0 && (a.window == 0 || a.mode == 1), "usdm_attention: window is a causal-mode (mode 1) option, >= 0");
  hipStream_t st = (hipStream_t)stream;
  usdm_attn_args a2 = a;
  if (a2.mode == 0 && a2.head_order == 0) a2.head_order = 1;      // default order of the bidirectional (Voicebox) form; -1 = flattest first
  else if (a2.head_order < 0) a2.head_order = 0;
  const bool small = false;       // (2-wave workgroups and the key-split pairing measured no gain and are not instantiated: r02 / r03 notes)
  dim3 grid(cdiv(a.Sq, 128), a.Hq, a.B), block(256);
  // 16-query waves (8 per workgroup) for the Voicebox form: MHA, d = 64, bidirectional (usdm_attn_args.variant = 1: the 32-query kernel)
  if (a.variant != 1 && a.dh == 64 && a.mode == 0 && a.Hq == a.Hkv) {
    hipLaunchKernelGGL((attn16_kernel<8>), grid, dim3(512), 0, st, a2);
    USDM_LAUNCH_CHECK();
    return 0;
  }
#define USDM_ATTN(DHV, MODEV)                                                                 \
  do {                                                                                         \
    hipLaunchKernelGGL((attn_kernel<DHV, MODEV, 4>), grid, block, 0, st, a2);                   \
  } while (0)
  if (a.dh == 64 && a.mode == 0) USDM_ATTN(64, 0);
  else if (a.dh == 64) USDM_ATTN(64, 1);
  else if (a.mode == 0) USDM_ATTN(128, 0);
  else USDM_ATTN(128, 1);
#undef USDM_ATTN
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_attn_args(void) { return (int)sizeof(usdm_attn_args); }
