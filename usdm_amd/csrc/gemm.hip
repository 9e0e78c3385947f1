// Universal tap-GEMM for gfx950 (see include/usdm_hip.h, usdm_gemm_args).
//
// Work decomposition (MI355X-first, not a translation of anything in the reference, which only
// calls torch.nn.functional):
//   * one workgroup = 256 threads = 4 waves in a 2x2 arrangement over a BM x BN output tile
//   * K is streamed in "chunks" of 64 bytes per row (32 bf16 / 16 f32); a K-step is two chunks.
//     A chunk belongs to exactly one tap, so convolution taps, dilation, stride, zero padding and
//     the two-source skip-concat are all just a per-chunk row/column remap of the A loads.
//   * global -> VGPR (16-B raw buffer loads; out-of-range rows are redirected to an out-of-bounds
//     offset so the hardware range check returns zeros: no divergent branches) -> XOR-swizzled LDS
//     (conflict-free ds_read_b128 for the 16x16 MFMA fragment shape) -> MFMA.
//   * two LDS stages, ONE barrier per K-step: the loads of step k+2 are issued, and step k+1 is
//     written to the other stage, while step k is multiplied.
//   * bf16: v_mfma_f32_16x16x32_bf16 ; f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
//   * epilogue through LDS: the accumulator tile is transposed in LDS so that bias / activation /
//     residual / rounding run on 4 consecutive outputs per thread and every global access is
//     8-16 B wide (row-major, transposed, head-split QKV and SwiGLU layouts alike).
//   * loader variants of the same kernel: register-staged (above), LDS-DMA with 2-4 stages, and - for the
//     big single-tap bf16 GEMMs (Voicebox QKV / FFN, LLM prefill) - the 8-wave PING-PONG loop on 256x128,
//     288x128 and 128x128 tiles at one workgroup per CU ("8-wave ping-pong loop" below; selection in
//     usdm_gemm; measurements in profiles/r02_gemm_ablation.txt sections 3-4).
#include "common.h"
#include "../../include/usdm_hip.h"
#include <stdlib.h>
#include <utility>
#include <type_traits>

// Epilogue outputs leave as WRITE-THROUGH stores (sc1; round 4).  A launch's output is consumed by the NEXT launch, mostly from other
// XCDs (their L2s are not coherent with ours), so plain stores only park 13 - 27 MB of dirty lines in this XCD's L2 for the
// write-back at the kernel boundary; written through while the epilogue runs, the boundary has nothing left to flush: Voicebox NFE
// -3.3 % on two boxes (profiles/r04_gemm_ablation.txt E; `nt` stores instead: +7.5 %, the consumers then miss the last-level cache
// as well).  A stronger scope bit is always a legal store.  The s_nop covers the store-data hazard hipcc handles for its own stores
// (a VALU write to the data registers of a > 8-byte store in the next cycle): without it one ragged-tile test read garbage.
#ifndef USDM_GEMM_WT_STORES
#define USDM_GEMM_WT_STORES 1   // 0: plain stores (A/B builds)
#endif
template <class T>
__device__ __forceinline__ void st_out(T* p, const T& v) {
#if USDM_GEMM_WT_STORES
  if constexpr (sizeof(T) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x4, v)) : "memory");
  else if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u32x2, v)) : "memory");
  else *p = v;
#else
  *p = v;
#endif
}
#ifndef USDM_GEMM_WIDE16
#define USDM_GEMM_WIDE16 1   // 0: the 8-byte-per-lane GELU / head-split epilogues of round 3 (A/B builds only)
#endif
namespace {

struct GemmDev {
  usdm_gemm_args a;
  int tiles_m, tiles_n;
};

// compile-time loop: every accumulator index below is a constant, so nothing can fall into scratch
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// LDS image of one operand stage: [sub-chunk 0..1][row][4 pieces of 16 B], piece index XOR f(row).
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ float silu_mul(float g, float u, bool rbf) {
  if (rbf) {
    g = round_bf(g); u = round_bf(u);
    return round_bf(round_bf(g / (1.0f + __expf(-g))) * u);
  }
  return (g / (1.0f + __expf(-g))) * u;
}

#ifdef USDM_GEMM_TRACE
// debugging aid (tools/gemm_trace.py): per-workgroup phase timestamps (100 MHz wall clock) + hardware ids
__device__ unsigned long long g_gemm_trace[8192 * 8];
#define TR(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TR(i) do { } while (0)
#endif

// bias of this thread's epilogue columns -> registers (SwiGLU: gate / up), per-column bias of the transposed epilogues -> LDS.
// A macro, not a lambda: captured by reference the small arrays fall into scratch.  Expanded before the K loop, or - ping-pong
// variants - behind the first LDS-DMA.
#define USDM_GEMM_FETCH_BIAS(tid, ec)                                                                                           \
  do {                                                                                                                      \
    if (bias) {                                                                                                             \
      if (swiglu) {                                                                                                         \
        const int c4 = (tid % (BN / 8)) * 4;                                                                                \
        const int ngate = n0 + (c4 >> 4) * 32 + (c4 & 15);                                                                  \
        if (ngate < a.N) {                                                                                                  \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) { bv[e] = bias[gcol + ngate + e]; bu[e] = bias[gcol + ngate + 16 + e]; } \
        }                                                                                                                   \
      } else if (wide16) {                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) { bv[e] = bias[gcol + n0 + ec8 + e]; bu[e] = bias[gcol + n0 + ec8 + 4 + e]; } \
      } else {                                                                                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                       \
          if (n0 + ec + e < a.N) bv[e] = bias[gcol + n0 + ec + e];                                                          \
      }                                                                                                                     \
    }                                                                                                                       \
    if (tr_mode)                                                                                                            \
      for (int i = tid; i < BN; i += NTH) sb[i] = (bias && n0 + i < a.N) ? bias[gcol + n0 + i] : 0.f;                       \
  } while (0)

template <typename T, int BM, int BN, int NWM, int NWN, bool DMA, int NST = 2, int NCH = 2, bool PP = false>
// NWM x NWN waves over the BM x BN tile; DMA: global->LDS direct with NST LDS stages of NCH 64-byte chunks each;
// PP: the 8-wave ping-pong K loop (two wave groups one barrier apart, see below)
__global__ __launch_bounds__(NWM * NWN * 64) void gemm_kernel(const GemmDev g) {
  const usdm_gemm_args& a = g.a;
  constexpr int NTH = NWM * NWN * 64;
  constexpr int WTM = BM / NWM, WTN = BN / NWN;  // wave tile
  constexpr int ES = sizeof(T);       // element size
  constexpr int CE = 64 / ES;         // elements per chunk
  constexpr int PE = 16 / ES;         // elements per 16-B piece
  constexpr int TM = WTM / 16, TN = WTN / 16;  // 16x16 MFMA tiles per wave
  constexpr int LA = BM * 8 / NTH, LB = BN * 8 / NTH;  // 16-B loads per thread per K-step
  constexpr int RSTEP = NTH / 8;      // rows covered by one load pass
  constexpr int STAGE = (BM + BN) * 64 * NCH; // bytes per LDS stage
  constexpr int CST = BN + 4;                // f32 row stride of the epilogue tile (bank-conflict-free)
  constexpr int CSTT = BM + 16;              // f32 column stride of the TRANSPOSED epilogue tile (outputs whose fast axis is m)
  constexpr int EPI = (BM * CST > BN * CSTT ? BM * CST : BN * CSTT) * 4;
  constexpr int SMEM = (NST * STAGE > EPI) ? NST * STAGE : EPI;
  // ONE LDS object (staging ring / epilogue tile + the per-column bias of the transposed epilogues): a second __shared__ array
  // beside an LDS-DMA ring can make hipcc drain the ring (vmcnt(0)) before every fragment read
  __shared__ __attribute__((aligned(16))) char smem[SMEM + BN * 4 + (PP ? BM * 8 : 0)];   // + (rstd, rstd * mean) per row of the folded LayerNorm (ping-pong tiles)

  const int tid = threadIdx.x;
  TR(0);
#ifdef USDM_GEMM_TRACE
  if (tid == 0 && blockIdx.x < 8192)
    g_gemm_trace[blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
#endif
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int lr = lane & 15, lc = lane >> 4;

  // tile mapping: blocks that share an XCD (blockIdx % 8) get consecutive tiles; N is walked first
  const int ntiles = g.tiles_m * g.tiles_n;
  int tile = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nsplit = a.split_k > 1 ? a.split_k : 1;
  const int ksi = blockIdx.z % nsplit;          // split-K share of this workgroup
  const int z = blockIdx.z / nsplit;
  const int bz = z / a.groups, gz = z - bz * a.groups;

  const char* Abase = (const char*)a.A + (a.a_gstride * gz + a.a_bstride * bz) * ES;
  const char* Wbase = (const char*)a.W + (a.w_gstride * gz) * ES;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, 0x80000000u, 0x00020000);
  auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wbase, 0, 0x80000000u, 0x00020000);

  const int cpt = a.Kc / CE;            // chunks per tap
  const int Q = a.taps * cpt;           // total chunks
  int q_lo = 0, q_hi = Q;                    // chunk range of this split (whole K when split_k <= 1)
  if (nsplit > 1) {
    const int per = ((Q + nsplit - 1) / nsplit + NCH - 1) / NCH * NCH;
    q_lo = ksi * per; q_hi = q_lo + per < Q ? q_lo + per : Q;
    if (q_lo > q_hi) q_lo = q_hi;
  }
  const int nks = (q_hi - q_lo + NCH - 1) / NCH;       // K-steps (NCH chunks each)

  // loader coordinates of this thread
  const int sub = (tid >> 2) & 1, pc = tid & 3, r0 = tid >> 3;
  const unsigned lda_b = (unsigned)(a.lda * ES), ldw_b = (unsigned)(a.ldw * ES);
  const unsigned OOB = 0xFFFFFFF0u;

  u32x4 ra0[LA], rb0[LB], ra1[LA], rb1[LB];  // two register stages: loads run two K-steps ahead of the MFMAs

  auto load_regs = [&](int ks, u32x4* ra, u32x4* rb) {
    const int q = q_lo + 2 * ks + sub;
    const bool qv = q < q_hi;
    const int tap = q / cpt;
    const int cb = (q - tap * cpt) * CE + pc * PE;
    const unsigned colA = (unsigned)((cb + (int64_t)tap * a.a_tap_stride) * ES);
    const unsigned colW = (unsigned)((q * CE + pc * PE) * ES);
    const int roff = a.a_row_off + tap * a.a_row_step;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + r0 + RSTEP * i;
      const int row = m * a.a_row_mul + roff;
      const bool v = qv && ((unsigned)row < (unsigned)a.rowsA);
      const unsigned off = v ? ((unsigned)row * lda_b + colA) : OOB;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int n = n0 + r0 + RSTEP * i;
      const bool v = qv && (n < a.N);
      const unsigned off = v ? ((unsigned)n * ldw_b + colW) : OOB;
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
    }
  };
  auto store_lds = [&](int stage, const u32x4* ra, const u32x4* rb) {
    char* sA = smem + stage * STAGE;
    char* sB = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = r0 + RSTEP * i;
      *(u32x4*)(sA + sub * (BM * 64) + r * 64 + ((pc ^ swz(r)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int r = r0 + RSTEP * i;
      *(u32x4*)(sB + sub * (BN * 64) + r * 64 + ((pc ^ swz(r)) << 4)) = rb[i];
    }
  };

  // epilogue ownership: thread -> 4 fixed output columns; their bias is fetched here, under the K loop
  const int gcol = a.c_gcol * gz;
  const float* bias = ksi == 0 ? a.bias : nullptr;   // bias and residual belong to split 0
  const bool swiglu = a.act == USDM_ACT_SWIGLU;
  constexpr int C4 = BN / 4, RPI = NTH / C4, NIT = BM / RPI;
  const int ec = (tid % C4) * 4, er = tid / C4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f};   // SwiGLU: gate / up; 16-byte-store epilogue: columns 0-3 / 4-7
  // 16-byte stores for the bf16 GELU epilogue of the ping-pong tiles (round 4): a thread owns 8 columns, so that a row of the tile
  // is 16 lanes x 16 B instead of 32 x 8 B - the epilogue of a one-workgroup-per-CU tile is store-ISSUE-bound (3.5 - 5 us per
  // workgroup, profiles/r04_gemm_ablation.txt) and half the store instructions is what shortens it (cdna_hip_programming.md T21)
  constexpr int C8 = BN / 8, RPI8 = NTH / C8, NIT8 = BM / RPI8;
  const int ec8 = (tid % C8) * 8, er8 = tid / C8;
  const bool wide_gelu = USDM_GEMM_WIDE16 && PP && BM % RPI8 == 0 && a.act == USDM_ACT_GELU && !a.round_bf16 && !a.residual && a.C16 && !a.C32 && !a.transpose_out &&
                         a.epi == USDM_EPI_PLAIN && n0 + BN <= a.N && (a.ldc & 7) == 0 && (gcol & 7) == 0 && (((uintptr_t)a.C16) & 15) == 0;
  // ... and for the Q / K tiles of the head-split epilogue (8 consecutive features of one head per lane)
  const bool wide_qk = USDM_GEMM_WIDE16 && PP && BM % RPI8 == 0 && a.epi == USDM_EPI_QKV_HEADS && n0 + BN <= 2 * a.qkv_H * a.qkv_D && (a.qkv_D & 7) == 0 &&
                       ((((uintptr_t)a.qkv_q) | ((uintptr_t)a.qkv_k)) & 15) == 0;
  const bool wide16 = wide_gelu || wide_qk;
  // outputs whose fast axis is m (transpose_out, V^T tiles of the head-split epilogue) go through a transposed LDS tile;
  // their per-column bias is staged in LDS (visible after the K loop's barriers)
  const bool tr_mode = a.transpose_out != 0 || (a.epi == USDM_EPI_QKV_HEADS && n0 >= 2 * a.qkv_H * a.qkv_D);
  float* sb = (float*)(smem + SMEM);
  float* rowst = sb + BN;   // [BM][2], ping-pong tiles only
  if constexpr (!PP) USDM_GEMM_FETCH_BIAS(tid, ec);   // the ping-pong variants fetch it behind their first LDS-DMA instead

  f32x4 acc[TM][TN];
  static_for<TM>([&](auto I) { static_for<TN>([&](auto J) { acc[I][J] = f32x4{0.f, 0.f, 0.f, 0.f}; }); });

  // ---- LDS-DMA loader (buffer_load_dwordx4 ... lds): a wave-instruction fills 64 consecutive 16-B slots of the LDS
  // image = 16 rows x 4 pieces of one sub-chunk plane; the XOR swizzle is applied on the SOURCE side (slot c' of row r
  // receives logical piece c' ^ swz(r)), out-of-range rows/chunks are redirected past the descriptor (zeros).
  // Per-lane offsets of the wave-instructions this wave issues are loop invariants: only the chunk column
  // (+64 B per chunk) and, for multi-tap operands, the row shift change from K-step to K-step.
  constexpr int QA = BM / 16 * NCH, QB = BN / 16 * NCH;  // wave-instructions per operand per K-step
  constexpr int NWV = NWM * NWN;
  constexpr int NIA = (QA + NWV - 1) / NWV, NIB = (QB + NWV - 1) / NWV;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // The LDS-DMA variants serve single-tap operands only (Linear layers: no per-tap row remap; the launcher routes multi-tap
  // operands to the register-staged loaders), so the tap arithmetic - an integer division per instruction - is not even
  // compiled into their K loop.
  constexpr bool simple = DMA;
  unsigned dofA[NIA], dofB[NIB];                         // byte offset of (row, swizzled piece) at chunk 0; >= 2^31 if the row is out of range
  int drA[NIA];                                          // tile row of the A instruction (multi-tap path)
  if constexpr (DMA && !PP) {
    const int lrow = lane >> 2, lp = lane & 3;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int qi = wv + NWV * i;
      const int sb = qi / (BM / 16), rg = qi - sb * (BM / 16);
      const int r = rg * 16 + lrow;
      drA[i] = r;
      const int row = (m0 + r) * a.a_row_mul + a.a_row_off;
      const bool v = (unsigned)row < (unsigned)a.rowsA;
      dofA[i] = (v ? (unsigned)row * lda_b : 0xC0000000u) + (unsigned)(((lp ^ swz(r)) * PE + sb * CE) * ES);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int qi = wv + NWV * i;
      const int sb = qi / (BN / 16), rg = qi - sb * (BN / 16);
      const int r = rg * 16 + lrow;
      const bool v = (n0 + r) < a.N;
      dofB[i] = (v ? (unsigned)(n0 + r) * ldw_b : 0xC0000000u) + (unsigned)(((lp ^ swz(r)) * PE + sb * CE) * ES);
    }
  }
  // issue the wave-instructions [i0, i1) of operand A and [j0, j1) of operand W for K-step ks into `stage`
  auto dma_issue = [&](int stage, int ks, int i0, int i1, int j0, int j1) {
    char* sAst = smem + stage * STAGE;
    char* sBst = sAst + BM * 64 * NCH;
    const unsigned kcol = (unsigned)((q_lo + NCH * ks) * 64);     // NCH chunks of 64 B per K-step
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      if (i < i0 || i >= i1) continue;
      const int qi = wv + NWV * i;
      if (QA % NWV != 0 && qi >= QA) break;
      const int sb = qi / (BM / 16);
      const int q = q_lo + NCH * ks + sb;
      unsigned off;
      if (simple) {
        off = (q < q_hi) ? dofA[i] + kcol : OOB;
      } else {
        const int lp = lane & 3, r = drA[i];
        const int tap = q / cpt;
        const int cb = (q - tap * cpt) * CE + (lp ^ swz(r)) * PE;
        const int row = (m0 + r) * a.a_row_mul + a.a_row_off + tap * a.a_row_step;
        const bool v = (q < q_hi) && ((unsigned)row < (unsigned)a.rowsA);
        off = v ? ((unsigned)row * lda_b + (unsigned)((cb + (int64_t)tap * a.a_tap_stride) * ES)) : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sAst + qi * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      if (i < j0 || i >= j1) continue;
      const int qi = wv + NWV * i;
      if (QB % NWV != 0 && qi >= QB) break;
      const int sb = qi / (BN / 16);
      const unsigned off = (q_lo + NCH * ks + sb < q_hi) ? dofB[i] + kcol : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(sBst + qi * 1024), 16, off, 0, 0, 0);
    }
  };

  // ---- ping-pong variants (PP): LDS image of an operand = [row / 8][8 rows][8 pieces of 16 B]: ONE LDS-DMA instruction moves 8 rows x
  // 128 B, i.e. whole cache lines (half-line pieces cost the texture path twice the lines per byte).  The piece index is XORed with
  // (row >> 1) & 7 on the source side, which makes the ds_read_b128 of a 16x16x32 fragment of either K half conflict-free.
  // A wave issues PFA + PFB whole instructions per K-step and, when BM / 8 is not a multiple of the wave count, one half
  // instruction (32 lanes = 4 rows) of the leftover blocks: NPW = the per-wave, per-step count the vmcnt waits are written in.
  constexpr int PBA = BM / 8, PBB = BN / 8;
  constexpr int PFA = PBA / NWV, PLA = PBA % NWV, PFB = PBB / NWV;
  constexpr int PNF = PFA + PFB, NPW = PNF + (PLA ? 1 : 0);
  static_assert(!PP || (PBB % NWV == 0 && (PLA == 0 || PLA == 4) && PNF >= 2), "ping-pong DMA split");   // odd PNF: the second phase issues one instruction more
  unsigned pofs[PNF + 1];                                // source byte offsets at chunk 0 (>= 2^31: row out of range): A..., W..., leftover A
  unsigned phi = 0;                                      // bit i: instruction i of this lane fetches a piece of the SECOND half of the step
  if constexpr (PP) {
    const int lrow = lane >> 3, lp = lane & 7;
#pragma unroll
    for (int i = 0; i < PFA + (PLA ? 1 : 0); ++i) {
      const int blk = i < PFA ? wv + NWV * i : PFA * NWV + (wv >> 1);
      const int r = blk * 8 + lrow;
      const int row = (m0 + r) * a.a_row_mul + a.a_row_off;
      const bool v = (unsigned)row < (unsigned)a.rowsA;
      const int pc2 = lp ^ ((r >> 1) & 7);
      pofs[i < PFA ? i : PNF] = (v ? (unsigned)row * lda_b : 0xC0000000u) + (unsigned)(pc2 << 4);
      phi |= (unsigned)(pc2 >> 2) << (i < PFA ? i : PNF);
    }
#pragma unroll
    for (int i = 0; i < PFB; ++i) {
      const int r = (wv + NWV * i) * 8 + lrow;
      const bool v = (n0 + r) < a.N;
      const int pc2 = lp ^ ((r >> 1) & 7);
      pofs[PFA + i] = (v ? (unsigned)(n0 + r) * ldw_b : 0xC0000000u) + (unsigned)(pc2 << 4);
      phi |= (unsigned)(pc2 >> 2) << (PFA + i);
    }
  }
  // part 0 / 1: the instructions issued in the first / second phase of a step; 2: all.  chk: the step's second half lies past K
  // chk: 1 = the step's second half lies past K (those lanes read zeros), 2 = the whole step does (a placeholder issue: range-checked
  // away, no memory traffic); only >= 0: just that instruction of the step (PNF = the leftover half instruction)
  // source columns of a step: W is [N][taps * Kc]; A adds a_tap_stride per tap (operands concatenated along K from several sources:
  // a_row_step == 0, Kc % 64 == 0, so a step never straddles two taps and the remap is one scalar division)
  auto pp_cols = [&](int step, unsigned& kcolA, unsigned& kcolW) {
    const int q = q_lo + 2 * step;
    kcolW = (unsigned)(q * 64);
    kcolA = kcolW;
    if (a.taps > 1) {
      const int tap = q / cpt;
      kcolA = (unsigned)(((int64_t)(q - tap * cpt) * CE + (int64_t)tap * a.a_tap_stride) * ES);
    }
  };
  auto pp_issue = [&](int slot, int step, int part, int chk, int only = -1, unsigned kcolA = 0u, unsigned kcolW = 0u) {
    if constexpr (PP) {
      char* sAs = smem + slot * STAGE;
      char* sBs = sAs + BM * 128;
      if (only < 0) pp_cols(step, kcolA, kcolW);      // (single-instruction calls bring the step's columns with them)
#pragma unroll
      for (int i = 0; i < PNF; ++i) {
        if (only >= 0 ? i != only : (part != 2 && (i < PNF / 2) != (part == 0))) continue;
        unsigned off = pofs[i] + (i < PFA ? kcolA : kcolW);
        if (chk == 2 || (chk && ((phi >> i) & 1))) off = OOB;
        if (i < PFA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sAs + (wv + NWV * i) * 1024), 16, off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(sBs + (wv + NWV * (i - PFA)) * 1024), 16, off, 0, 0, 0);
      }
      if (PLA && (only >= 0 ? only == PNF : part != 0)) {
        unsigned off = pofs[PNF] + kcolA;
        if (chk == 2 || (chk && ((phi >> PNF) & 1))) off = OOB;
        if ((lane >> 5) == (wv & 1))
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sAs + (PFA * NWV + (wv >> 1)) * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  auto compute = [&](int stage, auto&& between) {
    const char* sA = smem + stage * STAGE;
    const char* sB = sA + BM * 64 * NCH;
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * WTM + i * 16 + lr;
        fa[i] = *(const u32x4*)(sA + s * (BM * 64) + r * 64 + ((lc ^ swz(r)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * WTN + j * 16 + lr;
        fb[j] = *(const u32x4*)(sB + s * (BN * 64) + r * 64 + ((lc ^ swz(r)) << 4));
      }
      static_for<TM>([&](auto I) {
        static_for<TN>([&](auto J) {
          if constexpr (sizeof(T) == 2) {
            acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[I]), __builtin_bit_cast(bf16x8, fb[J]),
                                                                acc[I][J], 0, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // NB: bit_cast of a vector-element lvalue reads element 0; go through scalars.
              const unsigned ua = fa[I][e], ub = fb[J][e];
              acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua), __uint_as_float(ub), acc[I][J], 0, 0, 0);
            }
          }
        });
      });
      if (s == 0 && NCH > 1) between();
    }
  };

  // software pipeline: LDS stage (ks & 1) is multiplied while step ks+1 is written to the other stage.
  // Small tiles (cheap in registers, short MFMA phase) keep TWO K-steps of loads in flight in two register
  // sets; the 128x128 tile keeps one (a second set would halve its occupancy: measured slower).
  constexpr bool PF2 = (BM * BN <= 128 * 64) || (NTH > 256);
  if constexpr (PP) {
    // ---- 8-wave ping-pong loop for big tiles at one workgroup per CU (cdna_hip_programming.md 5, "8-phase template", re-cut).
    // Waves w and w + 4 share a SIMD; group 1 (waves 4-7) runs ONE barrier behind group 0, so that on every SIMD one wave
    // multiplies (MFMA section, raised priority) while its partner reads fragments and issues LDS-DMA.  A K-step is 64 deep and
    // owns one of the 3 LDS slots; a phase is one 32-deep half of it: TM + TN ds_read_b128, half of the DMA instructions of
    // step s + 2, TM x TN MFMAs, two raw barriers.
    //   RAW  the counted vmcnt that retires step s + 1 sits before the first barrier of phase (s, 1) in BOTH groups; group 0
    //        reads step s + 1 after its second barrier of that phase (= group 1's first), group 1 one barrier later still;
    //   WAR  slot (s + 2) % 3 held step s - 1; every wave retires its fragment reads (lgkmcnt(0)) BEFORE the first barrier of
    //        the phase that issued them, so group 1's last reads of step s - 1 are complete one barrier before group 0's issue.
    static_assert(DMA && NCH == 2 && NST == 3 && NWV == 8 && sizeof(T) == 2, "ping-pong loop: 8 waves, bf16, 3 slots of 64-deep K-steps");
    const int grp = wv >> 2;
    const int nst = nks;                                   // K-steps (NCH = 2 chunks each)
    const bool odd_tail = ((q_hi - q_lo) & 1) != 0;        // the last step has only its first half
    TR(1);
    if (nst > 0) pp_issue(0, 0, 2, nst == 1 && odd_tail);
    if (nst > 1) pp_issue(1, 1, 2, nst == 2 && odd_tail);
    if (nst > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TR(2);
    USDM_GEMM_FETCH_BIAS(tid, ec);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    // fragment byte offsets of this lane inside a slot, for either K half (loop invariants)
    unsigned foA[2][TM], foB[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * WTM + i * 16 + lr;
      foA[0][i] = (r >> 3) * 1024 + (r & 7) * 128 + ((lc ^ ((r >> 1) & 7)) << 4);
      foA[1][i] = foA[0][i] ^ 64;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int r = wn * WTN + j * 16 + lr;
      foB[0][j] = BM * 128 + (r >> 3) * 1024 + (r & 7) * 128 + ((lc ^ ((r >> 1) & 7)) << 4);
      foB[1][j] = foB[0][j] ^ 64;
    }
    u32x4 fa[TM], fb[TN];
    // one phase: half h of the step in `slot`; issue = DMA part h of step sn into slotn; wait_n: < 0 none, else the vmcnt bound
    auto phase = [&](const int slot, const int h, const int slotn, const int sn, const bool issue, const bool chk, const int wait_n) {
      const char* sS = smem + slot * STAGE;
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4*)(sS + foB[h][j]);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4*)(sS + foA[h][i]);
      if (issue) pp_issue(slotn, sn, h, chk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (wait_n > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else if (wait_n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      static_for<TM>([&](auto I) {
        static_for<TN>([&](auto J) {
          acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[I]), __builtin_bit_cast(bf16x8, fb[J]),
                                                              acc[I][J], 0, 0, 0);
        });
      });
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    };
    int s = 0;
    // steady state, unrolled over the ring (slot indices and halves are constants): every step of a round issues a whole step
    // (a round issues steps up to s + 4: it must stop short of a last step that is only half a step, whose DMA needs `chk`)
    for (; s + 4 < nst - (odd_tail ? 1 : 0); s += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        phase(u, 0, (u + 2) % 3, s + u + 2, true, false, -1);
        phase(u, 1, (u + 2) % 3, s + u + 2, true, false, 1);
      }
    }
    // the last steps: runtime slot, no issue past the end, the final step's DMA fully drained
    int sl = s % 3, sln = (s + 2) % 3;
    for (; s < nst; ++s) {
      const bool more = s + 2 < nst;
      const bool chk = odd_tail && s + 2 == nst - 1;
      phase(sl, 0, sln, s + 2, more, chk, -1);
      phase(sl, 1, sln, s + 2, more, chk, more ? 1 : 0);
      if (++sl == 3) sl = 0;
      if (++sln == 3) sln = 0;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __syncthreads();
  } else if constexpr (DMA) {
    // two LDS stages; step ks+1 streams into the idle stage by LDS-DMA while step ks is multiplied
    // NST LDS stages: the DMA of K-step ks+NST-1 is issued while step ks is multiplied; a counted vmcnt leaves the
    // youngest NST-2 steps in flight across the barrier (raw s_barrier: __syncthreads() would drain them).
    constexpr int NPS = NIA + NIB;                         // DMA instructions per wave per K-step
    constexpr int WCNT = (NST - 2) * NPS;                  // allowed outstanding after the per-step wait
    static_assert(QA % NWV == 0 && QB % NWV == 0, "every wave must issue the same number of DMA instructions");
    static_assert(WCNT <= 63, "vmcnt immediate");
    TR(1);
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < nks) dma_issue(p, p, 0, NIA, 0, NIB);
    if (nks >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TR(2);
    int st = 0;
    for (int ks = 0; ks < nks; ++ks) {
      const bool more = ks + NST - 1 < nks;
      int stn = st + NST - 1; if (stn >= NST) stn -= NST;
      if (more) dma_issue(stn, ks + NST - 1, 0, (NIA + 1) / 2, 0, (NIB + 1) / 2);
      compute(st, [&]() { if (more) dma_issue(stn, ks + NST - 1, (NIA + 1) / 2, NIA, (NIB + 1) / 2, NIB); });
      if (NCH == 1 && more) dma_issue(stn, ks + NST - 1, (NIA + 1) / 2, NIA, (NIB + 1) / 2, NIB);
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCNT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (++st == NST) st = 0;
    }
    __syncthreads();
  } else {
  TR(1);
  load_regs(0, ra0, rb0);
  store_lds(0, ra0, rb0);
  if (nks > 1) load_regs(1, ra0, rb0);
  if constexpr (PF2) {
    if (nks > 2) load_regs(2, ra1, rb1);
    __syncthreads();
    TR(2);
    for (int ks = 0; ks < nks; ks += 2) {
      compute(0, []() {});
      if (ks + 1 < nks) store_lds(1, ra0, rb0);
      if (ks + 3 < nks) load_regs(ks + 3, ra0, rb0);
      __syncthreads();
      if (ks + 1 >= nks) break;
      compute(1, []() {});
      if (ks + 2 < nks) store_lds(0, ra1, rb1);
      if (ks + 4 < nks) load_regs(ks + 4, ra1, rb1);
      __syncthreads();
    }
  } else {
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
      compute(ks & 1, []() {});
      if (ks + 1 < nks) store_lds((ks + 1) & 1, ra0, rb0);
      if (ks + 2 < nks) load_regs(ks + 2, ra0, rb0);
      __syncthreads();
    }
  }
  }

  // ------------------------------------------------------------------ epilogue through LDS
  // Every thread owns FIXED output columns (4 consecutive n in the row-major pass), so its bias values were fetched
  // before the K loop (bv/bu above) and residual rows are fetched a batch at a time ahead of their use: a load
  // inside the store loop costs a full memory latency per iteration (measured: 14 us of a 33 us 128x128 tile).
  TR(3);
  if constexpr (PP) {
    // folded LayerNorm: (rstd, rstd * mean) of this tile's rows from the producer's per-tile partial sums.  Requested HERE, after
    // the K loop (in the prologue the loads' wait would sit in front of the first K-step): the latency runs under the accumulator
    // dump below, and the barrier that follows the dump publishes rowst.
    if (a.ln_mode) {
      // per-tile statistics are (sum, M2 about the tile's own mean) of ln_C / ln_nt columns each: merged as Chan et al. do, so the
      // variance never goes through sum(x^2) - mean^2 (which loses |mean| / sigma squared in relative accuracy)
      const float ncol = (float)(a.ln_C / a.ln_nt), inv_ncol = 1.0f / ncol;
      bool unsafe = false;
      for (int i = tid; i < BM; i += NTH) {
        const int m = m0 + i;
        float s1 = 0.f, m2 = 0.f;
        float mean = 0.f;
        if (m < a.M) {
          const float2* sp = (const float2*)a.ln_stats + ((int64_t)bz * a.c_bstride + m) * a.ln_nt;
          for (int t = 0; t < a.ln_nt; ++t) s1 += sp[t].x;
          mean = s1 / (float)a.ln_C;
          for (int t = 0; t < a.ln_nt; ++t) { const float2 v = sp[t]; const float d = v.x * inv_ncol - mean; m2 += v.y + ncol * d * d; }
        }
        const float var = m2 / (float)a.ln_C;
        const float rstd = rsqrtf(var + a.ln_eps);
        rowst[2 * i] = rstd; rowst[2 * i + 1] = rstd * mean;
        // ln_mode 1 multiplies rows that were rounded to bf16 BEFORE centring: the operand's rounding noise relative to the
        // normalised signal grows with |mean| / sigma.  Rows beyond the caller's bound are reported, never silently accepted.
        if (a.ln_guard && m < a.M && fabsf(mean) * rstd > a.ln_guard_ratio) unsafe = true;
      }
      if (a.ln_guard && tn == 0 && unsafe) atomicOr(a.ln_guard, 1);
    }
  }
  float* ct = (float*)smem;  // [BM][CST] f32, or [BN][CSTT] in transposed mode
  const void* resid = ksi == 0 ? a.residual : nullptr;                                     // split-K: split 0 owns bias + residual
  float* C32p = a.C32 ? (float*)a.C32 + (int64_t)ksi * a.c_split_stride : nullptr;
  const bool rbf = a.round_bf16 != 0;
  const bool is_qkv = a.epi == USDM_EPI_QKV_HEADS;

  constexpr bool TR_OK = NTH % (BM / 4) == 0 && BN % (NTH / (BM / 4)) == 0;   // tile geometries the transposed store loops cover
  if (tr_mode) {
    if constexpr (!TR_OK) return;   // (the launcher routes transposed outputs to the other tiles: usdm_gemm, sel == 13)
    // ---- transposed mode: the accumulator fragment of a lane is 4 consecutive rows of one column, i.e. one float4 of the
    // transposed tile; the store loop then reads float4s along m without bank conflicts (a strided read of the row-major
    // tile was 8-way conflicted: 10 us per 128x128 V tile)
    static_for<TM>([&](auto I) {
      static_for<TN>([&](auto J) {
        const int col = wn * WTN + J * 16 + lr;
        const int row = wm * WTM + I * 16 + lc * 4;
        *(float4*)(ct + col * CSTT + row) = make_float4(acc[I][J][0], acc[I][J][1], acc[I][J][2], acc[I][J][3]);
      });
    });
    __syncthreads();
    TR(4);
    if (is_qkv) {
      // V^T[b][h][d][s]: lane = 4 consecutive tokens of one feature, a wave = 256 tokens of it -> dense 8-byte-per-lane stores
      // (4 bytes per lane took 5.5 us per 256x128 tile against 2.8 us for the Q / K tiles).  Lanes whose 4 tokens straddle a
      // sequence end, the matrix end or an odd position store token by token.
      constexpr int R4V = BM / 4, CPI4 = NTH / R4V, NITV = BN / CPI4;
      const int r4 = (tid % R4V) * 4, c0v = tid / R4V;
      const int m = m0 + r4;
      const int mv = (a.M - m) < 4 ? (a.M - m) : 4;
      if (mv > 0) {
        const int HD2 = 2 * a.qkv_H * a.qkv_D;
        const int b0 = m / a.qkv_S, s0 = m - b0 * a.qkv_S;
        const bool quad = mv == 4 && (s0 + 3 < a.qkv_S) && ((s0 & 1) == 0);
        int64_t tok[4];                                   // element offset of each token's (batch, position) inside V^T
        {
          int bb = b0, ss = s0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            tok[e] = (int64_t)bb * a.qkv_H * a.qkv_D * a.qkv_Spad + ss;
            if (++ss == a.qkv_S) { ss = 0; ++bb; }
          }
        }
        float4 cv = *(const float4*)(ct + c0v * CSTT + r4);
        int vh = (n0 + c0v - HD2) / a.qkv_D, vd = (n0 + c0v - HD2) - vh * a.qkv_D;   // head / feature of this thread's column, advanced
#pragma unroll 2                                                                       // incrementally (no division per column)
        for (int it = 0; it < NITV; ++it) {
          const int c = c0v + it * CPI4, n = n0 + c;
          if (n >= a.N) break;
          float v[4] = {cv.x, cv.y, cv.z, cv.w};
          if (it + 1 < NITV) cv = *(const float4*)(ct + (c + CPI4) * CSTT + r4);
          const float bc = sb[c];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = a.alpha * v[e] + bc;
            if (rbf) v[e] = round_bf(v[e]);
          }
          const int h = vh, d = vd;
          vd += CPI4;
          while (vd >= a.qkv_D) { vd -= a.qkv_D; ++vh; }
          bf16_t* vfeat = (bf16_t*)a.qkv_v + ((int64_t)h * a.qkv_D + d) * a.qkv_Spad;
          if (quad) {
            uint2 p; p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]);
            st_out((uint2*)(vfeat + tok[0]), p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < mv) vfeat[tok[e]] = f2bf(v[e]);
          }
        }
      }
#ifdef USDM_GEMM_TRACE
      TR(6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TR(5);
#endif
      return;
    }
    constexpr int R4 = BM / 4, CPI = NTH / R4, NIT2 = BN / CPI;
    const int r4 = (tid % R4) * 4, c0 = tid / R4;
    const int m = m0 + r4;
    const int mv = (a.M - m) < 4 ? (a.M - m) : 4;
    if (mv > 0) {
      const int64_t row = ((int64_t)bz * a.c_bstride + m) * a.c_row_mul + a.c_row_off;
      float4 cv = *(const float4*)(ct + c0 * CSTT + r4);
#pragma unroll 2
      for (int it = 0; it < NIT2; ++it) {
        const int c = c0 + it * CPI, n = n0 + c;
        if (n >= a.N) break;
        float v[4] = {cv.x, cv.y, cv.z, cv.w};
        if (it + 1 < NIT2) cv = *(const float4*)(ct + (c + CPI) * CSTT + r4);
        const float bc = sb[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = a.alpha * v[e] + bc;
          if (rbf) x = round_bf(x);
          if (a.act == USDM_ACT_GELU) x = gelu_erf(x);
          else if (a.act == USDM_ACT_TANH) x = tanhf(x);
          else if (a.act == USDM_ACT_LOGCLAMP) x = logf(fmaxf(x, 1e-5f));
          v[e] = x;
        }
        for (int e = 0; e < mv; ++e) {
          const int64_t rw = row + (int64_t)e * a.c_row_mul;
          float x = v[e];
          if (resid) {
            const int64_t ri = rw * a.ldr + gcol + n;
            x += (a.res_dtype == USDM_F32) ? ((const float*)resid)[ri] : bf2f(((const bf16_t*)resid)[ri]);
            if (rbf) x = round_bf(x);
          }
          const int64_t oi = (int64_t)(gcol + n) * a.ldc + rw;
          if (C32p) C32p[oi] = x;
          if (a.C16) ((bf16_t*)a.C16)[oi] = f2bf(x);
        }
      }
    }
#ifdef USDM_GEMM_TRACE
    TR(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR(5);
#endif
    return;
  }

  static_for<TM>([&](auto I) {
    static_for<TN>([&](auto J) {
      const int col = wn * WTN + J * 16 + lr;
      const int row = wm * WTM + I * 16 + lc * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) ct[(row + e) * CST + col] = acc[I][J][e];
    });
  });
  __syncthreads();
  TR(4);

  if (swiglu) {
    // column tiles (2p, 2p+1) of 16 hold gate / up of the same 16 output features
    constexpr int OC4 = BN / 8, RPS = NTH / OC4, NIS = BM / RPS;
    const int c4 = (tid % OC4) * 4, rr = tid / OC4;
    const int cg = (c4 >> 4) * 32 + (c4 & 15);   // gate column inside the tile
    const int ngate = n0 + cg;
    if (ngate < a.N) {
      const int nout = (n0 >> 1) + c4;
#pragma unroll 2
      for (int it = 0; it < NIS; ++it) {
        const int r = rr + it * RPS, m = m0 + r;
        if (m >= a.M) break;
        const float4 gv = *(const float4*)(ct + r * CST + cg), uv = *(const float4*)(ct + r * CST + cg + 16);
        const float gt[4] = {gv.x, gv.y, gv.z, gv.w}, up[4] = {uv.x, uv.y, uv.z, uv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = silu_mul(a.alpha * gt[e] + bv[e], a.alpha * up[e] + bu[e], rbf);
        const int64_t row = ((int64_t)bz * a.c_bstride + m) * a.c_row_mul + a.c_row_off;
        const int64_t oi = row * a.ldc + (gcol >> 1) + nout;
        if (C32p) st_out((float4*)(C32p + oi), make_float4(o[0], o[1], o[2], o[3]));
        if (a.C16) { uint2 p; p.x = pack_bf2(o[0], o[1]); p.y = pack_bf2(o[2], o[3]); st_out((uint2*)((bf16_t*)a.C16 + oi), p); }
      }
    }
    return;
  }

  // ---- row-major outputs (and Q / K tiles of the head-split epilogue): thread = 4 fixed columns, rows er, er + RPI, ...
  const int n = n0 + ec;
  const int nv = (a.N - n) < 4 ? (a.N - n) : 4;   // <= 0: this thread's columns are outside the matrix
  const bool vec_ok = is_qkv ? true : (((a.ldc | gcol) & 3) == 0 && (!resid || (a.ldr & 3) == 0));
  auto act_fn = [&](float x) -> float {
    if (a.act == USDM_ACT_GELU) return gelu_erf(x);
    if (a.act == USDM_ACT_TANH) return tanhf(x);
    if (a.act == USDM_ACT_LOGCLAMP) return logf(fmaxf(x, 1e-5f));
    return x;
  };
  if (is_qkv && wide_qk) {
    const int HD = a.qkv_H * a.qkv_D;
    const int n8 = n0 + ec8;
    const int part = n8 / HD;                         // 0: Q, 1: K (V tiles took the transposed path)
    const int hn = n8 - part * HD, qh = hn / a.qkv_D, qd = hn - qh * a.qkv_D;
    bf16_t* base = (bf16_t*)(part == 0 ? a.qkv_q : a.qkv_k);
    float4 ca = *(const float4*)(ct + er8 * CST + ec8), cb = *(const float4*)(ct + er8 * CST + ec8 + 4);
    int b = (m0 + er8) / a.qkv_S, sq = (m0 + er8) - b * a.qkv_S;    // batch / position of the row, advanced incrementally
#pragma unroll 2
    for (int it = 0; it < NIT8; ++it) {
      const int r = er8 + it * RPI8, m = m0 + r;
      if (m >= a.M) break;
      float v[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
      if (it + 1 < NIT8) { ca = *(const float4*)(ct + (r + RPI8) * CST + ec8); cb = *(const float4*)(ct + (r + RPI8) * CST + ec8 + 4); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = a.alpha * v[e] + (e < 4 ? bv[e] : bu[e - 4]);
        if (rbf) v[e] = round_bf(v[e]);
      }
      uint4 p; p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]); p.z = pack_bf2(v[4], v[5]); p.w = pack_bf2(v[6], v[7]);
      st_out((uint4*)(base + (((int64_t)b * a.qkv_H + qh) * a.qkv_Spad + sq) * a.qkv_D + qd), p);
      sq += RPI8;
      while (sq >= a.qkv_S) { sq -= a.qkv_S; ++b; }
    }
  } else if (is_qkv) {
    if (nv > 0) {
      const int HD = a.qkv_H * a.qkv_D;
      const int part = n / HD;                        // 0: Q, 1: K (V tiles took the transposed path)
      const int hn = n - part * HD, qh = hn / a.qkv_D, qd = hn - qh * a.qkv_D;
      bf16_t* base = (bf16_t*)(part == 0 ? a.qkv_q : a.qkv_k);
      float4 cv = *(const float4*)(ct + er * CST + ec);
      int b = (m0 + er) / a.qkv_S, sq = (m0 + er) - b * a.qkv_S;    // batch / position of the row, advanced incrementally
#pragma unroll 2
      for (int it = 0; it < NIT; ++it) {
        const int r = er + it * RPI, m = m0 + r;
        if (m >= a.M) break;
        float v[4] = {cv.x, cv.y, cv.z, cv.w};
        if (it + 1 < NIT) cv = *(const float4*)(ct + (r + RPI) * CST + ec);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = a.alpha * v[e] + bv[e];
          if (rbf) v[e] = round_bf(v[e]);
        }
        uint2 p; p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]);
        st_out((uint2*)(base + (((int64_t)b * a.qkv_H + qh) * a.qkv_Spad + sq) * a.qkv_D + qd), p);   // D % 4 == 0, N % 4 == 0 (host-checked)
        sq += RPI;
        while (sq >= a.qkv_S) { sq -= a.qkv_S; ++b; }
      }
    }
  } else if (wide_gelu) {
    const int64_t rstep = (int64_t)RPI8 * a.c_row_mul;
    int64_t row = ((int64_t)bz * a.c_bstride + m0 + er8) * a.c_row_mul + a.c_row_off;
    const f32x2_t b01 = {bv[0], bv[1]}, b23 = {bv[2], bv[3]}, b45 = {bu[0], bu[1]}, b67 = {bu[2], bu[3]};
    f32x2_t lc01 = {0.f, 0.f}, lc23 = {0.f, 0.f}, lc45 = {0.f, 0.f}, lc67 = {0.f, 0.f};
    if (a.ln_mode == 1) {
      const float4 c0 = *(const float4*)(a.ln_c + gcol + n0 + ec8), c1 = *(const float4*)(a.ln_c + gcol + n0 + ec8 + 4);
      lc01 = f32x2_t{c0.x, c0.y}; lc23 = f32x2_t{c0.z, c0.w}; lc45 = f32x2_t{c1.x, c1.y}; lc67 = f32x2_t{c1.z, c1.w};
    }
    float4 ca = *(const float4*)(ct + er8 * CST + ec8), cb = *(const float4*)(ct + er8 * CST + ec8 + 4);
#pragma unroll 2
    for (int it = 0; it < NIT8; ++it, row += rstep) {
      const int r = er8 + it * RPI8;
      if (m0 + r >= a.M) break;
      const f32x2_t v01 = {ca.x, ca.y}, v23 = {ca.z, ca.w}, v45 = {cb.x, cb.y}, v67 = {cb.z, cb.w};
      if (it + 1 < NIT8) { ca = *(const float4*)(ct + (r + RPI8) * CST + ec8); cb = *(const float4*)(ct + (r + RPI8) * CST + ec8 + 4); }
      f32x2_t g01, g23, g45, g67;
      if (a.ln_mode == 1) {          // folded LayerNorm: rstd * acc - rstd * mean * c[n] + bias'[n]
        const float2 rs = *(const float2*)(rowst + 2 * r);
        const f32x2_t r2 = {rs.x, rs.x}, nrm2 = {-rs.y, -rs.y};
        g01 = gelu_erf2(fma2(v01, r2, fma2(lc01, nrm2, b01))); g23 = gelu_erf2(fma2(v23, r2, fma2(lc23, nrm2, b23)));
        g45 = gelu_erf2(fma2(v45, r2, fma2(lc45, nrm2, b45))); g67 = gelu_erf2(fma2(v67, r2, fma2(lc67, nrm2, b67)));
      } else {
        g01 = gelu_erf2(v01 * a.alpha + b01); g23 = gelu_erf2(v23 * a.alpha + b23);
        g45 = gelu_erf2(v45 * a.alpha + b45); g67 = gelu_erf2(v67 * a.alpha + b67);
      }
      uint4 p; p.x = pack_bf2v(g01); p.y = pack_bf2v(g23); p.z = pack_bf2v(g45); p.w = pack_bf2v(g67);
      st_out((uint4*)((bf16_t*)a.C16 + row * a.ldc + gcol + n0 + ec8), p);
    }
  } else if (PP && nv == 4 && vec_ok && a.act == USDM_ACT_GELU && !rbf && !resid && a.C16 && !C32p) {
    // the feed-forward GELU epilogue of the one-workgroup-per-CU tiles: nothing overlaps it there, so it is written for VALU
    // throughput (two values per instruction, no per-element dispatch)
    const int64_t rstep = (int64_t)RPI * a.c_row_mul;
    int64_t row = ((int64_t)bz * a.c_bstride + m0 + er) * a.c_row_mul + a.c_row_off;
    const f32x2_t b01 = {bv[0], bv[1]}, b23 = {bv[2], bv[3]};
    f32x2_t lc01 = {0.f, 0.f}, lc23 = {0.f, 0.f};
    if (a.ln_mode == 1) { const float4 c4 = *(const float4*)(a.ln_c + gcol + n); lc01 = f32x2_t{c4.x, c4.y}; lc23 = f32x2_t{c4.z, c4.w}; }
    float4 cv = *(const float4*)(ct + er * CST + ec);
#pragma unroll 2
    for (int it = 0; it < NIT; ++it, row += rstep) {
      const int r = er + it * RPI;
      if (m0 + r >= a.M) break;
      const f32x2_t v01 = {cv.x, cv.y}, v23 = {cv.z, cv.w};
      if (it + 1 < NIT) cv = *(const float4*)(ct + (r + RPI) * CST + ec);
      f32x2_t g01, g23;
      if (a.ln_mode == 1) {          // folded LayerNorm: rstd * acc - rstd * mean * c[n] + bias'[n]
        const float2 rs = *(const float2*)(rowst + 2 * r);
        const f32x2_t r2 = {rs.x, rs.x}, nrm2 = {-rs.y, -rs.y};
        g01 = gelu_erf2(fma2(v01, r2, fma2(lc01, nrm2, b01)));
        g23 = gelu_erf2(fma2(v23, r2, fma2(lc23, nrm2, b23)));
      } else {
        g01 = gelu_erf2(v01 * a.alpha + b01); g23 = gelu_erf2(v23 * a.alpha + b23);
      }
      uint2 p; p.x = pack_bf2v(g01); p.y = pack_bf2v(g23);
      st_out((uint2*)((bf16_t*)a.C16 + row * a.ldc + gcol + n), p);
    }
  } else if (nv == 4 && vec_ok && a.act != USDM_ACT_NONE) {
    // activation epilogues: ROLLED loop so the transcendental code exists once (an unrolled epilogue grew the kernel to
    // 88 KB and ran out of the instruction cache); the next row's LDS read is issued under this row's math
    const bool has_res = resid != nullptr;
    const int64_t rstep = (int64_t)RPI * a.c_row_mul;
    int64_t row = ((int64_t)bz * a.c_bstride + m0 + er) * a.c_row_mul + a.c_row_off;
    float4 cv = *(const float4*)(ct + er * CST + ec);
#pragma unroll 2
    for (int it = 0; it < NIT; ++it, row += rstep) {
      const int r = er + it * RPI;
      if (m0 + r >= a.M) break;
      float v[4] = {cv.x, cv.y, cv.z, cv.w};
      if (it + 1 < NIT) cv = *(const float4*)(ct + (r + RPI) * CST + ec);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = a.alpha * v[e] + bv[e];
        if (rbf) x = round_bf(x);
        v[e] = act_fn(x);
      }
      if (has_res) {
        const int64_t ri = row * a.ldr + gcol + n;
        if (a.res_dtype == USDM_F32) {
          const float4 rv = *(const float4*)((const float*)resid + ri);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        } else {
          const uint2 rv = *(const uint2*)((const bf16_t*)resid + ri);
          v[0] += bf2f(rv.x & 0xffff); v[1] += bf2f(rv.x >> 16); v[2] += bf2f(rv.y & 0xffff); v[3] += bf2f(rv.y >> 16);
        }
        if (rbf) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);
        }
      }
      const int64_t oi = row * a.ldc + gcol + n;
      if (C32p) st_out((float4*)(C32p + oi), make_float4(v[0], v[1], v[2], v[3]));
      if (a.C16) { uint2 p; p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]); st_out((uint2*)((bf16_t*)a.C16 + oi), p); }
    }
  } else if (nv == 4 && vec_ok) {
    // no activation: batches of NB rows, the residual rows of a batch all in flight before their first use (a load inside
    // the store loop costs a full memory latency per iteration: measured 14 us of a 33 us 128x128 tile)
    constexpr int NB = NIT < 8 ? NIT : (NIT % 8 == 0 ? 8 : (NIT % 6 == 0 ? 6 : (NIT % 4 == 0 ? 4 : 1)));
    static_assert(NIT % NB == 0, "epilogue batches");
    const bool has_res = resid != nullptr;
    const int64_t rstep = (int64_t)RPI * a.c_row_mul;
    int64_t row = ((int64_t)bz * a.c_bstride + m0 + er) * a.c_row_mul + a.c_row_off;
    // folded LayerNorm, residual form (ln_mode 2): the residual rows are un-normalised x; LN(x) = x * (rstd gamma) - (rstd mean) gamma + beta
    float lg[4] = {0.f, 0.f, 0.f, 0.f}, lb[4] = {0.f, 0.f, 0.f, 0.f};
    const bool ln_res = PP && a.ln_mode == 2 && has_res;
    if (ln_res) {
      const float4 g4 = *(const float4*)(a.ln_gamma + gcol + n), b4 = *(const float4*)(a.ln_beta + gcol + n);
      lg[0] = g4.x; lg[1] = g4.y; lg[2] = g4.z; lg[3] = g4.w; lb[0] = b4.x; lb[1] = b4.y; lb[2] = b4.z; lb[3] = b4.w;
    }
    float* stp = (PP && (C4 == 32 || C4 == 16) && a.stats_out) ? a.stats_out : nullptr;      // producer form: per-row partial sums of this tile's columns
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += NB) {
      float4 rres[NB];
      if (has_res) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int m = m0 + er + (it0 + u) * RPI;
          rres[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < a.M) {
            const int64_t ri = (row + u * rstep) * a.ldr + gcol + n;
            if (a.res_dtype == USDM_F32) {
              rres[u] = *(const float4*)((const float*)resid + ri);
            } else {
              const uint2 rv = *(const uint2*)((const bf16_t*)resid + ri);
              rres[u] = make_float4(bf2f(rv.x & 0xffff), bf2f(rv.x >> 16), bf2f(rv.y & 0xffff), bf2f(rv.y >> 16));
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int r = er + (it0 + u) * RPI, m = m0 + r;
        if (m >= a.M) break;
        const float4 cv = *(const float4*)(ct + r * CST + ec);
        float v[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = a.alpha * v[e] + bv[e];
          if (rbf) v[e] = round_bf(v[e]);
        }
        if (has_res) {
          if (ln_res) {
            const float2 rs = *(const float2*)(rowst + 2 * r);
            const float xr[4] = {rres[u].x, rres[u].y, rres[u].z, rres[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += fmaf(xr[e], rs.x * lg[e], fmaf(-rs.y, lg[e], lb[e]));
          } else {
            v[0] += rres[u].x; v[1] += rres[u].y; v[2] += rres[u].z; v[3] += rres[u].w;
          }
          if (rbf) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);
          }
        }
        if (stp) {   // the C4 threads that share this row are one half (BN = 128) or one DPP row (BN = 64) of a wave: reduce, one 8-byte store per row and tile
          // (sum, M2 about the tile mean) of this tile's BN columns of the row: two dependent lane reductions instead of one,
          // but no sum of raw squares (see the consumer)
          const float t1 = (v[0] + v[1]) + (v[2] + v[3]);
          const float s1 = C4 == 32 ? half_sum(t1) : row16_sum(t1);
          const float mt = s1 * (1.0f / (float)BN);
          const float d0 = v[0] - mt, d1 = v[1] - mt, d2 = v[2] - mt, d3 = v[3] - mt;
          const float t2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          const float s2 = C4 == 32 ? half_sum(t2) : row16_sum(t2);
          if ((tid & (C4 - 1)) == 0) *(float2*)(stp + (((row + u * rstep) * g.tiles_n + tn) << 1)) = make_float2(s1, s2);
        }
        const int64_t oi = (row + u * rstep) * a.ldc + gcol + n;
        if (C32p) st_out((float4*)(C32p + oi), make_float4(v[0], v[1], v[2], v[3]));
        if (a.C16) { uint2 p; p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]); st_out((uint2*)((bf16_t*)a.C16 + oi), p); }
      }
      row += NB * rstep;
    }
  } else if (nv > 0) {
    // ragged / unaligned columns (N or ldc not a multiple of 4): scalar accesses, rolled
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
      const int r = er + it * RPI, m = m0 + r;
      if (m >= a.M) break;
      const int64_t row = ((int64_t)bz * a.c_bstride + m) * a.c_row_mul + a.c_row_off;
      for (int e = 0; e < nv; ++e) {
        float x = a.alpha * ct[r * CST + ec + e] + bv[e];
        if (rbf) x = round_bf(x);
        x = act_fn(x);
        if (resid) {
          const int64_t ri = row * a.ldr + gcol + n + e;
          x += (a.res_dtype == USDM_F32) ? ((const float*)resid)[ri] : bf2f(((const bf16_t*)resid)[ri]);
          if (rbf) x = round_bf(x);
        }
        const int64_t oi = row * a.ldc + gcol + n + e;
        if (C32p) C32p[oi] = x;
        if (a.C16) ((bf16_t*)a.C16)[oi] = f2bf(x);
      }
    }
  }
#ifdef USDM_GEMM_TRACE
  TR(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TR(5);
#endif
}

#undef USDM_GEMM_FETCH_BIAS

template <typename T, int BM, int BN, int NWM = 2, int NWN = 2, bool DMA = false, int NST = 2, int NCH = 2, bool PP = false>
int launch(const usdm_gemm_args& a, hipStream_t st) {
  GemmDev g;
  g.a = a;
  g.tiles_m = cdiv(a.M, BM);
  g.tiles_n = cdiv(a.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, 1, a.groups * a.batch * (a.split_k > 1 ? a.split_k : 1));
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, NWM, NWN, DMA, NST, NCH, PP>), grid, dim3(NWM * NWN * 64), 0, st, g);
  USDM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef USDM_GEMM_TRACE
extern "C" int usdm_dbg_gemm_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * n);
}
#endif

// validation + tile choice (+ launch unless tile_out: usdm_gemm_tile_for)
static int gemm_impl(const usdm_gemm_args* pa, usdm_stream_t stream, int* tile_out) {
  USDM_CHECK_ARG(pa != nullptr, "usdm_gemm: null args");
  usdm_gemm_args a = *pa;
  if (a.groups <= 0) a.groups = 1;
  if (a.batch <= 0) a.batch = 1;
  if (a.taps <= 0) a.taps = 1;
  if (a.c_row_mul == 0) a.c_row_mul = 1;
  if (a.a_row_mul == 0) a.a_row_mul = 1;
  const int es = a.dtype == USDM_BF16 ? 2 : 4;
  const int ce = 64 / es;
  USDM_CHECK_ARG(a.dtype == USDM_BF16 || a.dtype == USDM_F32, "usdm_gemm: bad dtype %d", a.dtype);
  USDM_CHECK_ARG(a.M > 0 && a.N > 0, "usdm_gemm: bad M/N %d %d", a.M, a.N);
  USDM_CHECK_ARG(a.Kc > 0 && a.Kc % ce == 0, "usdm_gemm: Kc=%d must be a multiple of %d", a.Kc, ce);
  USDM_CHECK_ARG(a.A && a.W, "usdm_gemm: null operand");
  USDM_CHECK_ARG(a.ldw >= (int64_t)a.taps * a.Kc, "usdm_gemm: ldw too small");
  USDM_CHECK_ARG(((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.W % 16) == 0, "usdm_gemm: operands must be 16-B aligned");
  USDM_CHECK_ARG((a.lda * es) % 16 == 0 && (a.ldw * es) % 16 == 0 && (a.a_gstride * es) % 16 == 0 &&
                     (a.w_gstride * es) % 16 == 0 && (a.a_bstride * es) % 16 == 0 && (a.a_tap_stride * es) % 16 == 0,
                 "usdm_gemm: strides must keep 16-B alignment");
  USDM_CHECK_ARG(a.a_tap_stride >= 0, "usdm_gemm: a_tap_stride must be >= 0");
  // all in-range byte offsets must stay below the 2 GiB descriptor range
  const int64_t amax = ((int64_t)a.rowsA * a.lda + (int64_t)(a.taps - 1) * a.a_tap_stride + a.Kc) * es;
  const int64_t wmax = ((int64_t)a.N * a.ldw) * es;
  USDM_CHECK_ARG(amax < 0x7FFFFF00ll && wmax < 0x7FFFFF00ll, "usdm_gemm: operand exceeds 2 GiB addressing window");
  USDM_CHECK_ARG(a.C32 || a.C16 || a.epi == USDM_EPI_QKV_HEADS, "usdm_gemm: no output");
  USDM_CHECK_ARG(a.split_k <= 1 || (a.split_k <= 16 && a.taps == 1 && a.C32 && !a.C16 && a.act == USDM_ACT_NONE && !a.round_bf16 &&
                                    !a.transpose_out && a.epi == USDM_EPI_PLAIN && a.c_split_stride > 0),
                 "usdm_gemm: split_k needs a single-tap GEMM with a plain f32 output (no activation / bf16 rounding / transpose)");
  if (a.act == USDM_ACT_SWIGLU)
    USDM_CHECK_ARG(a.N % 32 == 0 && !a.transpose_out && !a.residual && a.ldc % 4 == 0 && a.c_gcol % 8 == 0 && a.epi == USDM_EPI_PLAIN,
                   "usdm_gemm: swiglu needs N%%32==0, ldc%%4==0, row-major output, no residual");
  if (a.epi == USDM_EPI_QKV_HEADS)
    USDM_CHECK_ARG(a.qkv_q && a.qkv_k && a.qkv_v && a.N == 3 * a.qkv_H * a.qkv_D && (a.qkv_H * a.qkv_D) % 64 == 0 && a.qkv_S > 0 && a.qkv_Spad >= a.qkv_S &&
                       a.qkv_D % 4 == 0 && !a.transpose_out && !a.residual && a.groups == 1 && a.batch == 1,
                   "usdm_gemm: bad qkv epilogue args");
  hipStream_t st = (hipStream_t)stream;
  // tile selection, measured on MI355X (profiles/r01_gemm_tiles.txt, profiles/r01_gemm_ablation.txt)
  const int64_t z = (int64_t)a.groups * a.batch * (a.split_k > 1 ? a.split_k : 1);
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * z;
  const int64_t t12864 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 64) * z;
  int sel;  // 0-2: register-staged 128x128 / 128x64 / 64x64; 4-6: the same tiles with 2-stage LDS-DMA; 7-11: deeper DMA pipelines; 12-13: ping-pong
  // Single-tap GEMMs (Linear layers): chosen from tools/vb_gemm_bench.py (the Voicebox layer's GEMMs over cold weights) and
  // tools/bench_gemm_tiles.py, see profiles/r01_gemm_ablation.txt.
  constexpr int heur = 1;
  // Big single-tap bf16 GEMMs: the 8-wave ping-pong tiles (256x128, or 288x128 when that saves a round of workgroups) at one
  // workgroup per CU, 1.2-2x the 128x128 tile on the Voicebox and LLM-prefill shapes (profiles/r02_gemm_ablation.txt section 4)
  const int64_t t12 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 128) * z, t13 = (int64_t)cdiv(a.M, 288) * cdiv(a.N, 128) * z;
  const bool pp_taps = a.taps == 1 || (a.a_row_step == 0 && a.Kc % 64 == 0);      // single tap, or sources concatenated along K
  const bool pp_ok = heur == 1 && a.dtype == USDM_BF16 && pp_taps && a.N > 64 && t12 >= 96 && a.Kc / (a.split_k > 1 ? a.split_k : 1) >= 256 &&
                     (int64_t)cdiv(a.M, 256) * 256 * 2 <= (int64_t)a.M * 3;      // (= rows256_ok below)
  // ... and its 128x128 form where the big tiles would leave half the CUs idle (96-256 tiles of 128x128, one per CU)
  const int64_t t14 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * z;
  // (also where 256-row tiles would be mostly padding: the 33 - 100 row prefills of a reused prefix stream gate/up's 235 MB through
  // 224 such tiles at 4.1 TB/s, 57 us against 67 - 90 on the 128x64 tile; profiles/r04_prefill_small.log)
  const bool rows256_ok = (int64_t)cdiv(a.M, 256) * 256 * 2 <= (int64_t)a.M * 3;
  const bool pp_small = heur == 1 && a.dtype == USDM_BF16 && pp_taps && a.N > 64 &&   // (K-concatenated sources included: the skip Linear 31.1 -> 25.7 us, r03_vb_ablation.txt 8)
                        (t12 < 128 || !rows256_ok) && t14 >= 96 && t14 <= 256 &&
                        a.Kc / (a.split_k > 1 ? a.split_k : 1) >= 256;
  if (pp_small) sel = 14;
  else if (pp_ok) {
    const int64_t c12 = cdiv(t12, 256) * 256, c13 = cdiv(t13, 256) * 288;     // rounds x rows per tile
    sel = (!a.transpose_out && a.epi == USDM_EPI_PLAIN && c13 < c12) ? 13 : 12;
  }
  else if (a.N <= 64) sel = (a.dtype != USDM_F32 && cdiv(a.M, 128) * z >= 448) ? 1 : 2;   // (f32: the 64x64 tile is 7 - 17 % faster on BigVGAN's 24 / 48-channel stages, profiles/r04_bigvgan_conv_tiles.log)
  else if (heur == 1 && a.taps == 1) {
    if (a.split_k > 1 && t128 >= 224 && t128 <= 512) sel = 4;          // split-K partials filling one round of the big tile
    else if (t128 >= 640) sel = 4;                                     // many rounds of the big tile
    else if (a.Kc >= 8192 && t128 >= 128 && t128 <= 256) sel = 11;     // one deep-K tile per CU: 3-stage DMA pipeline
    else if (a.Kc >= 2048 && t12864 >= 256) sel = 10;                  // deep K, few tiles: 128x64, 3 stages
    else if (t128 >= 400 && t128 <= 512) sel = 4;                      // exactly one round of 2 workgroups per CU
    else if (a.N >= 2048 && t12864 >= 768) sel = 6;
    // few tiles: weight streaming through a handful of CUs is latency-bound - the 64x64 tile with FOUR DMA stages in flight (short
    // 7B prefills: qkv / o 52 -> 27 us, down 169 -> 84 us; f32 shapes measure the same on 2 and 4 stages and keep the 2-stage tile)
    else sel = (a.dtype == USDM_BF16 && a.Kc >= 1024) ? 8 : 5;
  }
  // multi-tap operands (convolutions): register-staged loaders (the DMA path recomputes the tap row shift per K-step)
  else if (t128 >= 640) sel = (a.taps == 1) ? 4 : 0;
  else if (a.N >= 4096 && t12864 >= 448) sel = 1;
  else sel = 2;
  if (a.tile_sel > 0) sel = (a.tile_sel & 0xff) - 1;   // benchmarking / test override (usdm_gemm_args.tile_sel; ops.gemm fills it from USDM_GEMM_TILE)
  if (a.taps != 1 && sel >= 4 && !(sel >= 12 && pp_taps)) sel = (sel == 6 || sel == 10) ? 1 : ((sel == 5 || sel == 7 || sel == 8) ? 2 : 0);   // DMA tiles are single-tap
  if (sel == 13 && (a.transpose_out || a.epi != USDM_EPI_PLAIN)) sel = 12;   // the 288-row tiles have row-major epilogues only
  if (!tile_out && (a.stats_out || a.ln_mode)) {      // folded LayerNorm: implemented in the epilogues of the ping-pong tiles only (see usdm_gemm_args)
    USDM_CHECK_ARG(sel >= 12 && sel <= 14 && a.dtype == USDM_BF16 && a.epi == USDM_EPI_PLAIN && !a.transpose_out && !a.round_bf16 &&
                       a.N % 128 == 0 && a.groups == 1 && (a.ldc & 3) == 0,
                   "usdm_gemm: stats_out / ln_mode need a bf16 GEMM on the ping-pong tiles with a row-major epilogue and N %% 128 == 0 (tile %d)", sel);
    USDM_CHECK_ARG(!a.stats_out || (a.act == USDM_ACT_NONE && a.split_k <= 1), "usdm_gemm: stats_out needs a plain, unsplit epilogue");
    USDM_CHECK_ARG(a.ln_mode == 0 || (a.ln_stats && a.ln_nt > 0 && a.ln_nt <= 64 && a.ln_C > 0 && a.ln_C == a.ln_nt * 128),
                   "usdm_gemm: ln_stats / ln_nt / ln_C (the producer's tiles are 128 columns wide)");
    USDM_CHECK_ARG(!a.ln_guard || a.ln_guard_ratio > 0.f, "usdm_gemm: ln_guard needs ln_guard_ratio > 0");
    USDM_CHECK_ARG(a.ln_mode != 1 || (a.ln_c && a.act == USDM_ACT_GELU && a.C16 && !a.C32 && !a.residual && a.alpha == 1.0f && a.split_k <= 1),
                   "usdm_gemm: ln_mode 1 is the GELU bf16-out epilogue (no residual, alpha 1)");
    USDM_CHECK_ARG(a.ln_mode != 2 || (a.ln_gamma && a.ln_beta && a.residual && a.res_dtype == USDM_F32 && a.act == USDM_ACT_NONE && (a.ldr & 3) == 0),
                   "usdm_gemm: ln_mode 2 needs an f32 residual, gamma / beta and a plain epilogue");
    USDM_CHECK_ARG(a.ln_mode >= 0 && a.ln_mode <= 2, "usdm_gemm: ln_mode");
  }
  if (tile_out) { *tile_out = sel; return 0; }
  if (a.dtype == USDM_BF16) {
    if (sel == 3) return launch<bf16_t, 128, 128, 2, 4>(a, st);
    if (sel == 4) return launch<bf16_t, 128, 128, 2, 2, true>(a, st);
    if (sel == 5) return launch<bf16_t, 64, 64, 2, 2, true>(a, st);
    if (sel == 6) return launch<bf16_t, 128, 64, 2, 2, true>(a, st);
    if (sel == 7) return launch<bf16_t, 64, 64, 2, 2, true, 3, 2>(a, st);
    if (sel == 8) return launch<bf16_t, 64, 64, 2, 2, true, 4, 2>(a, st);
    if (sel == 9) return launch<bf16_t, 128, 128, 2, 2, true, 4, 1>(a, st);
    if (sel == 10) return launch<bf16_t, 128, 64, 2, 2, true, 3, 2>(a, st);
    if (sel == 11) return launch<bf16_t, 128, 128, 2, 2, true, 3, 2>(a, st);
    if (sel == 12) return launch<bf16_t, 256, 128, 4, 2, true, 3, 2, true>(a, st);   // 8-wave ping-pong loop, one workgroup per CU
    if (sel == 13) return launch<bf16_t, 288, 128, 2, 4, true, 3, 2, true>(a, st);   // row-major epilogues only
    if (sel == 14) return launch<bf16_t, 128, 128, 4, 2, true, 3, 2, true>(a, st);   // the same loop on a 128x128 tile (wave tile 32x64)
    if (sel == 0) return launch<bf16_t, 128, 128>(a, st);
    if (sel == 1) return launch<bf16_t, 128, 64>(a, st);
    return launch<bf16_t, 64, 64>(a, st);
  } else {
    if (sel == 3) return launch<float, 128, 128, 2, 4>(a, st);
    if (sel == 4) return launch<float, 128, 128, 2, 2, true>(a, st);
    if (sel == 5) return launch<float, 64, 64, 2, 2, true>(a, st);
    if (sel == 6) return launch<float, 128, 64, 2, 2, true>(a, st);
    if (sel == 7) return launch<float, 64, 64, 2, 2, true, 3, 2>(a, st);
    if (sel == 8) return launch<float, 64, 64, 2, 2, true, 4, 2>(a, st);
    if (sel == 9) return launch<float, 128, 128, 2, 2, true, 4, 1>(a, st);
    if (sel == 10) return launch<float, 128, 64, 2, 2, true, 3, 2>(a, st);
    if (sel == 11) return launch<float, 128, 128, 2, 2, true, 3, 2>(a, st);
    if (sel == 0) return launch<float, 128, 128>(a, st);
    if (sel == 1) return launch<float, 128, 64>(a, st);
    return launch<float, 64, 64>(a, st);
  }
}

extern "C" int usdm_gemm(const usdm_gemm_args* pa, usdm_stream_t stream) { return gemm_impl(pa, stream, nullptr); }
// The tile usdm_gemm would run these arguments on (12 - 14: the ping-pong tiles, the only ones with the folded-LayerNorm epilogues),
// or a negative number if the arguments are invalid: lets a caller decide BEFORE building a plan whether stats_out / ln_mode apply.
extern "C" int usdm_gemm_tile_for(const usdm_gemm_args* pa) {
  int tile = -1;
  const int rc = gemm_impl(pa, nullptr, &tile);
  return rc == 0 ? tile : -rc;
}
