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
//   * register prefetch of K-step k+1 is in flight while K-step k is multiplied.
//   * bf16: v_mfma_f32_16x16x32_bf16 ; f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {

struct GemmDev {
  usdm_gemm_args a;
  int tiles_m, tiles_n;
};

// LDS image of one operand: [sub-chunk 0..1][row][4 pieces of 16 B], piece index XOR f(row).
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmDev g) {
  const usdm_gemm_args& a = g.a;
  constexpr int ES = sizeof(T);       // element size
  constexpr int CE = 64 / ES;         // elements per chunk
  constexpr int PE = 16 / ES;         // elements per 16-B piece
  constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 MFMA tiles per wave
  constexpr int LA = BM / 32, LB = BN / 32;  // 16-B loads per thread per K-step
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * 128];
  char* sA = smem;
  char* sB = smem + BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lc = lane >> 4;

  // tile mapping: consecutive blocks walk N first (share the A row panel)
  const int tile = blockIdx.x;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const int bz = z / a.groups, gz = z - bz * a.groups;

  const char* Abase = (const char*)a.A + (a.a_gstride * gz + a.a_bstride * bz) * ES;
  const char* Wbase = (const char*)a.W + (a.w_gstride * gz) * ES;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Abase, 0, 0x80000000u, 0x00020000);
  auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wbase, 0, 0x80000000u, 0x00020000);

  const int cpt = a.Kc / CE;            // chunks per tap
  const int Q = a.taps * cpt;           // total chunks
  const int nks = (Q + 1) >> 1;

  // loader coordinates of this thread
  const int sub = (tid >> 2) & 1, pc = tid & 3, r0 = tid >> 3;
  const unsigned lda_b = (unsigned)(a.lda * ES), ldw_b = (unsigned)(a.ldw * ES);
  const unsigned OOB = 0xFFFFFFF0u;

  u32x4 ra[LA], rb[LB];

  auto load_regs = [&](int ks) {
    const int q = 2 * ks + sub;
    const bool qv = q < Q;
    const int tap = q / cpt;
    const int cb = (q - tap * cpt) * CE + pc * PE;
    const unsigned colA = (unsigned)((cb + (int64_t)tap * a.a_tap_stride) * ES);
    const unsigned colW = (unsigned)((q * CE + pc * PE) * ES);
    const int roff = a.a_row_off + tap * a.a_row_step;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + r0 + 32 * i;
      const int row = m * a.a_row_mul + roff;
      const bool v = qv && ((unsigned)row < (unsigned)a.rowsA);
      const unsigned off = v ? ((unsigned)row * lda_b + colA) : OOB;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int n = n0 + r0 + 32 * i;
      const bool v = qv && (n < a.N);
      const unsigned off = v ? ((unsigned)n * ldw_b + colW) : OOB;
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, off, 0, 0);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = r0 + 32 * i;
      *(u32x4*)(sA + sub * (BM * 64) + r * 64 + ((pc ^ swz(r)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int r = r0 + 32 * i;
      *(u32x4*)(sB + sub * (BN * 64) + r * 64 + ((pc ^ swz(r)) << 4)) = rb[i];
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_regs(0);
  for (int ks = 0; ks < nks; ++ks) {
    store_lds();
    __syncthreads();
    if (ks + 1 < nks) load_regs(ks + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = wm * (BM / 2) + i * 16 + lr;
        fa[i] = *(const u32x4*)(sA + s * (BM * 64) + r * 64 + ((lc ^ swz(r)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = wn * (BN / 2) + j * 16 + lr;
        fb[j] = *(const u32x4*)(sB + s * (BN * 64) + r * 64 + ((lc ^ swz(r)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // NB: bit_cast of a vector-element lvalue reads element 0; go through scalars.
              const unsigned ua = fa[i][e], ub = fb[j][e];
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua), __uint_as_float(ub), acc[i][j], 0, 0, 0);
            }
          }
        }
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const int gcol = a.c_gcol * gz;
  const float* bias = a.bias;
  if (a.act == USDM_ACT_SWIGLU) {
    // tiles (2p, 2p+1) of a wave hold gate/up for the same 16 output features
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int nt = (n0 + wn * (BN / 2) + j * 16) >> 5;  // pair index
        const int nout = nt * 16 + lr;
        const int ngate = n0 + wn * (BN / 2) + j * 16 + lr;
        if (ngate >= a.N) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * (BM / 2) + i * 16 + lc * 4 + e;
          if (m >= a.M) continue;
          float gt = a.alpha * acc[i][j][e], up = a.alpha * acc[i][j + 1][e];
          if (bias) { gt += bias[gcol + ngate]; up += bias[gcol + ngate + 16]; }
          float o;
          if (a.round_bf16) {
            gt = round_bf(gt); up = round_bf(up);
            const float s = round_bf(gt / (1.0f + __expf(-gt)));
            o = round_bf(s * up);
          } else {
            o = (gt / (1.0f + __expf(-gt))) * up;
          }
          const int64_t row = ((int64_t)bz * a.c_bstride + m) * a.c_row_mul + a.c_row_off;
          if (a.C32) ((float*)a.C32)[row * a.ldc + (gcol >> 1) + nout] = o;
          if (a.C16) ((bf16_t*)a.C16)[row * a.ldc + (gcol >> 1) + nout] = f2bf(o);
        }
      }
    return;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + lr;
      if (n >= a.N) continue;
      const float bv = bias ? bias[gcol + n] : 0.f;
      const int mb = m0 + wm * (BM / 2) + i * 16 + lc * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = a.alpha * acc[i][j][e] + bv;
        if (a.round_bf16) x = round_bf(x);
        if (a.act == USDM_ACT_GELU) x = gelu_erf(x);
        else if (a.act == USDM_ACT_TANH) x = tanhf(x);
        v[e] = x;
      }
      if (a.epi == USDM_EPI_QKV_HEADS) {
        const int HD = a.qkv_H * a.qkv_D;
        const int part = n / HD, hn = n - part * HD;
        const int h = hn / a.qkv_D, d = hn - h * a.qkv_D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mb + e;
          if (m >= a.M) continue;
          const int b = m / a.qkv_S, s = m - b * a.qkv_S;
          const int64_t bh = (int64_t)b * a.qkv_H + h;
          if (part == 0)
            ((bf16_t*)a.qkv_q)[(bh * a.qkv_Spad + s) * a.qkv_D + d] = f2bf(v[e]);
          else if (part == 1)
            ((bf16_t*)a.qkv_k)[(bh * a.qkv_Spad + s) * a.qkv_D + d] = f2bf(v[e]);
          else
            ((bf16_t*)a.qkv_v)[(bh * a.qkv_D + d) * a.qkv_Spad + s] = f2bf(v[e]);
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mb + e;
        if (m >= a.M) continue;
        const int64_t row = ((int64_t)bz * a.c_bstride + m) * a.c_row_mul + a.c_row_off;
        float x = v[e];
        if (a.residual) {
          const int64_t ri = row * a.ldr + gcol + n;
          x += (a.res_dtype == USDM_F32) ? ((const float*)a.residual)[ri] : bf2f(((const bf16_t*)a.residual)[ri]);
          if (a.round_bf16) x = round_bf(x);
        }
        const int64_t oi = a.transpose_out ? ((int64_t)(gcol + n) * a.ldc + row) : (row * a.ldc + gcol + n);
        if (a.C32) ((float*)a.C32)[oi] = x;
        if (a.C16) ((bf16_t*)a.C16)[oi] = f2bf(x);
      }
    }
}

template <typename T, int BM, int BN>
int launch(const usdm_gemm_args& a, hipStream_t st) {
  GemmDev g;
  g.a = a;
  g.tiles_m = cdiv(a.M, BM);
  g.tiles_n = cdiv(a.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, 1, a.groups * a.batch);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN>), grid, dim3(256), 0, st, g);
  USDM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int usdm_gemm(const usdm_gemm_args* pa, usdm_stream_t stream) {
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
  if (a.act == USDM_ACT_SWIGLU) USDM_CHECK_ARG(a.N % 32 == 0 && !a.transpose_out && !a.residual, "usdm_gemm: swiglu needs N%%32==0");
  if (a.epi == USDM_EPI_QKV_HEADS)
    USDM_CHECK_ARG(a.qkv_q && a.qkv_k && a.qkv_v && a.N == 3 * a.qkv_H * a.qkv_D && a.qkv_S > 0 && a.qkv_Spad >= a.qkv_S,
                   "usdm_gemm: bad qkv epilogue args");
  hipStream_t st = (hipStream_t)stream;
  // tile heuristic: fill >= ~2 waves of the 256 CUs when possible
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.groups * a.batch;
  const bool small_n = a.N <= 64;
  if (a.dtype == USDM_BF16) {
    if (!small_n && t128 >= 384) return launch<bf16_t, 128, 128>(a, st);
    if (small_n) return launch<bf16_t, 128, 64>(a, st);
    return launch<bf16_t, 64, 64>(a, st);
  } else {
    if (!small_n && t128 >= 384) return launch<float, 128, 128>(a, st);
    if (small_n) return launch<float, 128, 64>(a, st);
    return launch<float, 64, 64>(a, st);
  }
}
