// Shared device/host helpers for libusdm_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;  // raw bf16 bits in memory

// ---- error plumbing (thread-local last error, C-ABI returns int) -------------------------------
void usdm_set_error(const char* fmt, ...);
#define USDM_CHECK_ARG(cond, ...)                       \
  do {                                                  \
    if (!(cond)) {                                      \
      usdm_set_error(__VA_ARGS__);                      \
      return 2;                                         \
    }                                                   \
  } while (0)
#define USDM_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      usdm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
#define USDM_LAUNCH_CHECK()                                                         \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      usdm_set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved by the plain cast) ---------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, ((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// ---- wave64 all-reductions without LDS traffic: DPP inside 16-lane rows, v_permlane{16,32}_swap across rows ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// sum over the 32-lane half of the wave this lane belongs to (lanes 0-31 / 32-63): wave_sum without its last step
__device__ __forceinline__ float half_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
// sum over the 16-lane DPP row this lane belongs to
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)), with erfc(|z|) from Abramowitz & Stegun 7.1.26 (|abs error| < 1.5e-7, i.e.
// ~1 ulp of the f32 result around |x| ~ 1): 1 + erf(z) = erfc(|z|) for z < 0 (no cancellation), 2 - erfc(|z|) otherwise.
// ~15 VALU ops instead of the ~50 of the library erff: the GELU epilogue of the feed-forward GEMMs was VALU-bound.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t = __frcp_rn(__fmaf_rn(0.3275911f, az, 1.0f));
  float p = __fmaf_rn(t, 1.061405429f, -1.453152027f);
  p = __fmaf_rn(t, p, 1.421413741f);
  p = __fmaf_rn(t, p, -0.284496736f);
  p = __fmaf_rn(t, p, 0.254829592f);
  const float q = p * t * __expf(-az * az);
  return 0.5f * x * (z < 0.f ? q : 2.0f - q);
}

// the same GELU on two values at once: the f32 arithmetic is written on 2-vectors (v_pk_fma_f32 / v_pk_mul_f32, two results per
// VALU instruction) and the reciprocal is the hardware approximation (1 ulp; __frcp_rn expands to the 10-instruction IEEE
// division).  For the GEMM epilogues that run at one workgroup per CU, where nothing hides the activation's VALU time.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t fma2(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t z = x * 0.70710678118654752440f;
  const f32x2_t az = {fabsf(z.x), fabsf(z.y)};
  const f32x2_t d = fma2(az, f32x2_t{0.3275911f, 0.3275911f}, f32x2_t{1.0f, 1.0f});
  const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2_t p = fma2(t, f32x2_t{1.061405429f, 1.061405429f}, f32x2_t{-1.453152027f, -1.453152027f});
  p = fma2(t, p, f32x2_t{1.421413741f, 1.421413741f});
  p = fma2(t, p, f32x2_t{-0.284496736f, -0.284496736f});
  p = fma2(t, p, f32x2_t{0.254829592f, 0.254829592f});
  const f32x2_t w = az * az * -1.4426950408889634f;                 // exp(-az^2) = 2^(-az^2 log2 e)
  const f32x2_t e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2_t q = p * t * e;
  const f32x2_t r = {z.x < 0.f ? q.x : 2.0f - q.x, z.y < 0.f ? q.y : 2.0f - q.y};
  return x * 0.5f * r;
}
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2v(f32x2_t v) {          // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v_t));
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Write-through (sc1) store of a kernel OUTPUT that the next launch consumes (see the note at st_out in gemm.hip: nothing dirty is
// left in this XCD's L2 for the write-back at the kernel boundary).  The s_nop covers the store-data hazard of > 8-byte stores.
template <class T>
__device__ __forceinline__ void st_wt(T* p, const T& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned int wt4;
  typedef __attribute__((ext_vector_type(2))) unsigned int wt2;
  if constexpr (sizeof(T) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(wt4, v)) : "memory");
  else if constexpr (sizeof(T) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(wt2, v)) : "memory");
  else *p = v;
}
