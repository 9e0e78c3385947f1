// BigVGAN-specific kernels (channels-last activations [time][channel]).
//
// usdm_aa_snake: the whole Activation1d(SnakeBeta) of the reference in ONE pass over HBM:
//   replicate-pad -> 2x polyphase up-sampling FIR (12 taps) -> x + sin^2(x e^a)/(e^b + 1e-9)
//   -> replicate-pad -> 12-tap low-pass, stride 2
// (alias_free_torch/act.py:23-28, resample.py:25-33, filter.py:86-95, activations.py:107-120),
// which the reference runs as >= 6 separate PyTorch kernels over 2x-length temporaries.
// Each thread owns two adjacent channels and slides along time with a 7-sample input window and
// a 13-sample activated window held in registers: every input is read once, every up-sampled
// sample gets exactly one sin(), and the 2x intermediate never exists in memory.
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {

__device__ __forceinline__ float snake1(float u, float a, float ib) {
  const float s = sinf(u * a);
  return u + ib * (s * s);
}

struct SnakeCh {  // per-thread state for one channel
  float xw[7];
  float vw[13];
};

__global__ __launch_bounds__(256) void aa_snake_kernel(const usdm_snake_args a) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + tx) * 2;
  const int chunk = blockIdx.y * 16 + ty;
  const int t0 = chunk * a.L;
  if (c >= a.C || t0 >= a.T) return;
  const int T = a.T;
  const int64_t ldx = a.ldx;
  const float* xp = a.x + c;
  float al[2], ib[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool cv = (c + k) < a.Creal;
    float av = cv ? a.alpha[c + k] : 0.f, bv = cv ? a.beta[c + k] : 0.f;
    if (a.logscale) { av = expf(av); bv = expf(bv); }
    al[k] = av;
    ib[k] = 1.0f / (bv + 1e-9f);
  }
  float fu[12], fd[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) { fu[j] = a.fup[j]; fd[j] = a.fdn[j]; }

  auto ldx2 = [&](int t) -> float2 {
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    return *(const float2*)(xp + (int64_t)t * ldx);
  };
  // u[2m], u[2m+1] from the 7-window xw = x~[m-3..m+3]
  auto up2 = [&](const float* xw, float& u0, float& u1) {
    u0 = 2.0f * (((fu[1] * xw[5] + fu[3] * xw[4]) + (fu[5] * xw[3] + fu[7] * xw[2])) + (fu[9] * xw[1] + fu[11] * xw[0]));
    u1 = 2.0f * (((fu[0] * xw[6] + fu[2] * xw[5]) + (fu[4] * xw[4] + fu[6] * xw[3])) + (fu[8] * xw[2] + fu[10] * xw[1]));
  };

  const int tend = min(t0 + a.L, T);
  const bool edge = (t0 < 3) || (tend + 3 > T);
  float v_first[2] = {0.f, 0.f}, v_last[2] = {0.f, 0.f};
  if (edge) {  // replicate padding applies to the ACTIVATED 2x signal: v[-k] = v[0], v[2T-1+k] = v[2T-1]
    float w0[2][7], w1[2][7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const float2 p = ldx2(k - 3), q = ldx2(T - 1 + k - 3);
      w0[0][k] = p.x; w0[1][k] = p.y; w1[0][k] = q.x; w1[1][k] = q.y;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float u0, u1;
      up2(w0[k], u0, u1);
      v_first[k] = snake1(u0, al[k], ib[k]);
      up2(w1[k], u0, u1);
      v_last[k] = snake1(u1, al[k], ib[k]);
    }
  }

  SnakeCh s[2];
#pragma unroll
  for (int k = 0; k < 13; ++k) { s[0].vw[k] = 0.f; s[1].vw[k] = 0.f; }
  // window for m = t0-3 holds x~[t0-6 .. t0]; pre-load all but the newest
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float2 p = ldx2(t0 - 7 + k + 1);  // becomes xw[k+1] -> shifted to xw[k] on first iteration
    s[0].xw[k + 1] = p.x; s[1].xw[k + 1] = p.y;
  }
  for (int m = t0 - 3; m < tend + 3; ++m) {
    const float2 nx = ldx2(m + 3);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int j = 0; j < 6; ++j) s[k].xw[j] = s[k].xw[j + 1];
      s[k].xw[6] = k == 0 ? nx.x : nx.y;
      float u0, u1;
      up2(s[k].xw, u0, u1);
      float v0 = snake1(u0, al[k], ib[k]), v1 = snake1(u1, al[k], ib[k]);
      if (edge) {
        const int n0 = 2 * m;
        if (n0 < 0) v0 = v_first[k]; else if (n0 > 2 * T - 1) v0 = v_last[k];
        if (n0 + 1 < 0) v1 = v_first[k]; else if (n0 + 1 > 2 * T - 1) v1 = v_last[k];
      }
#pragma unroll
      for (int j = 0; j < 11; ++j) s[k].vw[j] = s[k].vw[j + 2];
      s[k].vw[11] = v0;
      s[k].vw[12] = v1;
    }
    const int t = m - 3;
    if (t >= t0) {
      float o[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float* w = s[k].vw;
        o[k] = (((fd[0] * w[0] + fd[1] * w[1]) + (fd[2] * w[2] + fd[3] * w[3])) +
                ((fd[4] * w[4] + fd[5] * w[5]) + (fd[6] * w[6] + fd[7] * w[7]))) +
               ((fd[8] * w[8] + fd[9] * w[9]) + (fd[10] * w[10] + fd[11] * w[11]));
        if (c + k >= a.Creal) o[k] = 0.f;
      }
      if (a.out32) *(float2*)(a.out32 + (int64_t)t * a.ldo + c) = make_float2(o[0], o[1]);
      if (a.out16) *(unsigned*)((bf16_t*)a.out16 + (int64_t)t * a.ldo + c) = pack_bf2(o[0], o[1]);
    }
  }
}

// out = (a + b + c) * scale  (vocoder/models.py:198-204: AMP block outputs summed then /3)
__global__ void sum3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float scale, int64_t n4, float* out32, bf16_t* out16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i], z = ((const float4*)c)[i];
  float4 r;
  r.x = ((x.x + y.x) + z.x) * scale; r.y = ((x.y + y.y) + z.y) * scale;
  r.z = ((x.z + y.z) + z.z) * scale; r.w = ((x.w + y.w) + z.w) * scale;
  if (out32) ((float4*)out32)[i] = r;
  if (out16) { uint2 o; o.x = pack_bf2(r.x, r.y); o.y = pack_bf2(r.z, r.w); ((uint2*)out16)[i] = o; }
}

// channels-first f32 [B][C][T] -> channels-last [B][T][Cpad] (f32 and/or bf16), zero-padded channels,
// optional affine (x*scale + shift): used for mel de-normalisation (model_util.py:103) + layout change.
__global__ void cf_to_cl_kernel(const float* __restrict__ x, int C, int T, int Cpad, float scale, float shift,
                                float* out32, bf16_t* out16) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? x[((int64_t)b * C + c) * T + t] * scale + shift : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < Cpad) {
      const float v = (c < C) ? tile[tx][i] : 0.f;
      const int64_t o = ((int64_t)b * T + t) * Cpad + c;
      if (out32) out32[o] = v;
      if (out16) out16[o] = f2bf(v);
    }
  }
}
}  // namespace

extern "C" int usdm_aa_snake(const usdm_snake_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->x && pa->alpha && pa->beta, "usdm_aa_snake: null args");
  usdm_snake_args a = *pa;
  USDM_CHECK_ARG(a.T > 0 && a.C > 0 && a.C % 2 == 0 && a.Creal <= a.C, "usdm_aa_snake: bad T/C");
  USDM_CHECK_ARG(a.ldx % 2 == 0 && a.ldo % 2 == 0, "usdm_aa_snake: strides must be even");
  USDM_CHECK_ARG(a.out32 || a.out16, "usdm_aa_snake: no output");
  if (a.L <= 0) {
    a.L = 32;
    const int64_t cols = cdiv(a.C, 32);
    while (a.L > 8 && cols * cdiv(a.T, 16 * a.L) < 1024) a.L >>= 1;
  }
  dim3 grid(cdiv(a.C, 32), cdiv(a.T, 16 * a.L));
  hipLaunchKernelGGL(aa_snake_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_sum3_scale(const float* a, const float* b, const float* c, float scale, int64_t n,
                               float* out32, void* out16, usdm_stream_t stream) {
  USDM_CHECK_ARG(a && b && c && n > 0 && n % 4 == 0 && (out32 || out16), "usdm_sum3_scale: bad args");
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(sum3_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, scale, n4, out32,
                     (bf16_t*)out16);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_cf_to_cl(const float* x, int32_t B, int32_t C, int32_t T, int32_t Cpad, float scale, float shift,
                             float* out32, void* out16, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && B > 0 && C > 0 && T > 0 && Cpad >= C && (out32 || out16), "usdm_cf_to_cl: bad args");
  dim3 grid(cdiv(T, 32), cdiv(Cpad, 32), B);
  hipLaunchKernelGGL(cf_to_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, C, T, Cpad, scale, shift, out32,
                     (bf16_t*)out16);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_snake_args(void) { return (int)sizeof(usdm_snake_args); }

// ---------------------------------------------------------------------------------------------
// Mel front end of the speech prompt (vocoder/meldataset.py:55-78): reflect padding + framing + Hann
// window in one pass, and the magnitude sqrt(re^2 + im^2 + 1e-9) of the DFT-as-GEMM output.
// The DFT itself (frames x [cos | -sin] matrix) and the mel projection + log run on usdm_gemm (f32 MFMA).
namespace {
__global__ void stft_frames_kernel(const float* __restrict__ x, int n, int n_fft, int hop, int pad, const float* __restrict__ win,
                                   float* __restrict__ out, int T) {
  const int t = blockIdx.x;
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
    int j = t * hop + i - pad;            // index into the un-padded signal
    if (j < 0) j = -j;                    // reflect (no edge repeat), as F.pad(mode='reflect')
    if (j > n - 1) j = 2 * (n - 1) - j;
    float v = x[j];
    v = fminf(fmaxf(v, -1.0f), 1.0f);     // get_mel clamps float audio to [-1, 1] (model_util.py:32)
    out[(int64_t)t * n_fft + i] = v * win[i];
  }
}
__global__ void stft_mag_kernel(const float* __restrict__ ri, int64_t ld, int nbins, float eps, float* __restrict__ out, int64_t ldo,
                                int nbins_pad) {
  const int t = blockIdx.x;
  for (int k = threadIdx.x; k < nbins_pad; k += blockDim.x) {
    float v = 0.f;
    if (k < nbins) {
      const float re = ri[(int64_t)t * ld + k], im = ri[(int64_t)t * ld + nbins + k];
      v = sqrtf((re * re + im * im) + eps);
    }
    out[(int64_t)t * ldo + k] = v;
  }
}
}  // namespace

extern "C" int usdm_stft_frames(const float* x, int32_t n, int32_t n_fft, int32_t hop, int32_t pad, const float* window,
                                float* frames, int32_t T, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && window && frames && n > pad && T > 0, "usdm_stft_frames: bad args (signal must be longer than the reflect pad)");
  USDM_CHECK_ARG((int64_t)(T - 1) * hop + n_fft <= (int64_t)n + 2 * pad, "usdm_stft_frames: T frames exceed the padded signal");
  hipLaunchKernelGGL(stft_frames_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, x, n, n_fft, hop, pad, window, frames, T);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_stft_mag(const float* re_im, int64_t ld, int32_t T, int32_t nbins, float eps, float* out, int64_t ldo,
                             int32_t nbins_pad, usdm_stream_t stream) {
  USDM_CHECK_ARG(re_im && out && T > 0 && nbins > 0 && nbins_pad >= nbins && ld >= 2 * nbins && ldo >= nbins_pad, "usdm_stft_mag: bad args");
  hipLaunchKernelGGL(stft_mag_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, re_im, ld, nbins, eps, out, ldo, nbins_pad);
  USDM_LAUNCH_CHECK();
  return 0;
}

// frames[t][c] = x[t*hop + c - offset] (zero outside [0,n)), c < frame_len : im2col of a 1-D signal for the
// polyphase sample-rate converter (torchaudio.transforms.Resample as a GEMM).
namespace {
__global__ void frame_signal_kernel(const float* __restrict__ x, int n, int frame_len, int hop, int offset, float* __restrict__ out, int T) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x; c < frame_len; c += blockDim.x) {
    const int64_t j = (int64_t)t * hop + c - offset;
    out[(int64_t)t * frame_len + c] = (j >= 0 && j < n) ? x[j] : 0.f;
  }
}
}  // namespace
extern "C" int usdm_frame_signal(const float* x, int32_t n, int32_t frame_len, int32_t hop, int32_t offset, float* frames, int32_t T,
                                 usdm_stream_t stream) {
  USDM_CHECK_ARG(x && frames && n > 0 && frame_len > 0 && hop > 0 && T > 0, "usdm_frame_signal: bad args");
  hipLaunchKernelGGL(frame_signal_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, x, n, frame_len, hop, offset, frames, T);
  USDM_LAUNCH_CHECK();
  return 0;
}
