// XLS-R unit-extractor kernels that are not GEMMs / LayerNorms (those are usdm_gemm f32 / usdm_norm).
// The reference's tokenizer is third-party (seamless_communication UnitExtractor on fairseq2,
// src/inference.py:59,111-113); algorithm as restated in oracle/w2v_oracle.py.  Everything is fp32.
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {

// F.layer_norm over the WHOLE waveform (mean/var over n samples), single workgroup, two passes.
__global__ __launch_bounds__(1024) void wave_norm_kernel(const float* __restrict__ x, int n, float eps, float* __restrict__ y) {
  __shared__ float red[16];
  __shared__ float stat[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < n; i += 1024) s += x[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    stat[0] = t / (float)n;
  }
  __syncthreads();
  const float mean = stat[0];
  float q = 0.f;
  for (int i = tid; i < n; i += 1024) { const float d = x[i] - mean; q += d * d; }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    stat[1] = rsqrtf(t / (float)n + eps);
  }
  __syncthreads();
  const float rstd = stat[1];
  for (int i = tid; i < n; i += 1024) y[i] = (x[i] - mean) * rstd;
}

// First feature-extractor layer fused: Conv1d(1 -> C, k, stride) + LayerNorm(C) + GELU, channels-last out.
// A workgroup owns FR consecutive output frames: their input window ((FR-1)*stride + k samples, contiguous) is staged in LDS
// once with coalesced loads, the lane's CPL channels' taps / bias / LayerNorm parameters are fetched once into registers, and
// every wave then walks its FR/4 frames reading the window from LDS (broadcast reads) - instead of every frame's wave
// re-fetching the 20 KB of taps and its samples from L2.  One wave per output frame for the LayerNorm reduction;
// lane owns C/64 channels (C = 512 -> 8).  Per frame the arithmetic is unchanged.
template <int CPL, int KW, int FR>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, int n, int T, int stride, const float* __restrict__ w,
                                                    const float* __restrict__ b, const float* __restrict__ g,
                                                    const float* __restrict__ be, float eps, float* __restrict__ out) {
  constexpr int C = CPL * 64;
  constexpr int MAXW = (FR - 1) * 8 + KW;          // window for strides up to 8
  __shared__ float win[MAXW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * FR;
  const int64_t x0 = (int64_t)t0 * stride;
  const int nwin = (FR - 1) * stride + KW;
  for (int i = tid; i < nwin; i += 256) win[i] = (x0 + i < n) ? x[x0 + i] : 0.f;
  float wv[CPL][KW], bv[CPL], gv[CPL], bev[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane * CPL + c;
#pragma unroll
    for (int j = 0; j < KW; ++j) wv[c][j] = w[ch * KW + j];
    bv[c] = b[ch]; gv[c] = g[ch]; bev[c] = be[ch];
  }
  __syncthreads();
  for (int f = wave; f < FR; f += 4) {
    const int t = t0 + f;
    if (t >= T) break;
    float xv[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) xv[j] = win[f * stride + j];
    float v[CPL];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < KW; ++j) a = fmaf(wv[c][j], xv[j], a);
      a += bv[c];
      v[c] = a;
      s += a;
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const float d = v[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + eps);
    float o[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) o[c] = gelu_erf((v[c] - mean) * rstd * gv[c] + bev[c]);
    float4* dst = (float4*)(out + (int64_t)t * C + lane * CPL);
    static_assert(CPL % 4 == 0, "channels per lane");
#pragma unroll
    for (int c = 0; c < CPL; c += 4) dst[c / 4] = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
  }
}

// in-place softmax over segments: x[row][seg*ldseg + 0..n) ; columns n..npad are written as zero
__global__ __launch_bounds__(256) void softmax_seg_kernel(float* x, int rows, int nseg, int n, int npad, int64_t ldrow, int ldseg) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= rows * nseg) return;
  const int row = id / nseg, seg = id - row * nseg;
  float* p = x + (int64_t)row * ldrow + (int64_t)seg * ldseg;
  float m = -INFINITY;
  for (int i = lane; i < n; i += 64) m = fmaxf(m, p[i]);
  m = wave_max(m);
  float s = 0.f;
  for (int i = lane; i < n; i += 64) { const float e = expf(p[i] - m); p[i] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int i = lane; i < npad; i += 64) p[i] = i < n ? p[i] * inv : 0.f;
}

// ids[t] = argmin_n ( |x_t|^2 - 2*dots[t][n] + csq[n] ), first minimum wins (torch.argmin)
__global__ __launch_bounds__(256) void kmeans_argmin_kernel(const float* __restrict__ x, int D, const float* __restrict__ dots,
                                                            int64_t ldd, const float* __restrict__ csq, int n_units,
                                                            int64_t* __restrict__ ids, float* __restrict__ margin) {
  __shared__ float sv[4], sv2[4];
  __shared__ int si[4];
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xs = 0.f;
  for (int i = tid; i < D; i += 256) { const float v = x[(int64_t)t * D + i]; xs += v * v; }
  xs = wave_sum(xs);
  __shared__ float sx[4];
  if (lane == 0) sx[wave] = xs;
  __syncthreads();
  const float xsq = (sx[0] + sx[1]) + (sx[2] + sx[3]);
  float best = INFINITY, second = INFINITY;
  int bi = 0x7fffffff;
  for (int n = tid; n < n_units; n += 256) {
    const float d = (xsq - 2.0f * dots[(int64_t)t * ldd + n]) + csq[n];
    if (d < best) { second = best; best = d; bi = n; }
    else if (d < second) second = d;
  }
  // wave reduce (value asc, index asc), tracking the runner-up for the margin report
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { second = fminf(best, os); best = ob; bi = oi; }
    else second = fminf(second, ob);
  }
  if (lane == 0) { sv[wave] = best; sv2[wave] = second; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      if (sv[w] < best || (sv[w] == best && si[w] < bi)) { second = fminf(best, sv2[w]); best = sv[w]; bi = si[w]; }
      else second = fminf(second, sv[w]);
    }
    ids[t] = bi;
    if (margin) margin[t] = second - best;
  }
}
}  // namespace

extern "C" int usdm_wave_layernorm(const float* x, int32_t n, float eps, float* y, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && y && n > 0, "usdm_wave_layernorm: bad args");
  hipLaunchKernelGGL(wave_norm_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, eps, y);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_w2v_conv0(const float* x, int32_t n, int32_t T, int32_t C, int32_t k, int32_t stride, const float* w,
                              const float* b, const float* ln_g, const float* ln_b, float eps, float* out, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && w && b && ln_g && ln_b && out && T > 0, "usdm_w2v_conv0: null args");
  USDM_CHECK_ARG((int64_t)(T - 1) * stride + k <= n, "usdm_w2v_conv0: T frames do not fit in n samples");
  USDM_CHECK_ARG(C == 512 && k == 10, "usdm_w2v_conv0: built for the XLS-R first layer (1->512, k=10); got C=%d k=%d", C, k);
  USDM_CHECK_ARG(stride >= 1 && stride <= 8, "usdm_w2v_conv0: stride %d outside the staged window's range (1..8)", stride);
  constexpr int FR = 64;   // frames per workgroup: 325 staged samples at stride 5
  hipLaunchKernelGGL((conv0_kernel<8, 10, FR>), dim3(cdiv(T, FR)), dim3(256), 0, (hipStream_t)stream, x, n, T, stride, w, b, ln_g, ln_b, eps, out);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_softmax_segments(float* x, int32_t rows, int32_t nseg, int32_t n, int32_t npad, int64_t ldrow, int32_t ldseg,
                                     usdm_stream_t stream) {
  USDM_CHECK_ARG(x && rows > 0 && nseg > 0 && n > 0 && npad >= n && ldseg >= npad, "usdm_softmax_segments: bad args");
  hipLaunchKernelGGL(softmax_seg_kernel, dim3(cdiv((int64_t)rows * nseg, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, nseg, n,
                     npad, ldrow, ldseg);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_kmeans_argmin(const float* x, int32_t T, int32_t D, const float* dots, int64_t ldd, const float* csq,
                                  int32_t n_units, int64_t* ids, float* margin, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && dots && csq && ids && T > 0 && D > 0 && n_units > 0, "usdm_kmeans_argmin: bad args");
  hipLaunchKernelGGL(kmeans_argmin_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, x, D, dots, ldd, csq, n_units, ids, margin);
  USDM_LAUNCH_CHECK();
  return 0;
}
