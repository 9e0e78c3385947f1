// Voicebox-specific glue kernels (everything dense runs in gemm.hip / attn.hip / norm.hip).
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {

// A[bx][s][0:E] = table[id] (pre-scaled by sqrt(E) at pack time), [E:E+F] = y[:, s], [E+F:E+2F] = cond[:, s]
// (networks.py:305-307).  dup == 2 is the classifier-free-guidance doubling of voicebox.py:60-65:
// the first B_in rows of the batch are the unconditional copy (null token, zero cond).
template <typename OT>   // OT = bf16_t (MFMA-operand rows of the default plan) or float (the exact-f32 plan; table is f32 then)
__global__ void vb_build_input_kernel(const usdm_vb_input_args a) {
  const int s = blockIdx.x, bx = blockIdx.y;
  const int b = bx % a.B_in, half = bx / a.B_in;
  const bool uncond = (a.dup == 2) && (half == 0);
  const int64_t id = uncond ? (int64_t)a.null_id : a.ids[(int64_t)b * a.S + s];
  OT* out = (OT*)a.out + ((int64_t)bx * a.S + s) * a.ldo;
  const OT* row = (const OT*)a.table + id * a.E;
  constexpr int V = 16 / (int)sizeof(OT);
  for (int c = threadIdx.x * V; c < a.E; c += blockDim.x * V) *(u32x4*)(out + c) = *(const u32x4*)(row + c);
  auto cvt = [](float v) -> OT { if constexpr (sizeof(OT) == 2) return f2bf(v); else return v; };
  for (int c = threadIdx.x; c < a.F; c += blockDim.x) {
    const int64_t src = ((int64_t)b * a.F + c) * a.S + s;
    out[a.E + c] = cvt(a.y[src]);
    out[a.E + a.F + c] = cvt((uncond || !a.use_cond) ? 0.f : a.cond[src]);
  }
  for (int c = a.E + 2 * a.F + threadIdx.x; c < a.ldo; c += blockDim.x) out[c] = cvt(0.f);
}

// Attention probabilities of the exact-f32 Voicebox plan, in place: x[row][h*ldseg + j] holds (q_i . k_j) already scaled;
// p = softmax_j(x + bias) with bias = -slope_h |i - j| (0 for key 0: networks.py:319-327) over keys j < kv_len[b], masked keys
// get probability 0 exactly as finfo.min does in the reference (networks.py:334-341), pad columns [n, npad) are zeroed.
// One wave per (row, head); same operation order as torch (add bias, subtract the row maximum, exp, sum, divide).
__global__ __launch_bounds__(256) void softmax_alibi_kernel(float* x, int rows, int rows_per_batch, int nseg, int n, int npad, int64_t ldrow,
                                                            int ldseg, const float* slopes, const int* kv_len, int col0_zero) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (id >= rows * nseg) return;
  const int row = id / nseg, h = id - row * nseg;
  const int b = row / rows_per_batch, i = row - b * rows_per_batch;
  const int len = kv_len ? min(kv_len[b], n) : n;
  const float slope = slopes ? slopes[h] : 0.f;
  float* p = x + (int64_t)row * ldrow + (int64_t)h * ldseg;
  float m = -INFINITY;
  for (int j = lane; j < len; j += 64) {
    const float bias = (col0_zero && j == 0) ? 0.f : -slope * fabsf((float)(i - j));
    const float v = p[j] + bias;
    p[j] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) { const float e = expf(p[j] - m); p[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < npad; j += 64) p[j] = j < len ? p[j] * inv : 0.f;
}

// sinusoidal time token into row 0 of every batch (networks.py:19-28, 312-313)
__global__ void vb_time_token_kernel(const float* t, int t_stride, const float* freqs, int Bx, int H,
                                     int64_t rows_per_batch, float* h32, bf16_t* h16) {
  const int b = blockIdx.x;
  const int half = H / 2;
  const float tv = t[b * t_stride];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float v = (1000.0f * tv) * freqs[i];  // freqs = exp(arange(half) * -ln(1e4)/(half-1)), host-made
    const float sv = sinf(v), cv = cosf(v);
    const int64_t o = (int64_t)b * rows_per_batch * H;
    h32[o + i] = sv;
    h32[o + half + i] = cv;
    if (h16) { h16[o + i] = f2bf(sv); h16[o + half + i] = f2bf(cv); }
  }
}

// One elementwise pass of the CFM solvers (voicebox.py:66-72 CFG combine, :83-90 Euler, :112-131 Heun)
__global__ void vb_solver_kernel(const usdm_vb_solver_args a) {
  const int64_t n = (int64_t)a.B * a.F * a.S;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = (int)(i % a.S);
  float v;
  if (a.cfg) {
    const float vu = a.vout[i], vc = a.vout[n + i];
    v = vc + a.gs * (vc - vu);
  } else {
    v = a.vout[i];
  }
  float zn;
  if (a.mode == 0) {  // Euler / Heun predictor
    if (a.v1) a.v1[i] = v;
    zn = a.z[i] + a.dt * v;
  } else {            // Heun corrector
    zn = a.z[i] + (a.dt * (a.v1[i] + v)) / 2.0f;
  }
  if (a.eps && s < a.P) zn = a.c_eps * a.eps[i] + a.c_cond * a.cond[i];
  if (a.z_in) a.z_in[i] = zn;
  if (a.z_commit) a.z_commit[i] = zn;
  if (i == 0 && a.t_cur) {
    for (int b = 0; b < a.t_count; ++b) a.t_cur[b] = a.t_next;
  }
}
}  // namespace

extern "C" int usdm_vb_build_input(const usdm_vb_input_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->ids && pa->y && pa->table && pa->out, "usdm_vb_build_input: null args");
  const usdm_vb_input_args& a = *pa;
  USDM_CHECK_ARG(a.B_in > 0 && (a.dup == 1 || a.dup == 2) && a.S > 0 && a.E % 8 == 0 && a.ldo >= a.E + 2 * a.F && a.ldo % 8 == 0,
                 "usdm_vb_build_input: bad sizes");
  USDM_CHECK_ARG(!a.use_cond || a.cond, "usdm_vb_build_input: cond missing");
  USDM_CHECK_ARG(a.out_dtype == USDM_BF16 || a.out_dtype == USDM_F32, "usdm_vb_build_input: out_dtype");
  if (a.out_dtype == USDM_F32) hipLaunchKernelGGL(vb_build_input_kernel<float>, dim3(a.S, a.B_in * a.dup), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(vb_build_input_kernel<bf16_t>, dim3(a.S, a.B_in * a.dup), dim3(256), 0, (hipStream_t)stream, a);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_softmax_alibi(float* x, int32_t rows, int32_t rows_per_batch, int32_t nheads, int32_t n, int32_t npad, int64_t ldrow,
                                  int32_t ldseg, const float* slopes, const int32_t* kv_len, int32_t col0_zero, usdm_stream_t stream) {
  USDM_CHECK_ARG(x && rows > 0 && rows_per_batch > 0 && nheads > 0 && n > 0 && npad >= n && ldseg >= npad, "usdm_softmax_alibi: bad args");
  hipLaunchKernelGGL(softmax_alibi_kernel, dim3(cdiv((int64_t)rows * nheads, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, rows_per_batch,
                     nheads, n, npad, ldrow, ldseg, slopes, kv_len, col0_zero);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_vb_time_token(const float* t, int32_t t_stride, const float* freqs, int32_t Bx, int32_t H,
                                  int64_t rows_per_batch, float* h32, void* h16, usdm_stream_t stream) {
  USDM_CHECK_ARG(t && freqs && h32 && Bx > 0 && H % 2 == 0, "usdm_vb_time_token: bad args");
  hipLaunchKernelGGL(vb_time_token_kernel, dim3(Bx), dim3(256), 0, (hipStream_t)stream, t, t_stride, freqs, Bx, H,
                     rows_per_batch, h32, (bf16_t*)h16);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_vb_solver_step(const usdm_vb_solver_args* pa, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->vout && pa->z, "usdm_vb_solver_step: null args");
  const usdm_vb_solver_args& a = *pa;
  USDM_CHECK_ARG(a.mode == 0 || (a.mode == 1 && a.v1), "usdm_vb_solver_step: corrector needs v1");
  USDM_CHECK_ARG(!a.eps || a.cond, "usdm_vb_solver_step: re-noising needs cond");
  const int64_t n = (int64_t)a.B * a.F * a.S;
  hipLaunchKernelGGL(vb_solver_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_copy_bytes(void* dst, const void* src, int64_t nbytes, usdm_stream_t stream) {
  USDM_CHECK_ARG(dst && src && nbytes >= 0, "usdm_copy_bytes: bad args");
  USDM_HIP(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}
extern "C" int usdm_sizeof_vb_input_args(void) { return (int)sizeof(usdm_vb_input_args); }
extern "C" int usdm_sizeof_vb_solver_args(void) { return (int)sizeof(usdm_vb_solver_args); }

// ---------------------------------------------------------------------------------------------
// process_unit (model_util.py:50-54): repeat_interleave(rep) -> frames of `hop` -> per-frame mode,
// ties -> smallest id.  Integer work, one thread per output frame, nothing materialised.
namespace {
__global__ void process_unit_kernel(const int64_t* __restrict__ u, int n, int rep, int hop, int64_t* __restrict__ out,
                                    int nframes) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nframes) return;
  const int64_t lo = (int64_t)f * hop, hi = lo + hop;  // [lo, hi) in the repeated sequence
  const int i0 = (int)(lo / rep), i1 = (int)((hi - 1) / rep);
  int64_t best_v = 0;
  int64_t best_c = -1;
  for (int i = i0; i <= i1; ++i) {
    const int64_t v = u[i];
    int64_t c = 0;
    for (int j = i0; j <= i1; ++j) {
      if (u[j] != v) continue;
      const int64_t a = max((int64_t)j * rep, lo), b = min((int64_t)(j + 1) * rep, hi);
      c += b - a;
    }
    if (c > best_c || (c == best_c && v < best_v)) { best_c = c; best_v = v; }
  }
  out[f] = best_v;
}
}  // namespace

extern "C" int usdm_process_unit(const int64_t* units, int32_t n, int32_t rep, int32_t hop, int64_t* out,
                                 int32_t nframes, usdm_stream_t stream) {
  USDM_CHECK_ARG(units && out && n > 0 && rep > 0 && hop > 0, "usdm_process_unit: bad args");
  USDM_CHECK_ARG(nframes == (int)(((int64_t)n * rep) / hop), "usdm_process_unit: nframes must be floor(n*rep/hop)");
  if (nframes == 0) return 0;
  hipLaunchKernelGGL(process_unit_kernel, dim3(cdiv(nframes, 128)), dim3(128), 0, (hipStream_t)stream, units, n, rep, hop,
                     out, nframes);
  USDM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Padding mask of ragged batches (networks.py:330-333 `hidden_states[~mask] = 0`, the `* y_mask` of :94,:257-266,:370-372):
// zero every time step t >= valid_len[b] - off of x, for rows-major [B][T][C] (layout 0) or channels-first [B][C][T] (1).
namespace {
__global__ void mask_time_kernel(float* x32, bf16_t* x16, int B, int T, int C, int layout, const int* valid_len, int off) {
  const int64_t n = (int64_t)B * T * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int b, t;
    if (layout == 0) { const int64_t r = i / C; b = (int)(r / T); t = (int)(r - (int64_t)b * T); }
    else { b = (int)(i / ((int64_t)C * T)); t = (int)(i % T); }
    if (t >= valid_len[b] - off) {
      if (x32) x32[i] = 0.f;
      if (x16) x16[i] = 0;
    }
  }
}
}  // namespace
extern "C" int usdm_mask_time(float* x32, void* x16, int32_t B, int32_t T, int32_t C, int32_t layout, const int32_t* valid_len,
                              int32_t off, usdm_stream_t stream) {
  USDM_CHECK_ARG((x32 || x16) && valid_len && B > 0 && T > 0 && C > 0 && (layout == 0 || layout == 1), "usdm_mask_time: bad args");
  const int64_t n = (int64_t)B * T * C;
  hipLaunchKernelGGL(mask_time_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x32, (bf16_t*)x16, B, T, C, layout, valid_len, off);
  USDM_LAUNCH_CHECK();
  return 0;
}
