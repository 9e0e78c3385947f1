/*
 * libusdm_hip.so — C-ABI of the MI355X (gfx950) hot path of USDM inference.
 *
 * The reference (ishine/usdm) has no FFI / plugin boundary: its inference path is Python calling
 * PyTorch-CUDA ops.  This header is therefore the *new* native boundary that the
 * Python drop-ins (usdm_amd/voicebox/..., usdm_amd/unit_extractor.py, usdm_amd/llm.py) bind with
 * ctypes.  Every entry point cites the reference call site whose arithmetic it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor), unless named host_*
 *   - no allocation, no synchronisation, no hidden global state; all work is enqueued on `stream`
 *   - return value: 0 = ok, 1 = HIP runtime error, 2 = bad argument; text via usdm_last_error()
 *   - dtype codes: USDM_BF16 = 0 (raw bf16 bits), USDM_F32 = 1
 */
#ifndef USDM_HIP_H_
#define USDM_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* usdm_stream_t; /* hipStream_t */

enum { USDM_BF16 = 0, USDM_F32 = 1 };
enum { USDM_ACT_NONE = 0, USDM_ACT_GELU = 1, USDM_ACT_SWIGLU = 3, USDM_ACT_TANH = 4 };
enum { USDM_EPI_PLAIN = 0, USDM_EPI_QKV_HEADS = 1 };

const char* usdm_last_error(void);
int usdm_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Universal "tap-GEMM" on the MFMA matrix cores (bf16 16x16x32 / exact-f32 16x16x4):
 *
 *   acc[m][n] = sum_{tap<taps} sum_{c<Kc} A[row(m,tap)][c + tap*a_tap_stride] * W[n][tap*Kc + c]
 *   row(m,tap) = m*a_row_mul + a_row_off + tap*a_row_step      (rows outside [0,rowsA) read as 0)
 *   v = alpha*acc + bias[n];  v = act(v);  v += residual[m][n];  store
 *
 * With channels-last activations [time][channel] this one kernel is every dense op of the path:
 *   nn.Linear / Conv1d k=1        networks.py:141-144,219-222,299,301; voicebox transformer GEMMs
 *   dilated Conv1d                vocoder/models.py:33-49 (AMPBlock1 convs), :150 (conv_pre)
 *   ConvTranspose1d (per phase)   vocoder/models.py:157-162 (ups)
 *   grouped Conv1d                networks.py:70-76 (PositionalConvEmbedding)
 *   strided Conv1d                XLS-R feature extractor (third-party)
 *   skip Linear over cat[h,skip]  networks.py:364 (two-source K via a_tap_stride)
 *   Mistral prefill projections   HF MistralAttention/MistralMLP (third-party)
 * ---------------------------------------------------------------------------------------------- */
typedef struct usdm_gemm_args {
  int32_t dtype;         /* USDM_BF16 or USDM_F32: element type of A and W                        */
  int32_t M, N;          /* output rows / columns (per group, per batch)                          */
  int32_t taps, Kc;      /* Kc: channels per tap, multiple of 32 (bf16) / 16 (f32)                */
  const void* A;
  int64_t lda;           /* elements between consecutive A rows                                   */
  int32_t rowsA;         /* number of valid A rows (per batch); others read as zero               */
  int32_t a_row_mul, a_row_off, a_row_step;
  int64_t a_tap_stride;  /* elements added to the A column base per tap (two-source K)            */
  const void* W;
  int64_t ldw;           /* elements between consecutive W rows (>= taps*Kc)                      */
  int32_t groups, batch; /* grid z = batch*groups                                                 */
  int64_t a_gstride;     /* A column offset per group (elements)                                  */
  int64_t w_gstride;     /* W element offset per group                                            */
  int64_t a_bstride;     /* A element offset per batch                                            */
  int32_t c_gcol;        /* output/bias/residual column offset per group                          */
  int64_t c_bstride;     /* output/residual ROW offset per batch (before c_row_mul)               */
  const float* bias;     /* [groups*N] or NULL                                                    */
  float alpha;
  int32_t act;           /* USDM_ACT_*                                                            */
  int32_t round_bf16;    /* 1: round acc(+bias) to bf16 before act/residual (HF bf16 semantics)   */
  const void* residual;  /* [rows][ldr] of res_dtype or NULL; indexed like the output             */
  int32_t res_dtype;
  int64_t ldr;
  void* C32;             /* optional f32 output                                                   */
  void* C16;             /* optional bf16 output                                                  */
  int64_t ldc;           /* elements between output rows (or between columns if transpose_out)    */
  int32_t c_row_mul, c_row_off; /* output row = (b*c_bstride + m)*c_row_mul + c_row_off           */
  int32_t transpose_out; /* 1: store C^T, i.e. out[n][row]                                        */
  int32_t epi;           /* USDM_EPI_*                                                            */
  /* USDM_EPI_QKV_HEADS: N = 3*H*D, rows m = b*S + s; Q,K -> [B][H][S_pad][D], V -> [B][H][D][S_pad] */
  int32_t qkv_S, qkv_Spad, qkv_H, qkv_D;
  void *qkv_q, *qkv_k, *qkv_v; /* bf16 */
} usdm_gemm_args;

int usdm_gemm(const usdm_gemm_args* args, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm / RMSNorm over channels of [rows][C] activations (one wave per row).
 *   t = x (+ res);  [premask: rows s >= valid_len[b] zeroed first];  sum32/sum16 <- t (optional)
 *   y = (t - mean) * rsqrt(var + eps) * gamma + beta     (rms: mean := 0, no beta)
 *   y = act(y);  rows s >= valid_len[b] are written as zero
 * Replaces nn.LayerNorm at networks.py:245-247,297,349; XLS-R LayerNorms; HF MistralRMSNorm.
 * round_bf16 = 1 reproduces HF's bf16 rounding points: bf16(gamma * bf16(t * rstd)), and rounds
 * t = x + res to bf16 (the bf16 residual stream of HF Mistral).
 * ---------------------------------------------------------------------------------------------- */
typedef struct usdm_norm_args {
  const void* x; int32_t x_dtype; int64_t ldx;
  const void* res; int32_t res_dtype; int64_t ldr;
  const float* gamma; const float* beta; float eps;
  int32_t rows, C;
  int32_t rms, act, round_bf16, premask;
  const int32_t* valid_len; int32_t rows_per_batch;
  void* out32; void* out16; int64_t ldo;
  void* sum32; void* sum16; int64_t lds;
} usdm_norm_args;

int usdm_norm(const usdm_norm_args* args, usdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* USDM_HIP_H_ */
