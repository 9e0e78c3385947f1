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
enum { USDM_ACT_NONE = 0, USDM_ACT_GELU = 1, USDM_ACT_SWIGLU = 3, USDM_ACT_TANH = 4, USDM_ACT_LOGCLAMP = 5 /* log(max(x,1e-5)) */ };
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
  /* split-K (single-tap, plain f32 output only): grid z is multiplied by split_k; split s accumulates its share of the K
   * chunks and stores to C32 + s*c_split_stride; bias and residual are applied by split 0 only.  The consumer sums the
   * partials (e.g. usdm_norm's `res` input).  For deep-K problems whose output tiles do not fill the chip. */
  int32_t split_k; int64_t c_split_stride;
  /* LayerNorm folded into the neighbouring GEMMs (post-LN block of networks.py:236-266: h1 = LN(h + attn) is consumed by the
   * feed-forward GEMM and by the next residual add, never on its own; saves the LayerNorm launch and its 22 MB round trip).
   *   stats_out  PRODUCER (row-major plain epilogue, 128-column tiles): after bias / residual, every tile writes the sum of its
   *              128 columns of each output row and their M2 = sum (v - tile mean)^2 to stats_out[row][tiles_n][2] (f32; no
   *              atomics, so the result is run-to-run reproducible).
   *   ln_mode 1  CONSUMER, A operand = the UN-normalised rows x (bf16), W = weights with gamma folded in along K:
   *              v = rstd[m] * acc[m][n] - rstd[m] * mean[m] * ln_c[n] + bias[n]   (ln_c[n] = sum_k W[n][k]; bias already holds
   *              b[n] + sum_k W0[n][k] beta[k]), i.e. exactly LN(x) W0^T + b, then the activation.  GELU bf16-out epilogue of the
   *              ping-pong tiles only.
   *   ln_mode 2  CONSUMER, residual[m][n] is UN-normalised x (f32): the residual added is LN(x)[m][n] =
   *              (x - mean[m]) * rstd[m] * ln_gamma[n] + ln_beta[n].  Row-major plain epilogue.
   * mean / rstd come from ln_stats[row][ln_nt][2]: per 128-column tile the producer's (sum, M2 about the tile's own mean), merged
   * pairwise-exactly (Chan et al.) over the ln_nt tiles of ln_C = 128 * ln_nt columns: no sum(x^2) - mean^2 cancellation.
   *   ln_guard   optional device word (ln_mode 1 and 2): OR-ed with 1 when a row has |mean| * rstd > ln_guard_ratio.  ln_mode 1
   *              multiplies rows that were rounded to bf16 before centring, so its operand noise relative to the normalised
   *              signal is 2^-9 * sqrt(1 + (mean / sigma)^2): the caller bounds the ratio it accepts and re-runs such inputs with
   *              the separate LayerNorm kernel (usdm_amd/voicebox/model/networks.py does). */
  float* stats_out;
  const float* ln_stats; int32_t ln_nt, ln_mode, ln_C; float ln_eps;
  const float* ln_c; const float* ln_gamma; const float* ln_beta;
  int32_t* ln_guard; float ln_guard_ratio;
  /* 0: the launcher picks the tile from the shape (measured heuristics, csrc/gemm.hip); t + 1: force tile t (benchmarks and the
   * tile-equivalence tests; the Python wrapper fills it from USDM_GEMM_TILE so the library itself reads no environment per launch) */
  int32_t tile_sel;
} usdm_gemm_args;

int usdm_gemm(const usdm_gemm_args* args, usdm_stream_t stream);
int usdm_gemm_tile_for(const usdm_gemm_args* args); /* the tile usdm_gemm would pick (12..14 = ping-pong tiles: the only ones with the
                                                        stats_out / ln_mode epilogues), < 0 for invalid arguments; launches nothing */

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
  const float* res2; int32_t n_res2; int64_t res2_stride;   /* n_res2 more f32 addends [rows][ldr], res2_stride elements apart (split-K partials): x + res + res2[0] + ... */
} usdm_norm_args;

int usdm_norm(const usdm_norm_args* args, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused anti-aliased SnakeBeta: Activation1d(SnakeBeta) of the reference in one pass
 * (vocoder/alias_free_torch/act.py:23-28, resample.py:25-33, filter.py:86-95,
 *  vocoder/activations.py:107-120).  x is channels-last f32 [T][ldx]; channels >= Creal are padding
 * and are written as zero.  fup/fdn: the 12 Kaiser-sinc taps of kaiser_sinc_filter1d(0.25,0.3,12)
 * (filter.py:28-57) computed on the host.  L: time steps per thread (0 = choose).
 * ---------------------------------------------------------------------------------------------- */
typedef struct usdm_snake_args {
  const float* x; int64_t ldx;
  int32_t T, C, Creal, L;
  const float* alpha; const float* beta; int32_t logscale;
  float fup[12]; float fdn[12];
  float* out32; void* out16; int64_t ldo;
} usdm_snake_args;
int usdm_aa_snake(const usdm_snake_args* args, usdm_stream_t stream);

/* (a+b+c)*scale over n f32 elements (vocoder/models.py:198-204). n % 4 == 0. */
int usdm_sum3_scale(const float* a, const float* b, const float* c, float scale, int64_t n,
                    float* out32, void* out16_bf16, usdm_stream_t stream);

/* channels-first f32 [B][C][T] -> channels-last [B][T][Cpad] (f32 and/or bf16), y = x*scale+shift,
 * padded channels zero.  Mel de-normalisation + layout change (model_util.py:103-104). */
int usdm_cf_to_cl(const float* x, int32_t B, int32_t C, int32_t T, int32_t Cpad, float scale, float shift,
                  float* out32, void* out16_bf16, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Flash-style attention forward (bf16 MFMA, fp32 online softmax), nothing [S,S]-sized in memory.
 *   mode 0: bidirectional, score += -slopes[h]*|i-j| (0 for key 0 if alibi_col0_zero), keys >= kv_len[b]
 *           masked  -> Voicebox Attention.forward + Transformer.forward bias (networks.py:162-210,319-341)
 *   mode 1: causal (key j visible iff j <= q_pos0 + i), GQA -> HF Mistral prefill attention
 * Layouts (bf16, element strides): q[b][h][s][dh] via q_bs/q_hs/q_rs; k likewise with Hkv heads;
 * vt = V^T [b][hk][dh][key] via v_bs/v_hs/v_ds; o[b][s][h*dh] via o_bs/o_rs.
 * K/V^T must be allocated and finite up to Skv_alloc >= roundup(Skv,64) keys.
 * ---------------------------------------------------------------------------------------------- */
typedef struct usdm_attn_args {
  int32_t mode, dh, B, Hq, Hkv, Sq, Skv, Skv_alloc, q_pos0, alibi_col0_zero;
  float scale;
  int32_t head_order;   /* mode 0, speed only: 0 = library default, 1 = steepest ALiBi heads first and last in dispatch order, -1 = flattest first */
  const void* q; int64_t q_bs, q_hs, q_rs;
  const void* k; int64_t k_bs, k_hs, k_rs;
  const void* vt; int64_t v_bs, v_hs, v_ds;
  void* o; int64_t o_bs, o_rs;
  const int32_t* kv_len;  /* [B] or NULL (= Skv) */
  const float* slopes;    /* [Hq] or NULL */
  int32_t window;         /* mode 1: > 0 = sliding window, query at position p sees keys p-window+1 .. p (HF Mistral sliding_window,
                             src/model.py:337-371 keeps W - 1 past keys + the new one = the HF eager mask; the reference's flash-attn call
                             passes window_size = (W, W), src/model.py:510,532, which admits W + 1 keys in a prefill: the two reference
                             paths themselves differ by one key beyond 4096 tokens, and this follows the first); 0 = full causal */
  int32_t variant;        /* mode 0, d = 64, speed only: 0 = 16-query waves (default), 1 = the 32-query-wave kernel (benchmarks; results agree
                             to one bf16 ulp).  The library reads no environment variable per launch. */
} usdm_attn_args;
int usdm_attention(const usdm_attn_args* args, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Voicebox glue (dense math is usdm_gemm / usdm_attention / usdm_norm).
 * ---------------------------------------------------------------------------------------------- */
/* Estimator input assembly: embed(x)*sqrt(E) ++ y ++ cond, channels-last bf16 rows
 * (networks.py:305-307), with the classifier-free-guidance batch doubling of voicebox.py:60-65
 * done in place when dup == 2 (first B_in rows: null token, zero cond). */
typedef struct usdm_vb_input_args {
  const int64_t* ids;  /* [B_in][S] */
  const float* y;      /* [B_in][F][S] */
  const float* cond;   /* [B_in][F][S] */
  const void* table;   /* bf16 [n_tokens+1][E], already multiplied by sqrt(E) */
  int32_t B_in, dup, S, E, F, null_id, use_cond;
  void* out; int64_t ldo; /* [B_in*dup][S][ldo] of out_dtype */
  int32_t out_dtype;      /* USDM_BF16 (default plan: MFMA operand rows, table bf16) or USDM_F32 (exact-f32 plan: table f32) */
} usdm_vb_input_args;
int usdm_vb_build_input(const usdm_vb_input_args* args, usdm_stream_t stream);

/* Attention probabilities of the exact-f32 Voicebox plan (the reference runs fp32: networks.py:162-210): in place on the f32 score
 * matrix x[row][h*ldseg + j] = q_i.k_j (row = b*rows_per_batch + i): p = softmax_j(x - slopes[h]*|i-j|, key 0 unbiased when
 * col0_zero) over keys j < kv_len[b]; masked keys -> 0, pad columns [n, npad) -> 0 (networks.py:319-341). */
int usdm_softmax_alibi(float* x, int32_t rows, int32_t rows_per_batch, int32_t nheads, int32_t n, int32_t npad, int64_t ldrow,
                       int32_t ldseg, const float* slopes, const int32_t* kv_len, int32_t col0_zero, usdm_stream_t stream);

/* Sinusoidal time token (SinusoidalPosEmb, networks.py:13-28) into row 0 of each batch of h:
 * [sin(1000*t*freqs), cos(1000*t*freqs)], freqs[H/2] = exp(arange(H/2) * -ln(1e4)/(H/2-1)) from the host. */
int usdm_vb_time_token(const float* t, int32_t t_stride, const float* freqs, int32_t Bx, int32_t H,
                       int64_t rows_per_batch, float* h32, void* h16_bf16, usdm_stream_t stream);

/* One elementwise solver update (voicebox.py:66-72 CFG, :83-90 Euler, :112-131 Heun):
 *   v = cfg ? vc + gs*(vc - vu) : vout
 *   mode 0: v1 <- v ; zn = z + dt*v          mode 1: zn = z + dt*(v1 + v)/2
 *   if eps: zn[.., s < P] = c_eps*eps + c_cond*cond   (prompt re-noising, :115-117,:126-128)
 *   z_in <- zn (estimator input), z_commit <- zn (solver state), t_cur[0..t_count) <- t_next */
typedef struct usdm_vb_solver_args {
  const float* vout;  /* [B*(cfg?2:1)][F][S] */
  const float* z; float* v1; const float* eps; const float* cond;
  float* z_in; float* z_commit; float* t_cur;
  int32_t B, F, S, P, cfg, mode, t_count;
  float gs, dt, c_eps, c_cond, t_next;
} usdm_vb_solver_args;
int usdm_vb_solver_step(const usdm_vb_solver_args* args, usdm_stream_t stream);

/* Padding mask of ragged Voicebox batches (networks.py:330-333 and the `* y_mask` products): zero every time step
 * t >= valid_len[b] - off of x; layout 0 = [B][T][C] (channels-last), 1 = [B][C][T] (channels-first). */
int usdm_mask_time(float* x32, void* x16_bf16, int32_t B, int32_t T, int32_t C, int32_t layout, const int32_t* valid_len,
                   int32_t off, usdm_stream_t stream);

/* device-to-device async copy (graph-capturable plumbing) */
int usdm_copy_bytes(void* dst, const void* src, int64_t nbytes, usdm_stream_t stream);

/* process_unit (util/model_util.py:50-54): out[f] = mode of repeat_interleave(units, rep)[f*hop:(f+1)*hop],
 * ties -> smallest id; nframes = floor(n*rep/hop).  int64 in / int64 out. */
int usdm_process_unit(const int64_t* units, int32_t n, int32_t rep, int32_t hop, int64_t* out, int32_t nframes,
                      usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Mistral-7B speech-text LLM (third-party arithmetic of the reference: HF transformers
 * MistralForCausalLM + GenerationMixin.generate, call sites src/inference.py:63-83, load :116-124).
 * Prefill uses usdm_gemm / usdm_norm / usdm_attention(mode 1); the kernels below are the rest.
 * ---------------------------------------------------------------------------------------------- */
/* Batch-1 GEMV y = W x over bf16 weights [N][ldw] (nn.Linear layout), HBM-streaming:
 *   norm_w != NULL : x is the raw residual stream; RMSNorm(x)*norm_w is applied first (HF rounding)
 *   act == USDM_ACT_SWIGLU : rows packed in blocks of 32 = 16 gate + 16 up; output N/2 = silu(g)*u
 *   residual       : y += residual[n] (bf16)             round_bf16 : HF bf16 rounding points
 *   part_val/idx   : lm_head mode — per-block (max, argmax) of the bf16-rounded logits with ban[n]==1
 *                    excluded (bad_words_ids of inference.py:51-53); y32 (optional) receives the logits */
typedef struct usdm_gemv_args {
  const void* W; int64_t ldw; int32_t N, K;
  const void* x;
  const float* norm_w; float eps;
  int32_t act, round_bf16;
  const void* residual;
  void* y16; float* y32;
  const uint8_t* ban; float* part_val; int32_t* part_idx;
  int32_t idx_offset; /* added to the row index stored in part_idx (vocab-parallel shards) */
  /* tensor-parallel decode: the all-reduced f32 partial sum of the previous row-parallel projection is folded in here
   * instead of a separate residual-add launch: x' = bf16(x + bf16(x_delta)) is what gets normalised / multiplied, and
   * workgroup 0 writes x' to x_out (a DIFFERENT buffer than x: other workgroups are still reading x) */
  const float* x_delta; void* x_out;
  const int32_t* skip;  /* optional: *skip != 0 -> the launch returns immediately (see usdm_decode_state.done) */
  /* tensor-parallel decode over xGMI peer-to-peer (usdm_allreduce_p2p_*, below): this launch is a ROW-parallel projection
   * whose f32 partial sums are all-reduced INSIDE its epilogue.  p2p_mode 1 (fused): the workgroup that owns output rows
   * [r0, r1) writes its partials as {epoch, f32} granules into slot[site][rank] of EVERY rank's buffer, waits (bounded) for
   * the same rows from every rank in its own buffer, sums them in rank order and stores y16[n] = bf16(residual[n] +
   * bf16(sum)): the all-reduce, the residual add and the bf16 rounding of HF's residual stream in one epilogue, no
   * collective launch.  p2p_mode 2 (split): write only; usdm_allreduce_p2p_reduce finishes.  Needs residual and y16. */
  const struct usdm_p2p_dev* p2p; int32_t p2p_site, p2p_mode;
  /* EXPERIMENTAL, off by default (measured slower: profiles/r02_decode_ablation.txt section 1; see usdm_hip_experimental.h).
   * o_proj of the decode step: x is not read from memory but MERGED here, in the x-staging prologue, from the context-split
   * partials usdm_attn_decode left (defer_merge): x[h*128+d] = bf16( sum_s po[h][s][d]*w_s / sum_s pl[h][s]*w_s ),
   * w_s = exp(pm[h][s] - max_s pm[h][s]).  Replaces the separate combine launch (the prologue runs while this launch's first
   * weight loads are in flight).  K = heads*128; mrg_ns splits per head. */
  const float* mrg_pm; const float* mrg_pl; const float* mrg_po; int32_t mrg_ns;
  /* o_proj of the decode step, hand-off form (round 3): with cmb_gran the mrg_* partials are combined ONCE per head, by the
   * first K/128 workgroups of THIS launch (the arithmetic of the combine kernel, bit for bit), and handed to all workgroups as
   * 8-byte granules {tag 1, 2 x bf16} in cmb_gran[K/2] (device memory; its tags must be zero when the launch starts:
   * usdm_attn_decode(cmb_gran) clears them).  Every workgroup requests its first weight ring BEFORE it waits for the granules,
   * so the combine runs under the weight latency instead of in a launch of its own.  The wait is bounded (cmb_timeout_ms of the
   * 100 MHz clock); on expiry the workgroup ORs 1 into *cmb_err and carries on with zeros.  N = 16 * 256 outputs only (one
   * 16-wave workgroup per CU, all co-resident): the hand-off ASSUMES that workgroups 0 .. K/128-1 are dispatched and make progress
   * while the others poll, which holds exactly because the whole grid is resident at once; a launch for which it did not hold
   * would cost one timeout per layer (and raise through *cmb_err), not hang. */
  unsigned long long* cmb_gran; int32_t* cmb_err; int32_t cmb_timeout_ms;
} usdm_gemv_args;
int usdm_gemv(const usdm_gemv_args* args, usdm_stream_t stream);
int usdm_gemv_nblocks(int32_t N, int32_t act); /* number of partials the lm_head mode writes */

int usdm_gemv_threads(const usdm_gemv_args* args);   /* threads per workgroup usdm_gemv would launch this projection with */

/* Batched decode (the serving path's request lists, src/inference_vllm.py:109-125): the same projection over nb
 * input vectors, weights streamed ONCE per step.  g holds item 0's pointers; item b is at + b * stride.  Two forms:
 *   VALU (nb <= 4)          v_dot2 per sequence; per item the arithmetic of usdm_gemv, bit for bit.
 *   matrix cores (nb <= 16) the weights as the A operand of v_mfma_f32_16x16x32_bf16 (16 rows x 32 K per instruction, loaded from
 *                           HBM straight into the fragment layout), the nb activation vectors as B: the step costs what the batch-1
 *                           step costs in HBM time for any nb <= 16.  Same rounding points (RMSNorm, bf16 outputs, SwiGLU,
 *                           residual, bf16 logits), but K is summed in a different order than usdm_gemv: equal to f32 rounding of
 *                           the accumulation, not bit for bit (csrc/llm_mfma_k.hip).
 * form: 0 = VALU for nb <= 4, matrix cores above; 1 = matrix cores; -1 = VALU (nb <= 4 only); 3 = matrix cores with
 * fragment-shaped weight loads (A/B of the load pattern); 5 = matrix cores without the workgroup-level K split. */
typedef struct usdm_gemv_batch_args {
  usdm_gemv_args g;          /* x_delta / x_out must be NULL */
  int32_t nb;
  int64_t x_bs, y_bs, res_bs; /* element strides between items of x, y16 / y32, residual */
  int32_t part_bs;            /* element stride between items of part_val / part_idx (>= usdm_gemv_nblocks; the matrix-core form
                                 writes one partial per workgroup (<= 256) and fills the rest with "no candidate") */
  int32_t form;
  /* Matrix-core form, K split over workgroups (round 4; K > 4096 = down_proj): with ks_part / ks_cnt given and K %% 2048 == 0 (no
   * RMSNorm, SwiGLU or lm_head mode) workgroup (s, j) multiplies K slice s (2048 wide, its activation slice HELD in registers) of its
   * 16-row tiles; the K / 2048 partial tiles meet in ks_part (write-through stores), the workgroup that arrives last at a tile's
   * counter sums them in slice order (deterministic) and runs the epilogue.  ks_part: usdm_gemv_batch_ks_floats(N, K) floats;
   * ks_cnt: ceil(N / 16) int32, ZERO when the launch starts (every launch leaves them zero).  form 5 = do not split. */
  float* ks_part; int32_t* ks_cnt; int64_t ks_part_floats;
} usdm_gemv_batch_args;
int64_t usdm_gemv_batch_ks_floats(int32_t N, int32_t K);   /* 0: this shape is not split */
int usdm_gemv_batch(const usdm_gemv_batch_args* args, usdm_stream_t stream);

/* Device-resident greedy-decode state so that a decode step is replayable as one hipGraph. */
typedef struct usdm_decode_state {
  int32_t* next_token;  /* [1] token fed to the next step                     */
  int32_t* out_tokens;  /* [max_out] generated ids                            */
  int32_t* step;        /* [1] number of generated tokens so far              */
  int32_t* pos;         /* [1] number of tokens in the KV cache               */
  int32_t max_out, id_offset, advance_pos;
  int32_t batch;        /* batched decode: 0 or 1 = single; else arrays of `batch` items: next_token[b], step[b], pos[b],
                           out_tokens[b][max_out] */
  /* device-side end of sequence (optional; all device memory so that a captured graph follows per-call settings):
   * eos = { n_eos (<= 6), min_new, id0, id1, ... }.  Once the picked token is one of the ids and at least min_new tokens
   * exist, done[b] is set; later calls with done[b] != 0 return without touching the state.  The decode kernels take
   * the same word as `skip` and return at once, so steps launched past the EOS cost launch overhead only. */
  int32_t* done; const int32_t* eos;
} usdm_decode_state;
/* arg-max over the per-block partials (ties -> lowest id = torch.argmax on the masked logits; with
 * do_sample=True, top_k=1 the reference samples among exact ties, of which this is one outcome). */
int usdm_argmax_final(const float* part_val, const int32_t* part_idx, int32_t nparts,
                      const usdm_decode_state* st, const void* embed_table_bf16, int32_t Hd, void* h_out_bf16,
                      usdm_stream_t stream);
/* The same pick over nseg SEGMENTS of nparts partials per sequence, seg_stride elements apart (sequence b's partials of segment s
 * start at s * seg_stride + b * nparts): the tensor-parallel batched decode step all-gathers the ranks' [sequences][nparts] blocks
 * into [rank][sequences][nparts] (SURVEY.md 8e + 8f-2). */
int usdm_argmax_final_seg(const float* part_val, const int32_t* part_idx, int32_t nparts, int32_t nseg, int64_t seg_stride,
                          const usdm_decode_state* st, const void* embed_table, int32_t Hd, void* h_out, usdm_stream_t stream);  /* embed_table != NULL: also h_out = table[token] (next step's input);
                                                 batched state: part_val/part_idx are [batch][nparts], h_out [batch][Hd] */

/* Sampling twin of usdm_argmax_final: temperature -> top-k -> top-p (HF TemperatureLogitsWarper, TopKLogitsWarper,
 * TopPLogitsWarper: generate(do_sample=True, top_k, top_p, temperature) of src/inference.py:63-83 with the knobs the demo
 * exposes, streamlit_demo.py:201-211) and one multinomial draw with Philox4x32-10(seed, counter = *st->step).
 * logits: f32 [V] of this step with banned ids = -inf (usdm_gemv lm_head mode writes exactly that into y32).
 * top_k = 0 and top_p = 1 switch the filters off.  Ties at a filter boundary are kept or dropped as a block (HF's
 * unstable sort picks arbitrarily).  probs_out (optional, [V]) receives the filtered, renormalised distribution.
 * If no id has positive mass (every logit banned or NaN) the arg-max of the finite logits is taken, else id 0: the
 * kernel never emits an id outside [0, V). */
typedef struct usdm_sample_params {   /* the per-request knobs (vLLM SamplingParams of inference_vllm.py:42-66) */
  float temperature; int32_t top_k; float top_p; int32_t reserved;
  uint64_t seed;
} usdm_sample_params;
typedef struct usdm_sample_args {
  const float* logits; int32_t V;
  float temperature; int32_t top_k; float top_p;
  uint64_t seed;
  float* probs_out;
  /* optional DEVICE copy of the knobs: when non-NULL it overrides temperature / top_k / top_p / seed above, so that one
   * captured decode graph serves every request (the host rewrites 24 bytes instead of re-capturing per seed). */
  const usdm_sample_params* dev_params;
  /* batched decode (usdm_decode_state.batch > 1): sequence b reads logits + b * logits_bs, dev_params[b] (required), the state
   * words [b] and writes h_out + b * Hd; its Philox counter is ITS step, so its tokens equal those of the single-sequence call
   * (per-slot sampling inside a continuous batch: src/inference_vllm.py:109-123 passes one SamplingParams per request). */
  int64_t logits_bs;
} usdm_sample_args;
int usdm_sample_final(const usdm_sample_args* args, const usdm_decode_state* st, const void* embed_table_bf16, int32_t Hd,
                      void* h_out_bf16, usdm_stream_t stream);

/* out[r][:] = table[ids[r]][:] (bf16 rows; ids == NULL -> single row from *next_token) */
int usdm_embed_rows(const void* table, const int64_t* ids, const int32_t* next_token, int32_t n, int32_t Hd,
                    void* out, usdm_stream_t stream);

/* Prefill: HF apply_rotary_pos_emb (bf16 rounding) in place on q,k of qkv[S][(Hq+2Hkv)*128]; roped K and
 * V appended to the caches [Hkv][ctx_max][128]; V also written transposed to vt[Hkv][128][vt_ld] for
 * usdm_attention.  cos/sin: bf16 [max_pos][64] built on the host as HF's MistralRotaryEmbedding does. */
typedef struct usdm_rope_args {
  void* qkv; int64_t ld; int32_t S, pos0, Hq, Hkv, ctx_max, max_pos;
  const uint16_t* cos; const uint16_t* sin;
  void* kcache; void* vcache; void* vt; int64_t vt_ld;
} usdm_rope_args;
int usdm_rope_cache(const usdm_rope_args* args, usdm_stream_t stream);

/* Decode attention for ONE new token (GQA, head_dim 128), context split over NS workgroups per kv head
 * + combine.  Ropes q/k of qkv[(Hq+2Hkv)*128] at position *pos, appends K,V to the caches, writes
 * out[Hq*128] bf16.  Softmax in fp32, P rounded to bf16 for PV (flash-attention-2 semantics). */
typedef struct usdm_attn_decode_args {
  const void* qkv; const int32_t* pos;
  int32_t Hq, Hkv, ctx_max, NS; float scale;
  const uint16_t* cos; const uint16_t* sin;
  void* kcache; void* vcache;
  float* pm; float* pl; float* po; /* scratch [Hq][NS], [Hq][NS], [Hq][NS][128] */
  void* out;
  int32_t* counters; /* [Hkv] zero-initialised once, self-resetting: when given (NS > 1) the workgroup that finishes a
                        kv head last merges its NS partials itself and no separate combine kernel is launched */
  /* batched decode: `batch` sequences in one launch (0 or 1 = single).  Item b reads pos[b], qkv + b*qkv_bs, caches +
   * b*cache_bs, writes out + b*out_bs (element strides); scratch is [batch][Hq][NS]...; counters [batch][Hkv]. NS > 1. */
  int32_t batch; int64_t qkv_bs, out_bs, cache_bs;
  const int32_t* skip;  /* optional (single-sequence form): *skip != 0 -> return immediately */
  int32_t defer_merge;  /* NS > 1, no counters: leave the partials (pm, pl, po) for the consumer (usdm_gemv mrg_*): no combine
                           launch, `out` is not written */
  int32_t window;       /* > 0: sliding window, the token at *pos sees keys pos-window+1 .. pos only (the NS splits divide that
                           range); 0 = the whole cache */
  unsigned long long* cmb_gran;  /* optional: the Hq*64 granules of usdm_gemv's cmb_gran hand-off, cleared by this launch */
} usdm_attn_decode_args;
int usdm_attn_decode(const usdm_attn_decode_args* args, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One-shot peer-to-peer all-reduce for the tensor-parallel decode step (replaces the single-GPU
 * model.generate of src/inference.py:116-123 when the 7B is sharded over the 8 GPUs of a node).  Messages are 4096 f32
 * (16 KB): pure latency, so there is no ring and no collective kernel: every rank WRITES its partial rows straight into
 * every peer's buffer over xGMI (hipIpc-mapped, uncached device memory) as 8-byte granules {tag = epoch, f32 value} -
 * the data is its own flag (one aligned 8-byte system-scope store is never torn), so no fence, no counter and no
 * separate flag store are needed; the reader polls its OWN memory.
 *
 *   buffer of one rank:  [256-B header: epoch u32, err u32][parity 2][site n_sites][src rank 8][max_elems] granules
 *   epoch   : device word, starts at 1, +1 per decode step (usdm_argmax_p2p), so a captured hipGraph replays correctly
 *   parity  : epoch & 1 selects the half; together with the all-to-all of usdm_argmax_p2p once per token no slot is
 *             rewritten before every reader has left it
 *   waits   : every poll is BOUNDED (timeout_ms of the 100 MHz wall clock); on expiry the kernel ORs a code into the
 *             err word - its own AND every peer's (| USDM_P2P_ERR_PEER) - and carries on with zeros; once err != 0 no
 *             kernel waits again, so a protocol bug costs one timeout and surfaces on EVERY rank as
 *             usdm_allreduce_p2p_error() != 0 - never as a hang, never as silently diverging ranks
 *   order   : partials are summed in rank order 0..world-1 on every rank -> bit-identical results on all ranks
 * Host protocol: create -> export (64-byte hipIpcMemHandle) -> exchange handles by any host channel -> import each
 * peer (other process) or attach (same process, logical ranks) -> commit.  RCCL stays the prefill / validation path.
 * ---------------------------------------------------------------------------------------------- */
enum { USDM_P2P_MAX_RANKS = 8, USDM_P2P_HANDLE_BYTES = 64, USDM_P2P_HEADER_BYTES = 256 };
enum { USDM_P2P_ERR_TIMEOUT_ROWS = 1, USDM_P2P_ERR_TIMEOUT_PICK = 2, USDM_P2P_ERR_TIMEOUT_REDUCE = 4,
       USDM_P2P_ERR_PEER = 8 /* set by ANOTHER rank that timed out (with its code): results may have diverged there */ };
typedef struct usdm_p2p_dev {          /* device-visible view (device memory, constant after commit) */
  uint64_t base[8];                    /* address of rank r's buffer as mapped in THIS process; base[rank] = local */
  int32_t rank, world, n_sites, max_elems;
  uint64_t timeout_ticks;              /* 100 MHz ticks */
} usdm_p2p_dev;
typedef struct usdm_p2p usdm_p2p;      /* opaque host handle */
int64_t usdm_allreduce_p2p_bytes(int32_t n_sites, int32_t max_elems);   /* buffer size of one rank (layout above) */
int usdm_allreduce_p2p_create(int32_t rank, int32_t world, int32_t n_sites, int32_t max_elems, int32_t timeout_ms,
                              usdm_p2p** out);
int usdm_allreduce_p2p_export(const usdm_p2p* c, void* handle64);                    /* host bytes out */
int usdm_allreduce_p2p_import(usdm_p2p* c, int32_t peer, const void* handle64);      /* peer lives in another process */
int usdm_allreduce_p2p_attach(usdm_p2p* c, int32_t peer, void* peer_local_base);     /* peer lives in this process */
void* usdm_allreduce_p2p_base(const usdm_p2p* c);                                    /* this rank's buffer */
int usdm_allreduce_p2p_commit(usdm_p2p* c, usdm_stream_t stream);                    /* all peers known: publish the device view */
const usdm_p2p_dev* usdm_allreduce_p2p_dev(const usdm_p2p* c);                       /* device pointer for kernel args */
int usdm_allreduce_p2p_error(const usdm_p2p* c, int32_t* host_err, int32_t* host_epoch);  /* synchronous 8-byte read */
int usdm_allreduce_p2p_destroy(usdm_p2p* c);
/* split mode, second half (and the validation form): h[n] = bf16(h[n] + bf16(sum_r slot[site][r][n])), n < n_elems */
int usdm_allreduce_p2p_reduce(const usdm_p2p_dev* dev, int32_t site, int32_t n_elems, void* h_bf16, const int32_t* skip,
                              usdm_stream_t stream);
/* Vocab-parallel token pick: arg-max over this rank's lm_head partials, the (value, id) pair exchanged with every rank
 * through `site`, global arg-max (ties -> lowest id) identical on every rank, decode state advanced as
 * usdm_argmax_final does, epoch += 1.  One launch replaces all_gather x2 + usdm_argmax_final.
 * phase 0: put + get in one launch; phase 1 / 2: the put half / the get half as separate launches (split form). */
int usdm_argmax_p2p(const float* part_val, const int32_t* part_idx, int32_t nparts, const usdm_decode_state* st,
                    const usdm_p2p_dev* dev, int32_t site, int32_t phase, const void* embed_table_bf16, int32_t Hd,
                    void* h_out_bf16, usdm_stream_t stream);

/* h = bf16(h + bf16(delta)) : residual add after a tensor-parallel all-reduce of f32 partial sums */
int usdm_residual_add(void* h_bf16, const float* delta, int32_t n, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * XLS-R / k-means unit extractor (third-party seamless_communication UnitExtractor.predict, call sites
 * src/inference.py:59,111-113, util/model_util.py:79-80).  fp32 throughout; GEMMs are usdm_gemm(F32).
 * ---------------------------------------------------------------------------------------------- */
/* y = layer_norm(x) over all n samples (F.layer_norm(wave, wave.shape)) */
int usdm_wave_layernorm(const float* x, int32_t n, float eps, float* y, usdm_stream_t stream);
/* first conv layer fused: Conv1d(1->512,k=10,stride) + LayerNorm(512) + GELU -> channels-last [T][512] */
int usdm_w2v_conv0(const float* x, int32_t n, int32_t T, int32_t C, int32_t k, int32_t stride, const float* w,
                   const float* b, const float* ln_g, const float* ln_b, float eps, float* out, usdm_stream_t stream);
/* in-place softmax of x[row][seg*ldseg + 0..n), pad columns [n,npad) zeroed (attention probabilities) */
int usdm_softmax_segments(float* x, int32_t rows, int32_t nseg, int32_t n, int32_t npad, int64_t ldrow, int32_t ldseg,
                          usdm_stream_t stream);
/* ids[t] = argmin_n(|x_t|^2 - 2*dots[t][n] + csq[n]) (first minimum); margin[t] = runner-up - best (optional) */
int usdm_kmeans_argmin(const float* x, int32_t T, int32_t D, const float* dots, int64_t ldd, const float* csq,
                       int32_t n_units, int64_t* ids, float* margin, usdm_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Mel front end of the speech prompt: get_mel / mel_spectrogram (util/model_util.py:24-38,
 * vocoder/meldataset.py:55-78).  frames = Hann-windowed, reflect-padded (pad = (n_fft-hop)/2), clamped
 * signal [T][n_fft]; the DFT is usdm_gemm(F32) against a host-built [2*nbins][n_fft] cos/-sin matrix;
 * usdm_stft_mag = sqrt(re^2+im^2+eps); mel projection + log(clamp(.,1e-5)) = usdm_gemm with
 * USDM_ACT_LOGCLAMP and transpose_out.  Sample-rate conversion is a polyphase FIR, also a usdm_gemm.
 * ---------------------------------------------------------------------------------------------- */
int usdm_stft_frames(const float* x, int32_t n, int32_t n_fft, int32_t hop, int32_t pad, const float* window,
                     float* frames, int32_t T, usdm_stream_t stream);
/* frames[t][c] = x[t*hop + c - offset], zero outside the signal (im2col for the polyphase resampler GEMM) */
int usdm_frame_signal(const float* x, int32_t n, int32_t frame_len, int32_t hop, int32_t offset, float* frames, int32_t T,
                      usdm_stream_t stream);
int usdm_stft_mag(const float* re_im, int64_t ld, int32_t T, int32_t nbins, float eps, float* out, int64_t ldo,
                  int32_t nbins_pad, usdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* USDM_HIP_H_ */
