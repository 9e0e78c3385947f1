/*
 * libusdm_hip.so - EXPERIMENTAL entry points: real, tested code (tests/test_chain_gpu.py) that measured SLOWER than the
 * default path and is therefore not part of the stable C-ABI of include/usdm_hip.h.  Nothing in the product path calls
 * these unless an opt-in environment switch is set (USDM_GEMV_CHAIN, USDM_ATTN_MERGE_IN_OPROJ); they are kept because the
 * measurements they produced are on file (profiles/r02_decode_ablation.txt sections 1, 3, 4; profiles/r03_tp_ablation.txt)
 * and because the tensor-parallel shard shapes are where a persistent decode kernel could still win.
 * Signatures may change or disappear between rounds.
 */
#ifndef USDM_HIP_EXPERIMENTAL_H_
#define USDM_HIP_EXPERIMENTAL_H_
#include "usdm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Chained decode GEMVs in ONE persistent launch (round 2): up to 4 consecutive projections of the decode step, each consuming
 * the previous one's output vector (e.g. o_proj -> gate/up -> down_proj -> next layer's qkv), run by one resident grid.
 * Between two phases there is an all-to-all dependency (every workgroup needs the whole vector), i.e. a grid barrier; it is
 * hidden because the weights do not depend on the activations: every wave requests the first ring of its NEXT phase's weight
 * rows before it waits, so HBM keeps streaming across the phase boundary instead of draining and ramping up again as at a
 * launch boundary.  Per output row the arithmetic (lane partition of K, accumulation order, rounding points, RMSNorm / SwiGLU /
 * residual fusion) is that of usdm_gemv, bit for bit.
 *   ph[i]   : the usdm_gemv_args of phase i (plain or SwiGLU projections with y16 output; no lm_head / p2p / x_delta / merge
 *             modes; K a multiple of 512).  x of phase i+1 is normally y16 of phase i (or the residual stream it updated).
 *   sync    : 8 device words owned by the caller, zero-initialised once: [0] generation, [1] error, [2..4] arrival counters.
 *             Counters are monotonic (target = (generation + 1) * workgroups), so a captured hipGraph replays correctly
 *             without a memset node.
 *   Every wait is bounded (timeout_ms of the 100 MHz clock): on expiry the error word is set (USDM_CHAIN_ERR_TIMEOUT) and
 *   the kernel finishes with garbage instead of hanging; once set, later launches do not wait.  Needs all workgroups
 *   resident (2 x 448 threads per CU on the 256 CUs: nothing else may occupy the GPU for long). */
enum { USDM_CHAIN_MAX_PHASES = 4, USDM_CHAIN_ERR_TIMEOUT = 1 };
typedef struct usdm_gemv_chain_args {
  usdm_gemv_args ph[4];
  int32_t nph;
  uint32_t* sync;
  int32_t timeout_ms;
  uint64_t* gran;   /* usdm_gemv_engine only: 3 x 8192 eight-byte granules of hand-off space (192 KB), any content */
  int32_t norm_nth[4]; /* filled by the launcher: threads per workgroup of the usdm_gemv variant each phase would run with */
} usdm_gemv_chain_args;
int usdm_gemv_chain(const usdm_gemv_chain_args* args, usdm_stream_t stream);

/* The same chain on a loader / consumer ENGINE (round 2, second form): one 4-wave workgroup per CU; wave 0 only streams the
 * CU's share of every phase's weight rows into a 7 x 16 KiB LDS ring by LDS-DMA (buffer_load ... lds, non-temporal), running
 * ahead across phase boundaries as far as the ring allows; waves 1-3 take ring slots (FULL / FREE words in LDS), multiply
 * against the phase's input vector held in LDS and publish each output pair twice: as plain bf16 (for later launches) and as
 * an 8-byte granule {tag = epoch, 2 x bf16} that every CU's gathering wave sweeps into its LDS copy of the next phase's input
 * (the data is its own flag: no grid barrier, no counter).  Per row the arithmetic - lane partition of K, accumulation order,
 * RMSNorm partial-sum order of the equivalent usdm_gemv launch, rounding points - is that of usdm_gemv, bit for bit.
 * Shapes: every phase's output count a multiple of 512; K = 4096 (any phase) or 16 < K/512 <= 32 with K/512 even (plain
 * phases, e.g. 14336).  sync: as usdm_gemv_chain ([0] generation, [1] error); all waits bounded. */
int usdm_gemv_engine(const usdm_gemv_chain_args* args, usdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* USDM_HIP_EXPERIMENTAL_H_ */
