// Host side + standalone kernels of the one-shot peer-to-peer all-reduce (protocol: include/usdm_hip.h).
// The fused form lives in the row-parallel GEMV's epilogue (llm_k.hip, gemv_kernel p2p_mode 1).
#include "p2p.h"
#include <stdlib.h>

struct usdm_p2p {
  int rank, world, n_sites, max_elems;
  int64_t bytes;
  void* local;                       // this rank's buffer (uncached device memory)
  void* peer[USDM_P2P_MAX_RANKS];    // mapped bases (peer[rank] = local)
  bool opened[USDM_P2P_MAX_RANKS];   // mapped through hipIpcOpenMemHandle (must be closed)
  usdm_p2p_dev host_view;
  usdm_p2p_dev* dev_view;            // device copy handed to kernels
  bool committed;
};

extern "C" int64_t usdm_allreduce_p2p_bytes(int32_t n_sites, int32_t max_elems) {
  return (int64_t)USDM_P2P_HEADER_BYTES + (int64_t)2 * n_sites * USDM_P2P_MAX_RANKS * max_elems * (int64_t)sizeof(p2p_gran);
}

extern "C" int usdm_allreduce_p2p_create(int32_t rank, int32_t world, int32_t n_sites, int32_t max_elems, int32_t timeout_ms,
                                         usdm_p2p** out) {
  USDM_CHECK_ARG(out && world >= 1 && world <= USDM_P2P_MAX_RANKS && rank >= 0 && rank < world, "usdm_allreduce_p2p_create: rank/world (<= 8)");
  USDM_CHECK_ARG(n_sites > 0 && max_elems >= 2 && max_elems % 2 == 0 && timeout_ms > 0, "usdm_allreduce_p2p_create: n_sites / max_elems / timeout_ms");
  usdm_p2p* c = (usdm_p2p*)calloc(1, sizeof(usdm_p2p));
  USDM_CHECK_ARG(c, "usdm_allreduce_p2p_create: out of host memory");
  c->rank = rank; c->world = world; c->n_sites = n_sites; c->max_elems = max_elems;
  c->bytes = usdm_allreduce_p2p_bytes(n_sites, max_elems);
  // Uncached device memory: peers write it over xGMI behind the back of this GPU's L2s, so no line of it may live in a cache.
  hipError_t e = hipExtMallocWithFlags(&c->local, c->bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { usdm_set_error("usdm_allreduce_p2p_create: allocation of %lld bytes -> %s", (long long)c->bytes, hipGetErrorString(e)); free(c); return 1; }
  if (hipMemset(c->local, 0, c->bytes) != hipSuccess || hipMalloc((void**)&c->dev_view, sizeof(usdm_p2p_dev)) != hipSuccess) {
    usdm_set_error("usdm_allreduce_p2p_create: memset / view allocation failed");
    (void)hipFree(c->local); free(c); return 1;
  }
  const unsigned one = 1;   // epoch starts at 1: a zero-initialised tag never matches
  if (hipMemcpy(c->local, &one, 4, hipMemcpyHostToDevice) != hipSuccess) {
    usdm_set_error("usdm_allreduce_p2p_create: epoch init failed");
    (void)hipFree(c->dev_view); (void)hipFree(c->local); free(c); return 1;
  }
  c->peer[rank] = c->local;
  memset(&c->host_view, 0, sizeof(c->host_view));
  c->host_view.rank = rank; c->host_view.world = world; c->host_view.n_sites = n_sites; c->host_view.max_elems = max_elems;
  c->host_view.timeout_ticks = (uint64_t)timeout_ms * 100000ull;   // 100 MHz
  *out = c;
  return 0;
}

extern "C" int usdm_allreduce_p2p_export(const usdm_p2p* c, void* handle64) {
  USDM_CHECK_ARG(c && handle64, "usdm_allreduce_p2p_export: null");
  static_assert(sizeof(hipIpcMemHandle_t) <= USDM_P2P_HANDLE_BYTES, "handle size");
  hipIpcMemHandle_t h;
  USDM_HIP(hipIpcGetMemHandle(&h, c->local));
  memset(handle64, 0, USDM_P2P_HANDLE_BYTES);
  memcpy(handle64, &h, sizeof(h));
  return 0;
}

extern "C" int usdm_allreduce_p2p_import(usdm_p2p* c, int32_t peer, const void* handle64) {
  USDM_CHECK_ARG(c && handle64 && peer >= 0 && peer < c->world && peer != c->rank && !c->peer[peer], "usdm_allreduce_p2p_import: bad peer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* p = nullptr;
  USDM_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  c->peer[peer] = p; c->opened[peer] = true;
  return 0;
}

extern "C" int usdm_allreduce_p2p_attach(usdm_p2p* c, int32_t peer, void* base) {
  USDM_CHECK_ARG(c && base && peer >= 0 && peer < c->world && peer != c->rank && !c->peer[peer], "usdm_allreduce_p2p_attach: bad peer");
  c->peer[peer] = base; c->opened[peer] = false;
  return 0;
}

extern "C" void* usdm_allreduce_p2p_base(const usdm_p2p* c) { return c ? c->local : nullptr; }

extern "C" int usdm_allreduce_p2p_commit(usdm_p2p* c, usdm_stream_t stream) {
  USDM_CHECK_ARG(c, "usdm_allreduce_p2p_commit: null");
  for (int r = 0; r < c->world; ++r) {
    USDM_CHECK_ARG(c->peer[r], "usdm_allreduce_p2p_commit: rank %d has no mapping (import / attach every peer first)", r);
    c->host_view.base[r] = (uint64_t)c->peer[r];
  }
  USDM_HIP(hipMemcpy(c->dev_view, &c->host_view, sizeof(usdm_p2p_dev), hipMemcpyHostToDevice));
  (void)stream;
  c->committed = true;
  return 0;
}

extern "C" const usdm_p2p_dev* usdm_allreduce_p2p_dev(const usdm_p2p* c) { return (c && c->committed) ? c->dev_view : nullptr; }

extern "C" int usdm_allreduce_p2p_error(const usdm_p2p* c, int32_t* host_err, int32_t* host_epoch) {
  USDM_CHECK_ARG(c, "usdm_allreduce_p2p_error: null");
  unsigned w[2] = {0, 0};
  USDM_HIP(hipMemcpy(w, c->local, 8, hipMemcpyDeviceToHost));
  if (host_epoch) *host_epoch = (int32_t)w[0];
  if (host_err) *host_err = (int32_t)w[1];
  return 0;
}

extern "C" int usdm_allreduce_p2p_destroy(usdm_p2p* c) {
  if (!c) return 0;
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r] && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  if (c->dev_view) (void)hipFree(c->dev_view);
  if (c->local) (void)hipFree(c->local);
  free(c);
  return 0;
}

namespace {
// split mode, second half: one thread per element polls the world granules of its element (own buffer), sums in rank order
__global__ __launch_bounds__(256) void p2p_reduce_kernel(const usdm_p2p_dev* d, int site, int n, bf16_t* h, const int* skip) {
  if (skip && *skip) return;
  const unsigned epoch = p2p_load_epoch(d);
  const bool failed = p2p_load_err(d) != 0;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool want = i < n;
  float s = 0.f;
  for (int r = 0; r < d->world; ++r)
    s += p2p_get(d, p2p_slot(d, d->rank, epoch, site, r) + (want ? i : 0), epoch, want, USDM_P2P_ERR_TIMEOUT_REDUCE, failed);
  if (want) h[i] = f2bf(bf2f(h[i]) + round_bf(s));
}

// token pick across ranks (see usdm_argmax_p2p in the header)
__global__ __launch_bounds__(256) void argmax_p2p_kernel(const float* pv, const int* pi, int nparts, usdm_decode_state st,
                                                         const usdm_p2p_dev* d, int site, int phase, const bf16_t* E, int Hd,
                                                         bf16_t* h_out) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int s_tok;
  if (st.done && st.done[0]) return;
  const unsigned epoch = p2p_load_epoch(d);
  const bool failed = p2p_load_err(d) != 0;
  const int tid = threadIdx.x;
  if (phase != 2) {   // local arg-max over this rank's lm_head partials
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < nparts; i += 256) {
      const float v = pv[i];
      const int id = pi[i];
      if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
    }
    sv[tid] = bv; si[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const float v = sv[tid + s];
        const int id = si[tid + s];
        if (v > sv[tid] || (v == sv[tid] && id < si[tid])) { sv[tid] = v; si[tid] = id; }
      }
      __syncthreads();
    }
  }
  if (tid < 64) {   // wave 0: lane r < world talks to rank r
    const int world = d->world, r = tid;
    if (phase != 2 && r < world) {   // put this rank's pair into every rank's buffer (its own included)
      p2p_gran* g = p2p_slot(d, r, epoch, site, d->rank);
      p2p_put(g, epoch, sv[0]);
      p2p_put_bits(g + 1, epoch, (unsigned)si[0]);
    }
    if (phase == 1) return;          // split form: the get half is a later launch
    p2p_gran* mine = p2p_slot(d, d->rank, epoch, site, r < world ? r : 0);
    float v = p2p_get(d, mine, epoch, r < world, USDM_P2P_ERR_TIMEOUT_PICK, failed);
    int id = (int)p2p_get_bits(d, mine + 1, epoch, r < world, USDM_P2P_ERR_TIMEOUT_PICK, failed);
    if (r >= world) { v = -INFINITY; id = 0x7fffffff; }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(id, off, 64);
      if (ov > v || (ov == v && oi < id)) { v = ov; id = oi; }
    }
    if (tid == 0) {
      const int tok = (id == 0x7fffffff ? 0 : id) + st.id_offset;
      const int step = st.step[0];
      st.next_token[0] = tok;
      if (step < st.max_out) st.out_tokens[step] = tok;
      st.step[0] = step + 1;
      if (st.advance_pos) st.pos[0] = st.pos[0] + 1;
      if (st.done && st.eos) {
        const int n = st.eos[0], mn = st.eos[1];
        bool hit = false;
        for (int i = 0; i < n && i < 6; ++i) hit |= (st.eos[2 + i] == tok);
        if (hit && step + 1 >= mn) st.done[0] = 1;
      }
      s_tok = tok;
      // the step is complete on this rank: next epoch (kernels of the next step are stream-ordered behind this one)
      __hip_atomic_store(p2p_epoch_word(d), epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (phase == 1) return;
  if (E) {
    __syncthreads();
    const u32x4* src = (const u32x4*)(E + (int64_t)s_tok * Hd);
    u32x4* dst = (u32x4*)h_out;
    for (int i = tid; i < Hd / 8; i += 256) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int usdm_allreduce_p2p_reduce(const usdm_p2p_dev* dev, int32_t site, int32_t n_elems, void* h, const int32_t* skip,
                                         usdm_stream_t stream) {
  USDM_CHECK_ARG(dev && h && site >= 0 && n_elems > 0, "usdm_allreduce_p2p_reduce: bad args");
  hipLaunchKernelGGL(p2p_reduce_kernel, dim3(cdiv(n_elems, 256)), dim3(256), 0, (hipStream_t)stream, dev, site, n_elems, (bf16_t*)h, skip);
  USDM_LAUNCH_CHECK();
  return 0;
}

extern "C" int usdm_argmax_p2p(const float* part_val, const int32_t* part_idx, int32_t nparts, const usdm_decode_state* st,
                               const usdm_p2p_dev* dev, int32_t site, int32_t phase, const void* embed_table, int32_t Hd,
                               void* h_out, usdm_stream_t stream) {
  USDM_CHECK_ARG(phase >= 0 && phase <= 2, "usdm_argmax_p2p: phase 0 (put + get), 1 (put), 2 (get)");
  USDM_CHECK_ARG(part_val && part_idx && nparts > 0 && st && st->next_token && st->out_tokens && st->step && st->pos && dev && site >= 0,
                 "usdm_argmax_p2p: bad args");
  USDM_CHECK_ARG(st->batch <= 1, "usdm_argmax_p2p: single sequence only");
  USDM_CHECK_ARG(!embed_table || (h_out && Hd > 0 && Hd % 8 == 0), "usdm_argmax_p2p: embedding output missing");
  hipLaunchKernelGGL(argmax_p2p_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part_val, part_idx, nparts, *st, dev, site, phase,
                     (const bf16_t*)embed_table, Hd, (bf16_t*)h_out);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_p2p_dev(void) { return (int)sizeof(usdm_p2p_dev); }
