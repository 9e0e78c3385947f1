// Sampling head of the LLM decode step: temperature -> top-k -> top-p -> one multinomial draw, on the device, so that a
// sampled decode step is as replayable (hipGraph) as the greedy one.  Semantics follow HF transformers'
// TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper + torch.multinomial (the reference's
// model.generate(do_sample=True, top_p, top_k, temperature) at src/inference.py:63-83; the demo exposes the knobs,
// streamlit_demo.py:201-211).  What cannot be mirrored is torch's random stream: the draw uses Philox4x32-10 keyed by
// `seed` with the device-side step counter as counter, so a (seed, step) pair always gives the same token.
//
// One workgroup of 1024 threads, a handful of passes over the V logits (L2-resident, 168 KB at V = 42003):
//   top-k  : radix select (4 x 8 bits) of the k-th largest key with integer histograms;
//   top-p  : radix select over the ascending exp(x - max) keys with FIXED-POINT mass histograms (u64 LDS atomics are
//            associative, so the kept set does not depend on thread scheduling); ties are kept or dropped as a block;
//   draw   : u * kept_mass located by an exclusive scan over contiguous index ranges (index order, deterministic).
#include "common.h"
#include "../../include/usdm_hip.h"

namespace {
constexpr int NT = 1024;

__device__ __forceinline__ unsigned fkey(float x) {   // order-preserving float -> uint
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox_round(unsigned (&c)[4], const unsigned (&k)[2]) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k[0], n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k[1], n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ double philox_uniform(unsigned long long seed, unsigned ctr) {   // u in [0, 1), 53 bits
  unsigned c[4] = {ctr, 0u, 0u, 0u}, k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  const unsigned long long hi = c[0] >> 5, lo = c[1] >> 6;   // 27 + 26 bits
  return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
}

__device__ float block_max(float v, float* sred) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = sred[0];
  for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, sred[w]);
  __syncthreads();
  return m;
}

__global__ __launch_bounds__(NT) void sample_final_kernel(usdm_sample_args a, usdm_decode_state st, const bf16_t* E, int Hd,
                                                           bf16_t* h_out) {
  __shared__ unsigned long long hist[256];
  __shared__ unsigned long long sscan[NT];
  __shared__ float sred[NT / 64];
  __shared__ unsigned s_prefix;
  __shared__ unsigned long long s_rem;
  __shared__ int s_tok;
  const int tid = threadIdx.x, V = a.V;
  // batched decode (st.batch > 1): workgroup b = sequence b, with its OWN knobs (dev_params[b]), logits row, state words and
  // Philox counter (its own step): a sequence sampled inside a continuous batch gets the tokens it would get alone
  const int b = blockIdx.x;
  a.logits += (int64_t)b * a.logits_bs;
  if (a.probs_out) a.probs_out += (int64_t)b * a.logits_bs;
  st.next_token += b; st.step += b; st.pos += b; st.out_tokens += (int64_t)b * st.max_out;
  if (st.done) st.done += b;
  h_out += (int64_t)b * Hd;
  if (st.done && *st.done) return;
  if (a.dev_params) {   // per-request knobs live in device memory: one captured graph for every request
    a.dev_params += b;
    a.temperature = a.dev_params->temperature; a.top_k = a.dev_params->top_k; a.top_p = a.dev_params->top_p; a.seed = a.dev_params->seed;
    if (!(a.temperature > 0.f)) a.temperature = 1.0f;
    if (!(a.top_p > 0.f) || a.top_p > 1.0f) a.top_p = 1.0f;
    if (a.top_k < 0) a.top_k = 0;
  }
  const float invT = 1.0f / a.temperature;   // HF divides; x / T and x * (1 / T) differ by <= 1 ulp, below the logits' bf16 grain
  auto X = [&](int i) { return a.logits[i] * invT; };

  // ---- top-k: key of the k-th largest scaled logit (all keys >= it are kept, ties included: `scores < kth` is removed)
  unsigned kth = 0;   // keep everything
  if (a.top_k > 0 && a.top_k < V) {
    unsigned prefix = 0, mask = 0;
    unsigned long long rem = (unsigned long long)a.top_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += NT) {
        const unsigned k = fkey(X(i));
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1ull);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long r = rem;
        int b = 255;
        for (; b > 0; --b) {
          if (hist[b] >= r) break;
          r -= hist[b];
        }
        s_prefix = prefix | ((unsigned)b << shift);
        s_rem = r;
      }
      __syncthreads();
      prefix = s_prefix; rem = s_rem; mask |= 255u << shift;
      __syncthreads();
    }
    kth = prefix;
  }
  // ---- softmax numerators over the top-k survivors
  float m = -INFINITY;
  for (int i = tid; i < V; i += NT) {
    const float x = X(i);
    if (fkey(x) >= kth) m = fmaxf(m, x);
  }
  m = block_max(m, sred);
  auto Ei = [&](int i) -> float {   // exp(x - max) of a top-k survivor, 0 otherwise (banned = -inf -> 0)
    const float x = X(i);
    return (fkey(x) >= kth) ? __expf(x - m) : 0.f;
  };
  auto Q = [&](float e) -> unsigned long long { return (unsigned long long)((double)e * 4294967296.0); };   // 2^32 fixed point
  // total mass (fixed point, integer sum: order-independent)
  if (tid == 0) s_rem = 0;
  __syncthreads();
  {
    unsigned long long z = 0;
    for (int i = tid; i < V; i += NT) z += Q(Ei(i));
    atomicAdd(&s_rem, z);
  }
  __syncthreads();
  const unsigned long long Ztot = s_rem;
  __syncthreads();
  // ---- top-p: drop the ascending tail whose cumulative mass stays <= (1 - top_p) * Z
  unsigned pkey = 0;   // keep e-keys >= pkey (keys of positive floats are ordered like the floats)
  if (a.top_p < 1.0f) {
    unsigned long long R = (unsigned long long)((1.0 - (double)a.top_p) * (double)Ztot);
    unsigned prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += NT) {
        const float e = Ei(i);
        if (e > 0.f) {
          const unsigned k = __float_as_uint(e);
          if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], Q(e));
        }
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long r = R;
        int b = 0;
        for (; b < 255; ++b) {
          if (hist[b] > r) break;      // including bin b would exceed the removable mass: the threshold is inside it
          r -= hist[b];
        }
        s_prefix = prefix | ((unsigned)b << shift);
        s_rem = r;
      }
      __syncthreads();
      prefix = s_prefix; R = s_rem; mask |= 255u << shift;
      __syncthreads();
    }
    pkey = prefix;
  }
  // ---- draw: contiguous index ranges, exclusive scan of the kept fixed-point masses
  const int per = (V + NT - 1) / NT;
  const int i0 = tid * per, i1 = min(V, i0 + per);
  unsigned long long loc = 0;
  for (int i = i0; i < i1; ++i) {
    const float e = Ei(i);
    if (e > 0.f && __float_as_uint(e) >= pkey) loc += Q(e);
  }
  sscan[tid] = loc;
  __syncthreads();
  for (int off = 1; off < NT; off <<= 1) {   // Hillis-Steele inclusive scan
    const unsigned long long v = tid >= off ? sscan[tid - off] : 0ull;
    __syncthreads();
    sscan[tid] += v;
    __syncthreads();
  }
  const unsigned long long Zk = sscan[NT - 1];
  const int step = *st.step;
  // top_k == 1 is the reference's "greedy" (do_sample=True, top_k=1): among exact ties take the lowest id, as usdm_argmax_final does
  const double u = a.top_k == 1 ? 0.0 : philox_uniform(a.seed, (unsigned)step);
  unsigned long long target = (unsigned long long)(u * (double)Zk);
  if (Zk > 0 && target >= Zk) target = Zk - 1;
  const unsigned long long excl = sscan[tid] - loc;
  if (tid == 0) s_tok = -1;
  __syncthreads();
  if (loc > 0 && excl <= target && target < excl + loc) {
    unsigned long long c = excl;
    int pick = i0;
    for (int i = i0; i < i1; ++i) {
      const float e = Ei(i);
      if (e > 0.f && __float_as_uint(e) >= pkey) {
        const unsigned long long q = Q(e);
        if (target < c + q) { pick = i; break; }
        c += q;
      }
    }
    s_tok = pick;
  }
  __syncthreads();
  if (s_tok < 0) {   // no id has positive mass (all banned / NaN logits): arg-max of the finite logits, else id 0
    __shared__ float fbv[NT / 64];
    __shared__ int fbi[NT / 64];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += NT) {
      const float x = a.logits[i];
      if (x > bv) { bv = x; bi = i; }     // NaN and -inf never pass
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { fbv[tid >> 6] = bv; fbi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < NT / 64; ++w)
        if (fbv[w] > bv || (fbv[w] == bv && fbi[w] < bi)) { bv = fbv[w]; bi = fbi[w]; }
      s_tok = bi == 0x7fffffff ? 0 : bi;
    }
    __syncthreads();
  }
  if (a.probs_out) {
    const double inv = Zk > 0 ? 1.0 / (double)Zk : 0.0;
    for (int i = tid; i < V; i += NT) {
      const float e = Ei(i);
      a.probs_out[i] = (e > 0.f && __float_as_uint(e) >= pkey) ? (float)((double)Q(e) * inv) : 0.f;
    }
  }
  if (tid == 0) {
    const int tok = s_tok + st.id_offset;
    *st.next_token = tok;
    if (step < st.max_out) st.out_tokens[step] = tok;
    *st.step = step + 1;
    if (st.advance_pos) *st.pos = *st.pos + 1;
    if (st.done && st.eos) {   // device-side EOS: eos = {count, min_new, ids...}
      const int n = st.eos[0], mn = st.eos[1];
      bool hit = false;
      for (int i = 0; i < n && i < 6; ++i) hit |= (st.eos[2 + i] == tok);
      if (hit && step + 1 >= mn) *st.done = 1;
    }
  }
  if (E) {
    const u32x4* src = (const u32x4*)(E + (int64_t)(s_tok + st.id_offset) * Hd);
    u32x4* dst = (u32x4*)h_out;
    for (int i = tid; i < Hd / 8; i += NT) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int usdm_sample_final(const usdm_sample_args* pa, const usdm_decode_state* st, const void* embed_table, int32_t Hd,
                                 void* h_out, usdm_stream_t stream) {
  USDM_CHECK_ARG(pa && pa->logits && pa->V > 0 && pa->V <= (1 << 20), "usdm_sample_final: logits / V");
  USDM_CHECK_ARG(pa->dev_params || (pa->temperature > 0.f && pa->top_p > 0.f && pa->top_p <= 1.0f && pa->top_k >= 0),
                 "usdm_sample_final: temperature > 0, 0 < top_p <= 1, top_k >= 0 (0 = off)");
  USDM_CHECK_ARG(st && st->next_token && st->out_tokens && st->step && st->pos, "usdm_sample_final: decode state");
  USDM_CHECK_ARG(!embed_table || (h_out && Hd > 0 && Hd % 8 == 0), "usdm_sample_final: embedding output missing");
  const int nb = st->batch > 1 ? st->batch : 1;
  USDM_CHECK_ARG(nb == 1 || (pa->logits_bs >= pa->V && pa->dev_params), "usdm_sample_final: the batched form needs logits_bs >= V and dev_params[batch]");
  hipLaunchKernelGGL(sample_final_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, *pa, *st, (const bf16_t*)embed_table, Hd,
                     (bf16_t*)h_out);
  USDM_LAUNCH_CHECK();
  return 0;
}
extern "C" int usdm_sizeof_sample_args(void) { return (int)sizeof(usdm_sample_args); }
