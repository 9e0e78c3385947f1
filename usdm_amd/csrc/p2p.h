// Device-side helpers of the one-shot peer-to-peer all-reduce (include/usdm_hip.h, usdm_allreduce_p2p_*).
// A granule is one naturally aligned 8-byte word {tag = epoch (high 32), f32 bits (low 32)} written by ONE system-scope store
// (sc0 sc1: write-through, visible across xGMI) and read by system-scope loads of the reader's OWN uncached buffer.
#pragma once
#include "common.h"
#include "../../include/usdm_hip.h"

typedef unsigned long long p2p_gran;

__device__ __forceinline__ unsigned* p2p_epoch_word(const usdm_p2p_dev* d) { return (unsigned*)d->base[d->rank]; }
__device__ __forceinline__ unsigned* p2p_err_word(const usdm_p2p_dev* d) { return (unsigned*)d->base[d->rank] + 1; }

// granule n of slot[parity][site][src] inside the buffer of rank `owner`
__device__ __forceinline__ p2p_gran* p2p_slot(const usdm_p2p_dev* d, int owner, unsigned epoch, int site, int src) {
  const int64_t idx = ((int64_t)((epoch & 1) * d->n_sites + site) * USDM_P2P_MAX_RANKS + src) * d->max_elems;
  return (p2p_gran*)(d->base[owner] + USDM_P2P_HEADER_BYTES) + idx;
}

__device__ __forceinline__ unsigned p2p_load_epoch(const usdm_p2p_dev* d) {
  return __hip_atomic_load(p2p_epoch_word(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned p2p_load_err(const usdm_p2p_dev* d) {
  return __hip_atomic_load(p2p_err_word(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void p2p_put_bits(p2p_gran* g, unsigned epoch, unsigned bits) {
  __hip_atomic_store(g, ((unsigned long long)epoch << 32) | (unsigned long long)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void p2p_put(p2p_gran* g, unsigned epoch, float v) { p2p_put_bits(g, epoch, __float_as_uint(v)); }

// Bounded wait for one granule per ACTIVE lane (inactive lanes pass want = false); returns the value (0 on give-up).
// Wave-uniform loop: the wave leaves when every lane has its tag or the deadline passed / another kernel already failed.
__device__ __forceinline__ unsigned p2p_get_bits(const usdm_p2p_dev* d, p2p_gran* g, unsigned epoch, bool want, unsigned err_code,
                                                 bool already_failed) {
  unsigned long long x = 0;
  bool ok = !want;
  if (already_failed) return 0u;
  const unsigned long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    if (!ok) {
      x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ok = (unsigned)(x >> 32) == epoch;
    }
    if (__all(ok)) break;
    if ((spins & 63) == 63) {
      if (wall_clock64() - t0 > d->timeout_ticks) {   // give up: flag the failure, later kernels do not wait again
        if (!ok) __hip_atomic_fetch_or(p2p_err_word(d), err_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // ... and tell every peer (one lane of the wave, one system-scope OR per peer): a rank that timed out carries on with
        // zeros for what it missed, so its token stream may diverge; with the word set everywhere ALL ranks raise at their
        // next host sync instead of only this one
        const unsigned long long fm = __ballot(!ok);
        if (fm != 0ull && (int)(threadIdx.x & 63) == __ffsll((long long)fm) - 1) {
          for (int r = 0; r < d->world; ++r)
            if (r != d->rank)
              __hip_atomic_fetch_or((unsigned*)d->base[r] + 1, err_code | USDM_P2P_ERR_PEER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return ok && want ? (unsigned)x : 0u;
}
__device__ __forceinline__ float p2p_get(const usdm_p2p_dev* d, p2p_gran* g, unsigned epoch, bool want, unsigned err_code,
                                         bool already_failed) {
  return __uint_as_float(p2p_get_bits(d, g, epoch, want, err_code, already_failed));
}
