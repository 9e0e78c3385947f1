"""CPU restatement of the sampling head (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

Follows HF transformers' logits warpers as the reference's `model.generate(do_sample=True, top_p, top_k, temperature)`
applies them (src/inference.py:63-83; knobs exposed by streamlit_demo.py:201-211), in HF's order:
TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one multinomial draw.
Pinned by tests/test_oracle_cpu.py against the warper classes of the installed transformers (third-party arithmetic;
the reference pins transformers==4.40.2, setup.py:54 — the three warpers are unchanged since).  Two stated deviations,
shared with the HIP kernel: ties at a filter boundary are kept or dropped as a block (HF's unstable sort picks among
them arbitrarily), and the draw uses Philox4x32-10(seed, counter=step) instead of torch's generator.
"""
import numpy as np

M32 = 0xFFFFFFFF


def philox_uniform(seed, ctr):
    """53-bit uniform in [0, 1) from Philox4x32-10 with key = seed, counter = (ctr, 0, 0, 0)."""
    c = [ctr & M32, 0, 0, 0]
    k = [seed & M32, (seed >> 32) & M32]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k[1]) & M32, p0 & M32]
        k = [(k[0] + 0x9E3779B9) & M32, (k[1] + 0xBB67AE85) & M32]
    return float(((c[0] >> 5) << 26) | (c[1] >> 6)) / 9007199254740992.0


def filtered_probs(logits, temperature=1.0, top_k=0, top_p=1.0):
    """float64 probabilities after temperature / top-k / top-p (0 for removed ids).  logits: f32 [V], banned = -inf."""
    x = (np.asarray(logits, np.float32) * np.float32(1.0 / temperature)).astype(np.float32)
    V = x.shape[0]
    if 0 < top_k < V:
        kth = np.partition(x, V - top_k)[V - top_k]
        x = np.where(x < kth, -np.inf, x).astype(np.float32)          # TopKLogitsWarper: scores < k-th largest
    e = np.exp((x - x.max()).astype(np.float32)).astype(np.float32)   # f32 like the kernel; masses summed in fixed point
    q = np.floor(e.astype(np.float64) * 4294967296.0)
    if top_p < 1.0:
        order = np.argsort(e, kind="stable")
        cs = np.cumsum(q[order])
        R = np.floor((1.0 - np.float64(np.float32(top_p))) * q.sum())
        # block semantics for ties: an id is removed iff the mass of all ids with e <= its e stays <= R
        es = e[order]
        last_of_value = np.r_[es[1:] != es[:-1], True]
        block_cs = cs.copy()
        idx_last = np.where(last_of_value)[0]
        starts = np.r_[0, idx_last[:-1] + 1]
        for s0, s1 in zip(starts, idx_last):
            block_cs[s0:s1 + 1] = cs[s1]
        remove = np.zeros(V, bool)
        remove[order] = block_cs <= R
        q = np.where(remove, 0.0, q)
    return q / q.sum()


def sample(logits, step, temperature=1.0, top_k=0, top_p=1.0, seed=0):
    """(token id, probabilities): the id whose cumulative interval (index order) contains u * mass."""
    p = filtered_probs(logits, temperature, top_k, top_p)
    u = philox_uniform(seed, step)
    cs = np.cumsum(p)
    tok = int(np.searchsorted(cs, u * cs[-1], side="right"))
    return min(tok, len(p) - 1), p
