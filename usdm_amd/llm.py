"""Mistral-7B speech-text LLM on MI355X: drop-in for the object the reference obtains from
`AutoModelForCausalLM.from_pretrained('naver-ai/USDM-DailyTalk', torch_dtype=bf16)` (src/inference.py:116-124)
as far as the hot path uses it: `.generate(input_ids=[1,L], max_length, do_sample, bad_words_ids, top_p, top_k,
temperature, eos_token_id) -> LongTensor[1,L']` (src/inference.py:63-83).

Host side: Python plans of C-ABI launches (libusdm_hip.so).
  prefill : usdm_embed_rows, usdm_norm(rms), usdm_gemm (MFMA bf16; SwiGLU / residual epilogues), usdm_rope_cache,
            usdm_attention(mode 1, causal GQA)                                   — one eager plan per prompt length
  decode  : usdm_gemv x4 per layer (RMSNorm, residual add, SwiGLU fused), usdm_attn_decode, lm_head GEMV with
            ban-mask + arg-max, all state on the device -> ONE hipGraph replayed per token
  TP > 1  : Megatron-style shards (q/k/v heads, MLP columns, vocab rows).  Decode: the all-reduce of the o_proj/down_proj
            partial sums is fused into those GEMVs' epilogues as a one-shot peer-to-peer exchange over xGMI
            (usdm_amd.p2p.P2PComm; the decode step stays ONE hipGraph with no collective launch), the vocab-parallel token
            pick likewise (usdm_argmax_p2p).  Prefill (4 MB messages) and the validation path use RCCL through
            torch.distributed ('nccl'): f32 partial sums all-reduced, arg-max partials all-gathered.
Weights: HF state-dict key names (model.layers.N.self_attn.q_proj.weight, ...), bf16.
"""
import math
import os

import torch

from . import ops
from ._lib import ACT_SWIGLU
from .graph import GraphedPlan, GraphedSegments
from .plancache import LRU

MISTRAL_7B_USDM = dict(vocab_size=42003, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5,
                       rope_theta=10000.0, max_position_embeddings=32768)


def _pack_gate_up(gate, up):
    """[I,K] gate and up -> [2I,K] in blocks of 32 rows = 16 gate + 16 up (SwiGLU epilogue layout)."""
    I, K = gate.shape
    return torch.stack([gate.reshape(I // 16, 16, K), up.reshape(I // 16, 16, K)], 1).reshape(2 * I, K).contiguous()


def vocab_shard(V, rank, tp):
    """Vocab-parallel lm_head layout: (Vloc, v0, v1, nparts).  Every rank owns Vloc = ceil(V/tp) row SLOTS (the last rank
    fewer real rows: 42 003 / 8 -> 7 x 5251 + 5246) and nparts = usdm_gemv_nblocks(Vloc) arg-max partial slots, so the
    partial buffers that are exchanged have the SAME size on every rank (a collective with per-rank counts that differ is
    undefined behaviour in RCCL); unused slots keep their (-inf, 0x7fffffff) fill and can never win."""
    Vloc = (V + tp - 1) // tp
    v0 = min(V, rank * Vloc)
    v1 = min(V, v0 + Vloc)
    return Vloc, v0, v1, ops.gemv_nblocks(Vloc)


NO_CANDIDATE_IDX = 0x7fffffff


def shard_weights(sd_get, cfg, rank, tp, device, dtype=torch.bfloat16):
    """This rank's packed weights for tensor parallelism of degree `tp` (Megatron-style): q/k/v heads and MLP
    columns split by rank, o_proj/down_proj split along K (their outputs are partial sums), vocab rows split."""
    d = cfg["head_dim"]
    Hq, Hkv, I = cfg["num_attention_heads"] // tp, cfg["num_key_value_heads"] // tp, cfg["intermediate_size"] // tp
    V = cfg["vocab_size"]
    _, v0, v1, _ = vocab_shard(V, rank, tp)
    g = lambda n: sd_get(n).to(device, dtype)
    f = lambda n: sd_get(n).to(device, torch.float32).contiguous()
    W = {"embed": g("model.embed_tokens.weight").contiguous(), "norm": f("model.norm.weight"),
         "lm_head": g("lm_head.weight")[v0:v1].contiguous(), "layers": [], "v0": v0, "v1": v1}
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        q = g(p + "self_attn.q_proj.weight")[rank * Hq * d:(rank + 1) * Hq * d]
        k = g(p + "self_attn.k_proj.weight")[rank * Hkv * d:(rank + 1) * Hkv * d]
        v = g(p + "self_attn.v_proj.weight")[rank * Hkv * d:(rank + 1) * Hkv * d]
        o = g(p + "self_attn.o_proj.weight")[:, rank * Hq * d:(rank + 1) * Hq * d]
        ga = g(p + "mlp.gate_proj.weight")[rank * I:(rank + 1) * I]
        up = g(p + "mlp.up_proj.weight")[rank * I:(rank + 1) * I]
        dn = g(p + "mlp.down_proj.weight")[:, rank * I:(rank + 1) * I]
        W["layers"].append(dict(qkv=torch.cat([q, k, v], 0).contiguous(), o=o.contiguous(), gu=_pack_gate_up(ga, up),
                                down=dn.contiguous(), ln1=f(p + "input_layernorm.weight"),
                                ln2=f(p + "post_attention_layernorm.weight")))
    return W


class USDMForCausalLM:
    def __init__(self, cfg, device, ctx_max=2048, tp_rank=0, tp_size=1, group=None, decode_splits=None, tp_segments=None, p2p=None,
                 p2p_fused=None):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("USDMForCausalLM (usdm_amd) runs on the MI355X only; there is no CPU fallback")
        c = self.cfg
        if c["head_dim"] != 128:
            raise NotImplementedError("decode/prefill attention kernels are built for head_dim 128 (Mistral-7B)")
        self.tp_rank, self.tp_size, self.group = tp_rank, tp_size, group
        # tp_segments: run the tensor-parallel code path (f32 partial sums + all-reduce + residual-add kernels) even at
        # tp_size 1 — used to exercise that path on a single GPU
        self.tp_path = (tp_size > 1) if tp_segments is None else bool(tp_segments)
        # p2p: a committed usdm_amd.p2p.P2PComm -> the decode step exchanges its partial sums peer to peer inside the GEMV
        # epilogues (p2p_fused, default) or through put + usdm_allreduce_p2p_reduce launches (split form, USDM_P2P_FUSED=0)
        self.p2p = p2p
        _os = os
        self.p2p_fused = (_os.environ.get("USDM_P2P_FUSED", "1") == "1") if p2p_fused is None else bool(p2p_fused)
        if p2p is not None:
            if not self.tp_path:
                raise ValueError("p2p needs the tensor-parallel path (tp_size > 1 or tp_segments=True)")
            if (p2p.rank, p2p.world) != (tp_rank, tp_size) and os.environ.get("USDM_P2P_PROXY") != "1":   # (tools/tp8_proxy.py)
                raise ValueError(f"P2PComm is rank {p2p.rank}/{p2p.world}, the model shard is rank {tp_rank}/{tp_size}")
            if p2p.n_sites < 2 * c["num_hidden_layers"] + 1 or p2p.max_elems < c["hidden_size"]:
                raise ValueError("P2PComm too small: needs 2*layers+1 sites of hidden_size elements")
        self.Hq, self.Hkv = c["num_attention_heads"] // tp_size, c["num_key_value_heads"] // tp_size
        if self.Hq * tp_size != c["num_attention_heads"] or self.Hkv * tp_size != c["num_key_value_heads"] or self.Hkv < 1:
            raise ValueError("tp_size must divide both head counts")
        self.I = c["intermediate_size"] // tp_size
        if self.I * tp_size != c["intermediate_size"] or self.I % 16:
            raise ValueError("intermediate_size / tp_size must be a multiple of 16")
        V = c["vocab_size"]
        self.Vloc, self.v0, self.v1, self.nparts = vocab_shard(V, tp_rank, tp_size)
        self.ctx_max = (ctx_max + 63) // 64 * 64
        # Mistral-7B-v0.1 attends to the last `sliding_window` (4096) positions only (reference: src/model.py:337-371 keeps
        # W - 1 past keys + the new one; HF sliding-window mask: query p sees keys p-W+1 .. p).  The caches here keep every row up
        # to ctx_max and the attention kernels take the window as a key-range bound, so nothing changes while ctx_max <= W.
        W = c.get("sliding_window", 4096)
        self.window = int(W) if (W and self.ctx_max > int(W)) else 0
        # Decode attention is split over the context (NS workgroups per kv head).  The NS partials per head are merged in the
        # o_proj GEMV's x-staging prologue (usdm_gemv mrg_*; no combine launch) -> few, fat splits: every o_proj workgroup reads
        # all of them (NS x 16 KB from L2).  USDM_ATTN_MERGE_IN_OPROJ=0 restores the separate combine kernel (NS = 32).
        # Measured (profiles/r02_decode_ablation.txt) and OFF by default: on the single-GPU 7B shapes the merge costs the o_proj
        # launch more than the combine launch it removes (2.97 -> 3.10 ms/token at NS = 8: 128 KB of partials per workgroup and
        # two dependent L2 round trips no longer hide under the weight ring; NS = 32: 3.28), and the fewer, fatter splits it
        # wants make the latency-bound split kernel slower (rank-0-of-8 proxy: 1.10 -> 1.23 ms/token).  USDM_ATTN_MERGE_IN_OPROJ=1
        # enables it.
        self.merge_in_oproj = os.environ.get("USDM_ATTN_MERGE_IN_OPROJ", "0") == "1"
        # Hand-off form (round 3, usdm_gemv cmb_gran; the DEFAULT since round 4): no combine launch either, but the merge is done ONCE
        # per head by the o_proj launch's first 32 workgroups and handed to the others as granules, under the launch's first weight
        # ring.  Single-GPU 7B shape only (4096 outputs = one 16-wave workgroup per CU).  -0.8 % per token, token- and logit-identical
        # (profiles/r03_decode_ablation.txt 5, tests/test_fullsize_gpu.py).  The o_proj launch then carries the merge (7.7 -> 9.3 us);
        # bench.py's roofline object reports that launch both ways.  USDM_ATTN_CMB=0 restores the separate combine kernel.
        self.cmb = (os.environ.get("USDM_ATTN_CMB", "1") == "1" and tp_size == 1 and not self.tp_path and not self.merge_in_oproj
                    and c["hidden_size"] == 4096 and self.Hq * c["head_dim"] == 4096)
        dflt = max(8, -(-self.ctx_max // 512)) if self.merge_in_oproj else 32
        self.NS = int(os.environ.get("USDM_DECODE_SPLITS", str(dflt))) if decode_splits is None else decode_splits
        if self.NS == 1:
            self.merge_in_oproj = False          # one workgroup per kv head: nothing to merge
        # Chained decode GEMVs (usdm_gemv_chain): consecutive projections of a layer in ONE persistent launch whose weight stream
        # runs across the phase boundaries.  0 = off (one launch per projection; the DEFAULT: measured slower, see below),
        # 3 = o_proj -> gate/up -> down_proj, 4 = ... -> the next layer's qkv as well.  Single-GPU path only (the kernel needs the
        # whole GPU resident).  Measured (profiles/r02_decode_ablation.txt section 3): 85 vs 69 us per layer for 3 phases - a flat
        # counter grid barrier at 2 workgroups per CU costs ~13 us, more than the launch boundary + ramp it replaces, and one
        # workgroup shape for every phase streams 10-25 % slower than the per-shape tuned kernels.
        # "e3" / "e4": the same chains on the loader / consumer engine (usdm_gemv_engine: LDS-DMA weight ring, granule hand-offs).
        mode = os.environ.get("USDM_GEMV_CHAIN", "0")
        self.chain_engine = mode.startswith("e")
        self.chain = int(mode.lstrip("e") or 0)
        if self.tp_path or c["hidden_size"] != 4096:
            self.chain = 0
        self.chain_sync = None
        self.W = None
        # bounded caches (plancache.LRU): prefill plans are keyed by exact prompt length (a plan is argument structs + ~60 KB of
        # workspace per token; no hipGraph), decode plans / graphs by {greedy, sampling} only
        self._prefill_plans = LRU(24)
        self._decode = None
        self._decodes = {}
        self._batches = {}
        self._ban_cache = LRU(8)
        self.stats = {}
        self.logits_hook = None   # generate(_logits_hook=f): f() runs between the lm_head launch and the sampling pick of every step
        self.keep_logits = False  # debug/tests: keep the fp32 (bf16-valued) logits of the last step
        self.last_logits = None

    # ------------------------------------------------------------------ weights
    def _shard(self, sd_get):
        """Build this rank's packed weights from a getter name -> tensor (any device)."""
        W = shard_weights(sd_get, self.cfg, self.tp_rank, self.tp_size, self.device)
        assert (W["v0"], W["v1"]) == (self.v0, self.v1)
        return W

    @classmethod
    def from_state_dict(cls, sd, cfg, device, **kw):
        m = cls(cfg, device, **kw)
        m.W = m._shard(lambda n: sd[n])
        m._alloc()
        return m

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, device="cuda", cache_dir=None, torch_dtype=torch.bfloat16, **kw):
        """Load an HF-format Mistral checkpoint directory (config.json + [sharded] safetensors or .bin), as the reference does
        with AutoModelForCausalLM.from_pretrained('naver-ai/USDM-DailyTalk', cache_dir=..., torch_dtype=bf16)
        (src/inference.py:116-123).  Hub names resolve inside cache_dir (no network).  Tensors stream from the memory-mapped
        shards to the GPU one at a time; a tensor-parallel rank slices its shard on the way (tp_rank / tp_size in **kw)."""
        from .checkpoints import TensorSource, read_mistral_config, resolve_local
        if torch_dtype != torch.bfloat16:
            raise NotImplementedError("the decode kernels stream bf16 weights (the reference loads torch_dtype=torch.bfloat16)")
        path = pretrained_model_name_or_path
        if not os.path.isdir(path):
            if cache_dir is None:
                raise FileNotFoundError(f"{path}: not a directory, and no cache_dir to resolve the hub name in (no network)")
            path = resolve_local(cache_dir, path, must_contain=("config.json",))
        for drop in ("attn_implementation", "device_map", "low_cpu_mem_usage"):     # the reference's HF-only knobs
            kw.pop(drop, None)
        cfg = read_mistral_config(path)
        m = cls(cfg, device, **kw)
        src = TensorSource(path)
        m.W = m._shard(src)
        m._alloc()
        m.name_or_path = path
        return m

    def to(self, device):
        """The reference calls .to(device) on the loaded model (inference.py:123); weights already live on the GPU."""
        if torch.device(device).type != "cuda":
            raise RuntimeError("USDMForCausalLM (usdm_amd) runs on the MI355X only; there is no CPU fallback")
        return self

    def eval(self):
        return self

    @classmethod
    def random_init(cls, cfg, device, seed=0, **kw):
        """Random weights generated shard-by-shard ON the device (synthetic benchmark weights;
        values differ from oracle.mistral_oracle.random_state_dict, which is CPU-generated)."""
        m = cls(cfg, device, **kw)
        c, d = m.cfg, m.cfg["head_dim"]
        H, I, V = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
        gen = torch.Generator(device=m.device).manual_seed(seed)
        shapes = {"model.embed_tokens.weight": ((V, H), 1.0), "lm_head.weight": ((V, H), H ** -0.5)}
        for l in range(c["num_hidden_layers"]):
            p = f"model.layers.{l}."
            shapes.update({p + "self_attn.q_proj.weight": ((c["num_attention_heads"] * d, H), H ** -0.5),
                           p + "self_attn.k_proj.weight": ((c["num_key_value_heads"] * d, H), H ** -0.5),
                           p + "self_attn.v_proj.weight": ((c["num_key_value_heads"] * d, H), H ** -0.5),
                           p + "self_attn.o_proj.weight": ((H, c["num_attention_heads"] * d), H ** -0.5),
                           p + "mlp.gate_proj.weight": ((I, H), H ** -0.5), p + "mlp.up_proj.weight": ((I, H), H ** -0.5),
                           p + "mlp.down_proj.weight": ((H, I), I ** -0.5)})

        def get(n):
            if n.endswith("norm.weight") or n.endswith("layernorm.weight"):
                return torch.ones(H, device=m.device)
            shape, sc = shapes[n]
            return (torch.randn(shape, device=m.device, dtype=torch.float32, generator=gen) * sc).to(torch.bfloat16)
        m.W = m._shard(get)
        m._alloc()
        return m

    def weight_bytes_per_token(self):
        """bf16 bytes a decode step must stream on this rank (layers + lm_head shard)."""
        n = sum(l[k].numel() for l in self.W["layers"] for k in ("qkv", "o", "gu", "down")) + self.W["lm_head"].numel()
        return 2 * n

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        c, dev = self.cfg, self.device
        L, d = c["num_hidden_layers"], c["head_dim"]
        bf = torch.bfloat16
        self.kcache = torch.zeros(L, self.Hkv, self.ctx_max, d, dtype=bf, device=dev)
        self.vcache = torch.zeros(L, self.Hkv, self.ctx_max, d, dtype=bf, device=dev)
        # V^T of the prompt tokens (what the prefill attention consumes), kept across generate() calls so that a prompt which
        # extends the cached sequence only prefills its new tokens (the reference's three rounds: src/inference.py:61-83)
        # reuse_prefix = "exact" (default; USDM_PREFIX_REUSE=exact): only cache rows that a PREFILL launch wrote are reused.  A row of
        # the prefill path does not depend on how many tokens were prefilled with it (per-row GEMM and flash-attention arithmetic is
        # independent of the tile a row sits in), so the result is bit-identical with recomputing the whole prompt as the reference
        # does - checked on logits in tests/test_llm_gpu.py::test_exact_prefix_reuse_is_bit_identical.
        # reuse_prefix = True (USDM_PREFIX_REUSE=1): rows appended by decode steps are reused too; they come from the GEMV path and can
        # differ from a from-scratch prefill by a bf16 ulp (close to, not identical with, the reference).  False / 0: off.
        mode = os.environ.get("USDM_PREFIX_REUSE", "exact")
        self.reuse_prefix = {"0": False, "off": False, "1": True, "all": True}.get(mode, "exact")
        self.vtc = torch.zeros(L, self.Hkv, d, self.ctx_max, dtype=bf, device=dev)
        self._kv_ids, self._vt_upto = None, 0
        # rope tables exactly as HF MistralRotaryEmbedding computes them (fp32 on the host, cast to bf16)
        inv_freq = 1.0 / (c["rope_theta"] ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
        fr = torch.arange(self.ctx_max).float()[:, None] * inv_freq[None, :]
        self.cos = fr.cos().to(bf).to(dev).contiguous()
        self.sin = fr.sin().to(bf).to(dev).contiguous()
        i32 = lambda n, v=0: torch.full((n,), v, dtype=torch.int32, device=dev)
        self.max_out = self.ctx_max
        self.st_next, self.st_step, self.st_pos = i32(1), i32(1), i32(1)
        self.st_out = i32(self.max_out)
        # device-side end of sequence: st_eos = {count, min_new, ids...}; st_done is set by the token-picking kernel and read by
        # every decode kernel as its skip word, so steps launched past an EOS inside a host chunk return immediately
        self.st_done, self.st_eos = i32(1), i32(8)
        self.sample_params = ops.sample_params_tensor(dev)   # usdm_sample_params of the current request
        self.ban_all_off = torch.zeros(self.v1 - self.v0, dtype=torch.uint8, device=dev)
        self.ban = torch.zeros(self.v1 - self.v0, dtype=torch.uint8, device=dev)  # live mask read by the graphs
        self.h_dec = torch.zeros(c["hidden_size"], dtype=bf, device=dev)  # residual stream of the decode step
        # usdm_gemv_chain sync blocks (generation, error, arrival counters): ONE PER CHAIN PATTERN - the counters of a block are
        # monotonic in lockstep with its generation, so launches with different phase counts must not share a block
        self.chain_sync = torch.zeros(2, 8, dtype=torch.int32, device=dev)
        self.chain_gran = torch.zeros(3 * 8192, dtype=torch.int64, device=dev)     # usdm_gemv_engine hand-off granules
        # arg-max partials: nparts slots per rank, the same on every rank (vocab_shard); slots the lm_head launch does not
        # write (last rank's shorter shard) stay "no candidate"
        nv = lambda n: torch.full((n,), float("-inf"), dtype=torch.float32, device=dev)
        ni = lambda n: torch.full((n,), NO_CANDIDATE_IDX, dtype=torch.int32, device=dev)
        self.part_val, self.part_idx = nv(self.nparts * self.tp_size), ni(self.nparts * self.tp_size)
        self.part_val_loc = nv(self.nparts) if self.tp_path else self.part_val
        self.part_idx_loc = ni(self.nparts) if self.tp_path else self.part_idx

    # ------------------------------------------------------------------ collectives (TP only)
    def _host_staged(self):
        """A gloo group (validation runs with several ranks on ONE GPU, where RCCL refuses to form a group): collectives are
        staged through host memory.  Never the production transport."""
        import torch.distributed as dist
        return dist.get_backend(self.group) == "gloo"

    def _all_reduce(self, t):
        import torch.distributed as dist
        if hasattr(self.group, "usdm_all_reduce"):      # in-process logical ranks (usdm_amd.p2p.InProcessGroup)
            return self.group.usdm_all_reduce(self.tp_rank, t)
        if self._host_staged():
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
            return
        dist.all_reduce(t, group=self.group)

    def _gather_partials(self, dsts=None, srcs=None):
        """all-gather of the ranks' arg-max partials (rank-major: dst = concatenation of the ranks' src along dim 0).  Default: the
        single sequence's buffers; a batch slot / the batched step pass their own."""
        import torch.distributed as dist
        dsts = [self.part_val, self.part_idx] if dsts is None else dsts
        srcs = [self.part_val_loc, self.part_idx_loc] if srcs is None else srcs
        if hasattr(self.group, "usdm_all_gather"):
            return self.group.usdm_all_gather(self.tp_rank, dsts, srcs)
        if self._host_staged():
            for dst, src in zip(dsts, srcs):
                c = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(c, src.cpu(), group=self.group)
                dst.copy_(c)
            return
        for dst, src in zip(dsts, srcs):
            dist.all_gather_into_tensor(dst, src, group=self.group)

    def _lm_head_and_pick(self, plan, x, advance_pos, segs, sampling=None, x_delta=None, slot=None, skip=None):
        """lm_head GEMV + token choice.  sampling=None: ban-masked arg-max (the reference's top_k=1 path);
        sampling=True: usdm_sample_final over the ban-masked logits, knobs read from the device block self.sample_params
        (written per request by generate(): plans and graphs do not depend on temperature / top-k / top-p / seed)."""
        c = self.cfg
        sl = slot or self   # where the picked token, the decode state and the next input row live (self = the single sequence)
        single = sl is self
        want_logits = self.keep_logits or bool(sampling)
        if want_logits and self.last_logits is None:
            self.last_logits = torch.zeros(self.v1 - self.v0, dtype=torch.float32, device=self.device)
        # a batch slot samples from ITS logits row with ITS knobs (per-slot sampling inside a continuous batch)
        logits = (self.last_logits if single else sl.logits) if want_logits else None
        ops.gemv(self.W["lm_head"], x, N=self.v1 - self.v0, K=c["hidden_size"], norm_w=self.W["norm"], eps=c["rms_norm_eps"],
                 y32=logits, ban=self.ban, part_val=sl.part_val_loc, part_idx=sl.part_idx_loc, idx_offset=self.v0,
                 x_delta=x_delta, skip=skip, plan=plan)
        st = ops.decode_state(sl.st_next, sl.st_out, sl.st_step, sl.st_pos, advance_pos=advance_pos,
                              done=self.st_done if single else None, eos=self.st_eos if single else None)
        if sampling:
            if self.tp_path:
                raise NotImplementedError("sampling needs the full logit row on one GPU (tensor-parallel decode is greedy only)")
            if sampling == "hook":      # Python logits processors (usdm_amd.serving): a host call between the two kernels
                segs.append(plan)
                segs.append(lambda: self.logits_hook())
                plan = ops.Plan()
            ops.sample_final(logits, st, dev_params=self.sample_params if single else sl.sample_params,
                             embed=self.W["embed"], h_out=sl.h_dec, Hd=c["hidden_size"], plan=plan)
            plan.hold(st)
            segs.append(plan)
            return
        if self.p2p is not None and single:
            # vocab-parallel pick across ranks in ONE launch (pairs exchanged peer to peer; advances the exchange epoch)
            kw = dict(embed=self.W["embed"], h_out=sl.h_dec, Hd=c["hidden_size"])
            site = 2 * c["num_hidden_layers"]
            if self.p2p_fused:
                ops.argmax_p2p(sl.part_val_loc, sl.part_idx_loc, self.nparts, st, self.p2p, site, phase=0, plan=plan, **kw)
            else:       # split form: put | get as two launches with a segment boundary between them
                ops.argmax_p2p(sl.part_val_loc, sl.part_idx_loc, self.nparts, st, self.p2p, site, phase=1, plan=plan, **kw)
                plan.hold(st)
                segs.append(plan)
                plan = ops.Plan()
                ops.argmax_p2p(sl.part_val_loc, sl.part_idx_loc, self.nparts, st, self.p2p, site, phase=2, plan=plan, **kw)
            plan.hold(st)
            segs.append(plan)
            return
        if self.tp_path:
            segs.append(plan)
            segs.append(self._gather_partials if single else
                        (lambda: self._gather_partials([sl.part_val, sl.part_idx], [sl.part_val_loc, sl.part_idx_loc])))
            plan = ops.Plan()
        # the picked token's embedding row is written straight into the decode step's input vector
        ops.argmax_final(sl.part_val, sl.part_idx, self.nparts * self.tp_size, st, embed=self.W["embed"], h_out=sl.h_dec,
                         Hd=c["hidden_size"], plan=plan)
        plan.hold(st)
        segs.append(plan)

    # ------------------------------------------------------------------ plans
    def _build_prefill(self, S, sampling=None, slot=None, past=0):
        """Prefill of S new tokens at positions past .. past+S-1 (past > 0: the KV cache already holds the first `past`
        tokens of the same sequence; only the single-sequence cache keeps the V^T that makes this possible)."""
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        H, d, L = c["hidden_size"], c["head_dim"], c["num_hidden_layers"]
        Hq, Hkv, I, tp = self.Hq, self.Hkv, self.I, (2 if self.tp_path else 1)
        nq = (Hq + 2 * Hkv) * d
        Spad = (S + 63) // 64 * 64
        segs, plan = [], ops.Plan()
        Z = lambda *s, dt=bf: plan.hold(torch.zeros(*s, device=dev, dtype=dt))
        io = dict(ids=Z(S, dt=torch.int64))
        h, xn, qkv, ao, act = Z(S, H), Z(S, H), Z(S, nq), Z(S, Hq * d), Z(S, I)
        vt = Z(Hkv, d, Spad) if slot is not None else None     # batch slots: scratch V^T of this prompt only
        assert past == 0 or slot is None
        part = Z(S, H, dt=torch.float32) if tp > 1 else None
        ops.embed_rows(self.W["embed"], h, Hd=H, ids=io["ids"], n=S, plan=plan)
        for l in range(L):
            w = self.W["layers"][l]
            ops.norm(h, w["ln1"], None, rows=S, C=H, eps=c["rms_norm_eps"], rms=True, round_bf16=True, out16=xn, plan=plan)
            ops.gemm(xn, w["qkv"], M=S, N=nq, Kc=H, out16=qkv, plan=plan)
            if slot is not None:
                ops.rope_cache(qkv, self.cos, self.sin, slot.kcache[l], slot.vcache[l], ld=nq, S=S, pos0=0, Hq=Hq, Hkv=Hkv,
                               ctx_max=self.ctx_max, max_pos=self.ctx_max, vt=vt, vt_ld=Spad, plan=plan)
                ops.attention(qkv, slot.kcache[l], vt, ao, mode=1, dh=d, B=1, Hq=Hq, Hkv=Hkv, Sq=S, Skv=S, Skv_alloc=Spad,
                              q_strides=(0, d, nq), k_strides=(0, self.ctx_max * d, d), v_strides=(0, d * Spad, Spad),
                              o_strides=(0, Hq * d), scale=d ** -0.5, window=self.window, plan=plan)
            else:
                ops.rope_cache(qkv, self.cos, self.sin, self.kcache[l], self.vcache[l], ld=nq, S=S, pos0=past, Hq=Hq, Hkv=Hkv,
                               ctx_max=self.ctx_max, max_pos=self.ctx_max, vt=self.vtc[l][:, :, past:], vt_ld=self.ctx_max, plan=plan)
                ops.attention(qkv, self.kcache[l], self.vtc[l], ao, mode=1, dh=d, B=1, Hq=Hq, Hkv=Hkv, Sq=S, Skv=past + S,
                              Skv_alloc=self.ctx_max, q_pos0=past, q_strides=(0, d, nq), k_strides=(0, self.ctx_max * d, d),
                              v_strides=(0, d * self.ctx_max, self.ctx_max), o_strides=(0, Hq * d), scale=d ** -0.5, window=self.window, plan=plan)
            if tp == 1:
                ops.gemm(ao, w["o"], M=S, N=H, Kc=Hq * d, residual=h, ldr=H, round_bf16=True, out16=h, plan=plan)
            else:
                ops.gemm(ao, w["o"], M=S, N=H, Kc=Hq * d, out32=part, plan=plan)
                segs += [plan, (lambda t=part: self._all_reduce(t))]
                plan = ops.Plan()
                ops.residual_add(h, part, S * H, plan=plan)
            ops.norm(h, w["ln2"], None, rows=S, C=H, eps=c["rms_norm_eps"], rms=True, round_bf16=True, out16=xn, plan=plan)
            ops.gemm(xn, w["gu"], M=S, N=2 * I, Kc=H, act=ACT_SWIGLU, round_bf16=True, out16=act, ldc=I, plan=plan)
            if tp == 1:
                ops.gemm(act, w["down"], M=S, N=H, Kc=I, residual=h, ldr=H, round_bf16=True, out16=h, plan=plan)
            else:
                ops.gemm(act, w["down"], M=S, N=H, Kc=I, out32=part, plan=plan)
                segs += [plan, (lambda t=part: self._all_reduce(t))]
                plan = ops.Plan()
                ops.residual_add(h, part, S * H, plan=plan)
        self._lm_head_and_pick(plan, h[S - 1], False, segs, sampling, slot=slot)
        segs[0].hold(*[t for s in segs if isinstance(s, ops.Plan) for t in s.keep])
        return segs, io

    def _build_decode_p2p(self):
        """Tensor-parallel decode step with the all-reduces done peer to peer: the launch sequence of the single-GPU step
        over this rank's shards; o_proj / down_proj carry the exchange in their epilogues (fused) or are followed by
        usdm_allreduce_p2p_reduce (split).  Returned as segments cut at every exchange so that a single-process harness can
        interleave logical ranks; a real rank runs them back to back inside one hipGraph."""
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        H, d, L = c["hidden_size"], c["head_dim"], c["num_hidden_layers"]
        Hq, Hkv, I = self.Hq, self.Hkv, self.I
        nq = (Hq + 2 * Hkv) * d
        segs, plan = [], ops.Plan()
        Z = lambda *s, dt=bf: plan.hold(torch.zeros(*s, device=dev, dtype=dt))
        h, qkv, ao, act = self.h_dec, Z(nq), Z(Hq * d), Z(I)
        pm, pl, po = Z(Hq * self.NS, dt=torch.float32), Z(Hq * self.NS, dt=torch.float32), Z(Hq * self.NS * d, dt=torch.float32)
        skp, mode = self.st_done, (1 if self.p2p_fused else 2)

        mrg = (pm, pl, po, self.NS) if self.merge_in_oproj else None

        def row_parallel(plan, W, x, K, site, merge=None):
            ops.gemv(W, x, N=H, K=K, residual=h, y16=h, skip=skp, p2p=self.p2p, p2p_site=site, p2p_mode=mode, merge=merge, plan=plan)
            segs.append(plan)
            plan = ops.Plan()
            if mode == 2:
                ops.p2p_reduce(self.p2p, site, H, h, skip=skp, plan=plan)
            return plan

        for l in range(L):
            w = self.W["layers"][l]
            ops.gemv(w["qkv"], h, N=nq, K=H, norm_w=w["ln1"], eps=c["rms_norm_eps"], y16=qkv, skip=skp, plan=plan)
            ops.attn_decode(qkv, self.st_pos, self.cos, self.sin, self.kcache[l], self.vcache[l], pm, pl, po, ao, Hq=Hq, Hkv=Hkv,
                            ctx_max=self.ctx_max, NS=self.NS, scale=d ** -0.5, skip=skp, defer_merge=mrg is not None, window=self.window, plan=plan)
            plan = row_parallel(plan, w["o"], ao, Hq * d, 2 * l, merge=mrg)
            ops.gemv(w["gu"], h, N=2 * I, K=H, norm_w=w["ln2"], eps=c["rms_norm_eps"], act=ACT_SWIGLU, y16=act, skip=skp, plan=plan)
            plan = row_parallel(plan, w["down"], act, I, 2 * l + 1)
        self._lm_head_and_pick(plan, h, True, segs, None, skip=skp)
        segs[0].hold(*[t for s_ in segs for t in s_.keep])
        return segs

    def _build_decode(self, sampling=None):
        if self.p2p is not None:
            if sampling:
                raise NotImplementedError("sampling needs the full logit row on one GPU (tensor-parallel decode is greedy only)")
            return self._build_decode_p2p()
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        H, d, L = c["hidden_size"], c["head_dim"], c["num_hidden_layers"]
        Hq, Hkv, I, tp = self.Hq, self.Hkv, self.I, (2 if self.tp_path else 1)
        nq = (Hq + 2 * Hkv) * d
        segs, plan = [], ops.Plan()
        Z = lambda *s, dt=bf: plan.hold(torch.zeros(*s, device=dev, dtype=dt))
        h, qkv, ao, act = self.h_dec, Z(nq), Z(Hq * d), Z(I)
        pm, pl, po = Z(Hq * self.NS, dt=torch.float32), Z(Hq * self.NS, dt=torch.float32), Z(Hq * self.NS * d, dt=torch.float32)
        part = Z(H, dt=torch.float32) if tp > 1 else None
        # last-arriver counters of the fused partial merge (self-resetting).  Off by default: measured 11.3-11.6 us per layer
        # against 5.9 + 4.6 us for the split kernel + merge kernel (profiles/r01_decode_ablation.txt)
        cnt = Z(Hkv, dt=torch.int32) if os.environ.get("USDM_ATTN_FUSED_MERGE", "0") == "1" else None
        # Tensor-parallel path: the residual add that follows each all-reduce is folded into the NEXT GEMV's prologue
        # (usdm_gemv x_delta / x_out) instead of a usdm_residual_add launch; the residual stream ping-pongs between two
        # buffers because workgroup 0 publishes the updated stream while the others still read the old one.
        h_alt = Z(H) if tp > 1 else None
        fuse_res = tp > 1 and os.environ.get("USDM_TP_FUSED_RESIDUAL", "1") == "1"
        pend = None   # f32 partial (already all-reduced) not yet added to the residual stream
        part2 = Z(H, dt=torch.float32) if tp > 1 else None

        def flip(cur):
            return h_alt if cur is self.h_dec else self.h_dec

        skp = self.st_done   # decode kernels return at once after a device-side EOS (see _alloc)
        cmb_gran = Z(L, Hq * 64, dt=torch.int64) if self.cmb else None      # one granule block per layer (tags cleared by the layer's attention launch)
        if self.cmb and not hasattr(self, "cmb_err"):
            self.cmb_err = torch.zeros(1, dtype=torch.int32, device=dev)

        if tp == 1 and self.chain in (3, 4) and cnt is None and not self.merge_in_oproj:
            G = lambda *a_, **k_: ops.gemv(*a_, skip=skp, only_args=True, **k_)
            for l in range(L):
                w = self.W["layers"][l]
                if l == 0 or self.chain == 3:
                    ops.gemv(w["qkv"], h, N=nq, K=H, norm_w=w["ln1"], eps=c["rms_norm_eps"], y16=qkv, skip=skp, plan=plan)
                ops.attn_decode(qkv, self.st_pos, self.cos, self.sin, self.kcache[l], self.vcache[l], pm, pl, po, ao, Hq=Hq, Hkv=Hkv,
                                ctx_max=self.ctx_max, NS=self.NS, scale=d ** -0.5, skip=skp, window=self.window, plan=plan)
                ph = [G(w["o"], ao, N=H, K=Hq * d, residual=h, y16=h),
                      G(w["gu"], h, N=2 * I, K=H, norm_w=w["ln2"], eps=c["rms_norm_eps"], act=ACT_SWIGLU, y16=act),
                      G(w["down"], act, N=H, K=I, residual=h, y16=h)]
                if self.chain == 4 and l + 1 < L:
                    w2 = self.W["layers"][l + 1]
                    ph.append(G(w2["qkv"], h, N=nq, K=H, norm_w=w2["ln1"], eps=c["rms_norm_eps"], y16=qkv))
                sync = self.chain_sync[0 if len(ph) == self.chain else 1]
                if self.chain_engine:
                    ops.gemv_engine(ph, sync, self.chain_gran, plan=plan)
                else:
                    ops.gemv_chain(ph, sync, plan=plan)
            self._lm_head_and_pick(plan, h, True, segs, sampling, skip=skp)
            segs[0].hold(*[t for s in segs if isinstance(s, ops.Plan) for t in s.keep])
            return segs
        for l in range(L):   # h already holds the embedding of the current token (written by usdm_argmax_final)
            w = self.W["layers"][l]
            if pend is not None:
                ops.gemv(w["qkv"], h, N=nq, K=H, norm_w=w["ln1"], eps=c["rms_norm_eps"], y16=qkv, x_delta=pend, x_out=flip(h), skip=skp, plan=plan)
                h, pend = flip(h), None
            else:
                ops.gemv(w["qkv"], h, N=nq, K=H, norm_w=w["ln1"], eps=c["rms_norm_eps"], y16=qkv, skip=skp, plan=plan)
            use_cmb = self.cmb and tp == 1 and cnt is None and self.NS > 1 and cmb_gran is not None
            mrg = (pm, pl, po, self.NS) if ((self.merge_in_oproj or use_cmb) and cnt is None) else None
            gran = cmb_gran[l] if use_cmb else None
            ops.attn_decode(qkv, self.st_pos, self.cos, self.sin, self.kcache[l], self.vcache[l], pm, pl, po, ao, Hq=Hq, Hkv=Hkv,
                            ctx_max=self.ctx_max, NS=self.NS, scale=d ** -0.5, counters=cnt, skip=skp, defer_merge=mrg is not None, window=self.window,
                            cmb_gran=gran, plan=plan)
            if tp == 1:
                ops.gemv(w["o"], ao, N=H, K=Hq * d, residual=h, y16=h, skip=skp, merge=mrg, cmb=(gran, self.cmb_err) if use_cmb else None, plan=plan)
            else:
                ops.gemv(w["o"], ao, N=H, K=Hq * d, round_bf16=False, y32=part, skip=skp, merge=mrg, plan=plan)
                segs += [plan, (lambda t=part: self._all_reduce(t))]
                plan = ops.Plan()
                if fuse_res:
                    pend = part
                else:
                    ops.residual_add(h, part, H, plan=plan)
            if pend is not None:
                ops.gemv(w["gu"], h, N=2 * I, K=H, norm_w=w["ln2"], eps=c["rms_norm_eps"], act=ACT_SWIGLU, y16=act, x_delta=pend,
                         x_out=flip(h), skip=skp, plan=plan)
                h, pend = flip(h), None
            else:
                ops.gemv(w["gu"], h, N=2 * I, K=H, norm_w=w["ln2"], eps=c["rms_norm_eps"], act=ACT_SWIGLU, y16=act, skip=skp, plan=plan)
            if tp == 1:
                ops.gemv(w["down"], act, N=H, K=I, residual=h, y16=h, skip=skp, plan=plan)
            else:
                ops.gemv(w["down"], act, N=H, K=I, round_bf16=False, y32=part2, skip=skp, plan=plan)
                segs += [plan, (lambda t=part2: self._all_reduce(t))]
                plan = ops.Plan()
                if fuse_res:
                    pend = part2
                else:
                    ops.residual_add(h, part2, H, plan=plan)
        if pend is not None:   # the last down-projection's sum goes into the final norm + lm_head
            self._lm_head_and_pick(plan, h, True, segs, sampling, x_delta=pend, skip=skp)
        else:
            self._lm_head_and_pick(plan, h, True, segs, sampling, skip=skp)
        segs[0].hold(*[t for s in segs if isinstance(s, ops.Plan) for t in s.keep])
        return segs

    @staticmethod
    def _run_segs(segs):
        for s in segs:
            if isinstance(s, ops.Plan):
                s.run()
            else:
                s()

    # ------------------------------------------------------------------ batched decode
    class _Slot:
        """Views of the batch buffers that stand in for the single-sequence attributes during a per-item prefill."""

    def _batch_buffers(self, B):
        if B in self._batches:
            return self._batches[B]
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        L, d, H = c["num_hidden_layers"], c["head_dim"], c["hidden_size"]
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
        bb = dict(kc=torch.zeros(B, L, self.Hkv, self.ctx_max, d, dtype=bf, device=dev),
                  vc=torch.zeros(B, L, self.Hkv, self.ctx_max, d, dtype=bf, device=dev),
                  nxt=i32(B), step=i32(B), pos=i32(B), out=i32(B, self.max_out), h=torch.zeros(B, H, dtype=bf, device=dev),
                  pv=torch.zeros(B, self.nparts, dtype=torch.float32, device=dev), pi=i32(B, self.nparts), prefill=LRU(16), decode=None,
                  decode_sampled=None, logits=torch.zeros(B, self.v1 - self.v0, dtype=torch.float32, device=dev),
                  sp=ops.sample_params_tensor(dev, B).view(B, -1))
        if self.tp_path:
            # tensor parallel: pv / pi hold THIS rank's partials; the decode step gathers them rank-major into pvg / pig
            # [rank][sequence][nparts] (usdm_argmax_final_seg), a slot's prefill into its own row of pvs / pis [sequence][rank * nparts]
            tp = self.tp_size
            bb.update(pvg=torch.zeros(tp, B, self.nparts, dtype=torch.float32, device=dev), pig=i32(tp, B, self.nparts),
                      pvs=torch.zeros(B, tp * self.nparts, dtype=torch.float32, device=dev), pis=i32(B, tp * self.nparts))
        slots = []
        for b in range(B):
            sl = self._Slot()
            sl.kcache, sl.vcache = bb["kc"][b], bb["vc"][b]
            sl.st_next, sl.st_step, sl.st_pos, sl.st_out = bb["nxt"][b:b + 1], bb["step"][b:b + 1], bb["pos"][b:b + 1], bb["out"][b]
            sl.h_dec = bb["h"][b]
            sl.part_val = sl.part_val_loc = bb["pv"][b]
            sl.part_idx = sl.part_idx_loc = bb["pi"][b]
            if self.tp_path:
                sl.part_val, sl.part_idx = bb["pvs"][b], bb["pis"][b]
            sl.logits, sl.sample_params = bb["logits"][b], bb["sp"][b]
            slots.append(sl)
        bb["slots"] = slots
        self._batches[B] = bb
        return bb

    def _build_decode_batch(self, B, sampling=False):
        """One decode step of B sequences: weights streamed once (usdm_gemv_batch), attention / token pick batched over items.
        sampling: the pick is usdm_sample_final's batched form - every slot draws with its OWN knobs (bb["sp"][b]: temperature,
        top-k, top-p, seed) and its own Philox counter; a greedy slot carries top_k = 1."""
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        H, d, L = c["hidden_size"], c["head_dim"], c["num_hidden_layers"]
        Hq, Hkv, I = self.Hq, self.Hkv, self.I
        nq = (Hq + 2 * Hkv) * d
        bb = self._batch_buffers(B)
        plan = ops.Plan()
        Z = lambda *s, dt=bf: plan.hold(torch.zeros(*s, device=dev, dtype=dt))
        h, qkv, ao, act = bb["h"], Z(B, nq), Z(B, Hq * d), Z(B, I)
        NS = max(2, self.NS)
        if B > 4:
            # many sequences: B x Hkv x NS workgroups of ctx / NS keys each.  The batch-1 choice (32 splits of ~20 keys: latency-bound,
            # one per CU) would be 4096 tiny workgroups at B = 16 (measured 35.6 us per layer); ~512 workgroups with up to 512 keys
            # (the split kernel's LDS bound) keep the KV stream at full width.  (B <= 4 keeps the batch-1 splits: bit-identical.)
            NS = max(-(-self.ctx_max // 512), min(self.NS, max(2, getattr(self, "batch_attn_wgs", 512) // (B * Hkv))))
        pm, pl, po = Z(B * Hq * NS, dt=torch.float32), Z(B * Hq * NS, dt=torch.float32), Z(B * Hq * NS * d, dt=torch.float32)
        # (tools/batch_rate.py A/B: the merge by the last-arriving workgroup of a kv head instead of the combine launch)
        cnt = Z(B * Hkv, dt=torch.int32) if B > 4 and getattr(self, "batch_fused_merge", False) else None
        cache_bs = L * Hkv * self.ctx_max * d
        # down_proj on the matrix cores (K = 14336): K split over workgroups, each holding its activation slice (usdm_gemv_batch ks_*);
        # one scratch for all layers - the launches of a step are serial and each leaves the counters zero
        ksf = ops.gemv_batch_ks_floats(H, I) if B > 4 else 0
        ks = (Z(ksf, dt=torch.float32), Z(-(-H // 16), dt=torch.int32)) if ksf else None
        # Tensor parallel (SURVEY.md 8e x 8f-2; round 4, the collective form): the row-parallel projections leave f32 partial sums
        # [B][H], all-reduced through the job's process group (RCCL: one collective of B x 16 KB per projection instead of B of them),
        # then usdm_residual_add applies HF's rounding points; the plan is cut into segments at the collectives, as the
        # single-sequence RCCL path is.  Greedy only (the sampling kernel needs the full logit row on one GPU).
        tp = self.tp_path
        if tp and sampling:
            raise NotImplementedError("sampling needs the full logit row on one GPU (tensor-parallel decode is greedy only)")
        segs = []
        part, part2 = (Z(B, H, dt=torch.float32), Z(B, H, dt=torch.float32)) if tp else (None, None)

        def row_parallel(W, x, K, buf, plan):
            ops.gemv_batch(W, x, nb=B, N=H, K=K, x_bs=K, y_bs=H, round_bf16=False, y32=buf, plan=plan)
            segs.extend([plan, (lambda t=buf: self._all_reduce(t))])
            plan = ops.Plan()
            ops.residual_add(h, buf, B * H, plan=plan)
            return plan
        for l in range(L):
            w = self.W["layers"][l]
            ops.gemv_batch(w["qkv"], h, nb=B, N=nq, K=H, x_bs=H, y_bs=nq, norm_w=w["ln1"], eps=c["rms_norm_eps"], y16=qkv, plan=plan)
            ops.attn_decode(qkv, bb["pos"], self.cos, self.sin, bb["kc"][0, l], bb["vc"][0, l], pm, pl, po, ao, Hq=Hq, Hkv=Hkv,
                            ctx_max=self.ctx_max, NS=NS, scale=d ** -0.5, batch=B, qkv_bs=nq, out_bs=Hq * d, cache_bs=cache_bs, window=self.window,
                            counters=cnt, plan=plan)
            if tp:
                plan = row_parallel(w["o"], ao, Hq * d, part, plan)
            else:
                ops.gemv_batch(w["o"], ao, nb=B, N=H, K=Hq * d, x_bs=Hq * d, y_bs=H, res_bs=H, residual=h, y16=h, plan=plan)
            ops.gemv_batch(w["gu"], h, nb=B, N=2 * I, K=H, x_bs=H, y_bs=I, norm_w=w["ln2"], eps=c["rms_norm_eps"], act=ACT_SWIGLU,
                           y16=act, plan=plan)
            if tp:
                plan = row_parallel(w["down"], act, I, part2, plan)
            else:
                ops.gemv_batch(w["down"], act, nb=B, N=H, K=I, x_bs=I, y_bs=H, res_bs=H, residual=h, y16=h, ks=ks, plan=plan)
        ops.gemv_batch(self.W["lm_head"], h, nb=B, N=self.v1 - self.v0, K=H, x_bs=H, part_bs=self.nparts, norm_w=self.W["norm"],
                       eps=c["rms_norm_eps"], ban=self.ban, part_val=bb["pv"], part_idx=bb["pi"], idx_offset=self.v0,
                       **(dict(y32=bb["logits"], y_bs=self.v1 - self.v0) if sampling else {}), plan=plan)
        st = ops.decode_state(bb["nxt"], bb["out"], bb["step"], bb["pos"], advance_pos=True, batch=B)
        if sampling:
            ops.sample_final(bb["logits"], st, dev_params=bb["sp"], embed=self.W["embed"], h_out=h, Hd=H, plan=plan)
        elif tp:      # vocab-parallel pick: the ranks' [B][nparts] partials gathered rank-major, one pick per sequence over all of them
            segs.extend([plan, (lambda: self._gather_partials([bb["pvg"], bb["pig"]], [bb["pv"], bb["pi"]]))])
            plan = ops.Plan()
            ops.argmax_final(bb["pvg"], bb["pig"], self.nparts, st, embed=self.W["embed"], h_out=h, Hd=H, nseg=self.tp_size,
                             seg_stride=B * self.nparts, plan=plan)
        else:
            ops.argmax_final(bb["pv"], bb["pi"], self.nparts, st, embed=self.W["embed"], h_out=h, Hd=H, plan=plan)
        plan.hold(st)
        if tp:
            segs.append(plan)
            segs[0].hold(*[t for sg in segs if isinstance(sg, ops.Plan) for t in sg.keep])
            return segs
        return plan

    MAX_BATCH = 16      # sequences per decode step (usdm_gemv_batch: VALU form up to 4, matrix-core form up to 16)

    def max_batch(self):
        """Sequences one decode step can take with THIS model's (per-rank) shapes: 16 on the matrix-core form, which splits K over
        8 waves in chunks of 32 (every projection's K must be a multiple of 256), else the 4 of the VALU form."""
        c = self.cfg
        ks = (c["hidden_size"], self.Hq * c["head_dim"], self.I)
        return self.MAX_BATCH if all(k % 256 == 0 for k in ks) else 4

    @torch.no_grad()
    def generate_batch(self, input_ids_list, max_new_tokens, bad_words_ids=None, eos_token_id=None, min_new_tokens=0, group=None):
        """Greedy generation of several utterances in lockstep (the serving-side batching of inference_vllm.py:109-125): up to
        `group` (default 16) sequences per step, longer lists run in groups.  Each prompt is prefilled on its own; every decode
        step then streams the weights once for the whole group.  Groups of <= 4 run on the VALU kernel and equal generate() per
        sequence bit for bit; larger groups run on the matrix cores (usdm_gemv_batch form 1): the same rounding points, K summed
        in another order - equal to the oracle up to its near-ties, not bit-identical with generate()."""
        group = self.max_batch() if group is None else max(1, min(int(group), self.max_batch()))
        outs = []
        for g0 in range(0, len(input_ids_list), group):
            outs += self._generate_group(input_ids_list[g0:g0 + group], max_new_tokens, bad_words_ids, eos_token_id, min_new_tokens)
        return outs

    def _generate_group(self, ids_list, max_new_tokens, bad_words_ids, eos_token_id, min_new_tokens):
        B = len(ids_list)
        for ids in ids_list:
            if ids.dim() != 2 or ids.shape[0] != 1:
                raise ValueError("every prompt must be a LongTensor of shape [1, L]")
        bb = self._batch_buffers(B)
        self.ban.copy_(self._ban_mask(bad_words_ids))
        L0 = [int(ids.shape[1]) for ids in ids_list]
        max_new = min(max_new_tokens, self.ctx_max - max(L0), self.max_out)
        if max_new <= 0:
            return [ids.clone() for ids in ids_list]
        bb["step"].zero_()
        bb["pos"].copy_(torch.tensor(L0, dtype=torch.int32))
        for b, ids in enumerate(ids_list):            # per-item prefill into that item's cache / state slot (+ first token)
            key = (L0[b], b)
            segs, io = bb["prefill"].get_or_build(key, lambda: self._build_prefill(L0[b], None, slot=bb["slots"][b]))
            io["ids"].copy_(ids[0])
            self._run_segs(segs)
        if bb["decode"] is None:
            built = self._build_decode_batch(B)
            bb["decode"] = GraphedSegments(built, self._run_segs) if isinstance(built, list) else GraphedPlan(built)
        eos = set(eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id]) if eos_token_id is not None else set()
        produced, chunk = 1, 8
        ends = [None] * B
        while True:
            toks = bb["out"][:, :produced].tolist()     # host sync point (EOS check)
            for b in range(B):
                if ends[b] is None:
                    hit = [i for i, t in enumerate(toks[b]) if t in eos and i + 1 >= min_new_tokens]
                    if hit:
                        ends[b] = hit[0] + 1
            if all(e is not None for e in ends) or produced >= max_new:
                break
            n = min(chunk, max_new - produced)
            for _ in range(n):
                bb["decode"].run()
            produced += n
        res = []
        for b, ids in enumerate(ids_list):
            n = ends[b] if ends[b] is not None else min(produced, max_new)
            res.append(torch.cat([ids[0], torch.tensor(toks[b][:n], dtype=torch.long, device=ids.device)]).unsqueeze(0))
        return res

    # ------------------------------------------------------------------ generate
    def _ban_mask(self, bad_words_ids):
        if not bad_words_ids:
            return self.ban_all_off
        key = id(bad_words_ids)
        hit = self._ban_cache.get(key)
        if hit is not None and hit[0] is bad_words_ids:
            return hit[1]
        m = torch.zeros(self.cfg["vocab_size"], dtype=torch.uint8)
        for w in bad_words_ids:
            if len(w) != 1:
                raise NotImplementedError("multi-token bad words are not used by the reference path (inference.py:41-45)")
            m[w[0]] = 1
        t = m[self.v0:self.v1].to(self.device).contiguous()
        self._ban_cache.put(key, (bad_words_ids, t))
        return t

    def _setup_call(self, input_ids, past, sampling, bad_words_ids, eos_token_id, min_new_tokens, ban_mask=None):
        """Per-call device state of generate(): prompt ids into the (cached) prefill plan, ban mask, position / step counters,
        device-side EOS list.  Returns (prefill segments, the EOS ids the device checks)."""
        L0 = input_ids.shape[1]
        key = (L0 - past, past, sampling)
        segs, io = self._prefill_plans.get_or_build(key, lambda: self._build_prefill(L0 - past, sampling, past=past))
        io["ids"].copy_(input_ids[0, past:])
        self._kv_ids, self._vt_upto = None, L0      # (set again once this call's decode steps are known)
        if ban_mask is not None:      # a ready-made [vocab] 0/1 mask (usdm_amd.serving: static logits processors)
            self.ban.copy_(ban_mask.to(torch.uint8)[self.v0:self.v1])
        else:
            self.ban.copy_(self._ban_mask(bad_words_ids))
        self.st_pos.fill_(L0)
        self.st_step.zero_()
        eos_list = sorted(set(eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id])) if eos_token_id is not None else []
        dev_eos = eos_list if len(eos_list) <= 6 else []      # more ids than the device list holds: host-side check only
        self.st_eos.copy_(torch.tensor(([len(dev_eos), int(min_new_tokens)] + dev_eos + [0] * 6)[:8], dtype=torch.int32))
        self.st_done.zero_()
        return segs, dev_eos

    @torch.no_grad()
    def generate(self, input_ids=None, max_length=None, do_sample=False, bad_words_ids=None, top_p=1.0, top_k=None,
                 temperature=1.0, eos_token_id=None, max_new_tokens=None, min_new_tokens=0, seed=None, ban_mask=None,
                 _logits_hook=None, **unused):
        """Generation with the call shape of src/inference.py:63-83.  Greedy when do_sample is False or top_k == 1 (what the
        reference passes: arg-max of the ban-masked logits).  Otherwise temperature / top-k / top-p sampling on the device
        (usdm_sample_final).  `seed` keys its Philox stream; seed=None draws a fresh one from torch's global CPU generator,
        so calls differ from each other as HF sampling does and are reproducible under torch.manual_seed."""
        if input_ids is None or input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise ValueError("input_ids must be a LongTensor of shape [1, L] (batch 1, as the reference calls it)")
        sampling = False
        if _logits_hook is not None:      # arbitrary Python logits processors: eager steps, knobs still on the device
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            sampling, self.logits_hook = "hook", _logits_hook
            ops.set_sample_params(self.sample_params, temperature if do_sample else 1.0, int(top_k or 0) if do_sample else 1,
                                  top_p if do_sample else 1.0, seed)
        elif do_sample and top_k != 1:
            if not (temperature > 0) or not (0 < top_p <= 1):
                raise ValueError("temperature must be > 0 and top_p in (0, 1]")
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            sampling = True
            ops.set_sample_params(self.sample_params, temperature, int(top_k or 0), top_p, seed)
        elif temperature != 1.0 or top_p != 1.0:
            if not do_sample:
                raise ValueError("temperature / top_p only apply with do_sample=True")
        L0 = input_ids.shape[1]
        if max_new_tokens is None:
            if max_length is None:
                raise ValueError("max_length or max_new_tokens is required")
            max_new_tokens = max_length - L0
        max_new_tokens = min(max_new_tokens, self.ctx_max - L0, self.max_out)
        if max_new_tokens <= 0:
            return input_ids.clone()
        # prefix reuse: tokens whose K/V are already cached (same ids at the same positions) are not prefilled again
        past, ids_host = 0, None
        if self.reuse_prefix:
            ids_host = input_ids[0].tolist()
            if self._kv_ids is not None:
                n = min(len(ids_host) - 1, len(self._kv_ids))
                while past < n and ids_host[past] == self._kv_ids[past]:
                    past += 1
                if self.reuse_prefix == "exact":
                    past = min(past, self._vt_upto)      # rows beyond that were appended by decode steps
                if past < 16:
                    past = 0
            if past > self._vt_upto:   # K/V appended by decode steps have no V^T yet: one transposed copy over all layers
                a0 = self._vt_upto
                self.vtc[:, :, :, a0:past] = self.vcache[:, :, a0:past, :].transpose(2, 3)
        segs, dev_eos = self._setup_call(input_ids, past, sampling, bad_words_ids, eos_token_id, min_new_tokens, ban_mask=ban_mask)
        self._run_segs(segs)  # prefill + first token
        if sampling not in self._decodes:
            dsegs = self._build_decode(sampling)
            if self.p2p is not None:      # kernels only (the exchange lives inside them): one plan, one hipGraph
                merged = ops.Plan()
                for s_ in dsegs:
                    merged.calls += s_.calls
                    merged.hold(*s_.keep)
                dsegs = [merged]
            if sampling == "hook":
                self._decodes[sampling] = GraphedSegments(dsegs, self._run_segs, enabled=False)      # host code inside: never captured
            else:
                self._decodes[sampling] = GraphedPlan(dsegs[0]) if (len(dsegs) == 1) else GraphedSegments(dsegs, self._run_segs)
        self._decode = self._decodes[sampling]
        eos = set(eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id]) if eos_token_id is not None else set()
        produced, done, chunk = 1, False, 8
        toks = []
        while True:
            n_dev = int(self.st_step.item()) if dev_eos else produced   # steps past a device-side EOS did not run
            produced = min(produced, n_dev)
            toks = self.st_out[:produced].tolist()  # host sync point (EOS check)
            if self.p2p is not None:
                self.p2p.raise_if_failed()          # a peer that never delivered surfaces here, not as a hang
            if self.chain and int(self.chain_sync[:, 1].sum().item()):
                raise RuntimeError("usdm_gemv_chain: a grid barrier timed out (the persistent decode kernel was not fully resident); "
                                   "results are invalid - rerun with USDM_GEMV_CHAIN=0")
            if getattr(self, "cmb_err", None) is not None and int(self.cmb_err.item()):
                raise RuntimeError("usdm_gemv (cmb_gran): the in-launch attention hand-off timed out (the o_proj launch was not fully "
                                   "resident); results are invalid - rerun with USDM_ATTN_CMB=0")
            hit = [i for i, t in enumerate(toks) if t in eos and i + 1 >= min_new_tokens]
            if hit:
                toks = toks[:hit[0] + 1]
                break
            if produced >= max_new_tokens:
                toks = toks[:max_new_tokens]
                break
            n = min(chunk, max_new_tokens - produced)
            for _ in range(n):
                self._decode.run()
            produced += n
        if self.reuse_prefix:   # ids whose K/V now sit in the cache: the prompt and every generated token that was fed back
            fed = self.st_out[:produced - 1].tolist() if produced > 1 else []
            self._kv_ids = ids_host + fed
        out = torch.cat([input_ids[0], torch.tensor(toks, dtype=torch.long, device=input_ids.device)])
        return out.unsqueeze(0)
