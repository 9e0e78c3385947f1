"""Oracle: Mistral-7B-style causal LM forward + greedy generate, restated functionally on CPU torch
(TEST INFRASTRUCTURE).

The reference's LLM arithmetic is THIRD-PARTY: HF `transformers==4.40.2` MistralForCausalLM +
GenerationMixin.generate (+ flash-attn 2.6.3), called at src/inference.py:63-83 and loaded at :116-124
(bf16).  None of it is in /root/reference.  This file restates the published HF algorithm
(modeling_mistral.py: MistralRMSNorm, MistralRotaryEmbedding/apply_rotary_pos_emb, MistralAttention with
repeat_kv, MistralMLP, logits in fp32) with the same torch ops in the same dtype, so running it in bf16
reproduces HF's rounding points.  Pinning: tests/test_oracle_cpu.py compares it with the `transformers`
build installed in this image (5.x, eager attention) on a small random-init model — the reference's own
tests pin nothing at this boundary, and the pinned 4.40.2 wheel is absent ("parity unpinned" beyond that).
Token-id layout / masks: src/train_pt.py:104-128, src/inference.py:41-53.
"""
import math

import torch
import torch.nn.functional as F

MISTRAL_7B_USDM = dict(vocab_size=42003, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                       num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5,
                       rope_theta=10000.0, max_position_embeddings=32768)


def rms_norm(x, w, eps):
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_tables(cfg, positions, dtype):
    d = cfg["head_dim"]
    inv_freq = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def forward(sd, cfg, ids, cache=None):
    """ids int64 [S] (new tokens) ; cache: list of (k,v) [Hkv,T,d] per layer or None. -> logits f32 [S,V], cache"""
    H, nh, nkv, d = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    dt = sd["model.embed_tokens.weight"].dtype
    S = ids.numel()
    past = 0 if cache is None else cache[0][0].shape[1]
    pos = torch.arange(past, past + S)
    cos, sin = rope_tables(cfg, pos, dt)
    h = F.embedding(ids, sd["model.embed_tokens.weight"])
    new_cache = []
    T = past + S
    mask = torch.full((S, T), torch.finfo(dt).min, dtype=dt)
    vis = torch.arange(T)[None, :] <= pos[:, None]
    W = cfg.get("sliding_window")
    if W:       # HF sliding-window causal mask: query q sees the W keys q-W+1 .. q (reference: src/model.py:337-371 keeps W-1 past keys + the new one)
        vis = vis & (torch.arange(T)[None, :] > pos[:, None] - W)
    mask = mask.masked_fill(vis, 0)
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        x = rms_norm(h, sd[p + "input_layernorm.weight"], cfg["rms_norm_eps"])
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"]).view(S, nh, d).transpose(0, 1)
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"]).view(S, nkv, d).transpose(0, 1)
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"]).view(S, nkv, d).transpose(0, 1)
        q = (q * cos) + (rotate_half(q) * sin)
        k = (k * cos) + (rotate_half(k) * sin)
        if cache is not None:
            k = torch.cat([cache[l][0], k], 1)
            v = torch.cat([cache[l][1], v], 1)
        new_cache.append((k, v))
        kk = k.repeat_interleave(nh // nkv, 0)
        vv = v.repeat_interleave(nh // nkv, 0)
        w = torch.matmul(q, kk.transpose(1, 2)) * (d ** -0.5)
        w = w + mask
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(dt)
        o = torch.matmul(w, vv).transpose(0, 1).reshape(S, nh * d)
        h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        x = rms_norm(h, sd[p + "post_attention_layernorm.weight"], cfg["rms_norm_eps"])
        m = F.linear(F.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) * F.linear(x, sd[p + "mlp.up_proj.weight"]),
                     sd[p + "mlp.down_proj.weight"])
        h = h + m
    h = rms_norm(h, sd["model.norm.weight"], cfg["rms_norm_eps"])
    logits = F.linear(h, sd["lm_head.weight"]).float()
    return logits, new_cache


def ban_mask(vocab, bad_words_ids):
    m = torch.zeros(vocab, dtype=torch.bool)
    for w in bad_words_ids or []:
        assert len(w) == 1, "only single-token bad words occur on the path (inference.py:41-45)"
        m[w[0]] = True
    return m


def greedy_generate(sd, cfg, ids, max_new_tokens, bad_words_ids=None, eos_token_id=None, return_logits=False):
    """HF generate(do_sample=True, top_k=1, top_p=1, temperature=1) == arg-max of the masked logits
    (up to exact ties).  Returns the full id list (prompt + generated, EOS included)."""
    ban = ban_mask(cfg["vocab_size"], bad_words_ids)
    out = list(ids.tolist())
    logits, cache = forward(sd, cfg, ids)
    all_logits = []
    for _ in range(max_new_tokens):
        last = logits[-1].clone()
        last[ban] = -float("inf")
        all_logits.append(last)
        tok = int(torch.argmax(last))
        out.append(tok)
        if eos_token_id is not None and tok == eos_token_id:
            break
        logits, cache = forward(sd, cfg, torch.tensor([tok]), cache)
    return (out, torch.stack(all_logits)) if return_logits else out


def random_state_dict(cfg, seed=0, dtype=torch.bfloat16, std=0.02):
    g = torch.Generator().manual_seed(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv, d = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    r = lambda *s, sc=std: (torch.randn(*s, generator=g) * sc).to(dtype)
    sd = {"model.embed_tokens.weight": r(V, H, sc=1.0), "lm_head.weight": r(V, H, sc=H ** -0.5),
          "model.norm.weight": (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)}
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        sd[p + "self_attn.q_proj.weight"] = r(nh * d, H, sc=H ** -0.5)
        sd[p + "self_attn.k_proj.weight"] = r(nkv * d, H, sc=H ** -0.5)
        sd[p + "self_attn.v_proj.weight"] = r(nkv * d, H, sc=H ** -0.5)
        sd[p + "self_attn.o_proj.weight"] = r(H, nh * d, sc=(nh * d) ** -0.5)
        sd[p + "mlp.gate_proj.weight"] = r(I, H, sc=H ** -0.5)
        sd[p + "mlp.up_proj.weight"] = r(I, H, sc=H ** -0.5)
        sd[p + "mlp.down_proj.weight"] = r(H, I, sc=I ** -0.5)
        sd[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
        sd[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    return sd
