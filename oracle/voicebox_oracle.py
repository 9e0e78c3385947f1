"""Oracle: token-Voicebox estimator + CFM solvers restated functionally on CPU torch (TEST INFRASTRUCTURE).

Follows (src/decoder/voicebox/ in the reference):
  model/networks.py:302-374  Transformer.forward        model/networks.py:162-210  Attention.forward
  model/networks.py:13-28    SinusoidalPosEmb           model/networks.py:67-95    PositionalConvEmbedding
  model/networks.py:99-115   get_slopes                 model/networks.py:236-266  EncoderLayer.forward
  model/voicebox.py:51-72    CFM.sample (CFG)           model/voicebox.py:74-150   solve_euler/solve_heun/generate
Pinned by tests/golden/voicebox_*.npz (reference classes imported in the build container, random-init,
fixed seeds, caller-supplied noise).  Noise is an explicit argument because device RNG != CPU RNG.
"""
import math

import torch
import torch.nn.functional as F

VOICEBOX_CFG = dict(  # configs/YOUR_DATA_NAME/config.json:6-32
    n_feats=80, n_tokens=10000, embedding_dim=1280, hidden_size=1024, intermediate_size=4096,
    num_attention_heads=16, num_hidden_layers=24, convpos_width=31, convpos_groups=16, convpos_depth=2,
    sigma_min=1e-4)


def get_slopes(n):
    """networks.py:99-115."""
    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + get_slopes(2 * c)[0::2][: n - c]


def posconv_weight(sd, prefix):
    """weight-norm (dim=2) parametrisation folded: g * v / ||v|| with the norm over dims (0,1)."""
    if prefix + ".conv.weight" in sd:
        return sd[prefix + ".conv.weight"]
    g = sd[prefix + ".conv.parametrizations.weight.original0"]
    v = sd[prefix + ".conv.parametrizations.weight.original1"]
    return g * v / v.norm(dim=(0, 1), keepdim=True)


def time_embedding(t, dim, scale=1000.0):
    """networks.py:19-28 ; t [B] -> [B, dim]."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = scale * t.view(-1, 1) * e.view(1, -1)
    return torch.cat([e.sin(), e.cos()], dim=-1)


def estimator_forward(sd, cfg, x, y, cond, t, lengths, pre="estimator."):
    """x int64 [B,S], y/cond f32 [B,80,S], t f32 [B,1,1] (or [B]), lengths int64 [B] -> f32 [B,80,S]."""
    H, nh, L = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_hidden_layers"]
    hd = H // nh
    B, _, S1 = y.shape
    emb = F.embedding(x, sd[pre + "embed.weight"]) * math.sqrt(cfg["embedding_dim"])
    inp = torch.cat([emb.transpose(1, 2), y, cond], dim=1)
    h = F.conv1d(inp, sd[pre + "proj_in.weight"], sd[pre + "proj_in.bias"])          # [B,H,S]
    S = S1 + 1
    lengths = lengths + 1
    te = time_embedding(t.reshape(B), H)                                              # [B,H]
    h = torch.cat([te.unsqueeze(-1), h], dim=-1).transpose(1, 2).contiguous()         # [B,S,H]
    valid = torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)                       # [B,S]
    ymask = valid.float()
    slope = -torch.tensor(get_slopes(nh))
    r = torch.arange(S)
    alibi = slope.view(nh, 1, 1) * (r.view(1, -1) - r.view(-1, 1)).abs().unsqueeze(0).float()
    alibi[:, :, 0] = 0
    alibi = alibi.unsqueeze(0).expand(B, nh, S, S)
    h = h * ymask.unsqueeze(-1)
    alibi = alibi * ymask[:, None, None, :]
    amask = (1.0 - ymask[:, None, None, :]) * torch.finfo(torch.float32).min
    res = h
    for i in range(cfg["convpos_depth"]):
        p = f"{pre}pos_conv_embeds.{i}"
        w = posconv_weight(sd, p)
        c = F.conv1d(h.transpose(1, 2), w, sd[p + ".conv.bias"], padding=cfg["convpos_width"] // 2,
                     groups=cfg["convpos_groups"])
        if cfg["convpos_width"] % 2 == 0:
            c = c[:, :, :-1]
        h = F.gelu(c).transpose(1, 2) * ymask.unsqueeze(-1)
    h = h + res
    h = F.layer_norm(h, (H,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"], 1e-5)
    h = h * ymask.unsqueeze(-1)
    skips = [h]

    def layer(h, n):
        p = f"{pre}layers.{n}."
        q = F.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]) * hd ** -0.5
        k = F.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"])
        v = F.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"])
        sh = lambda z: z.view(B, S, nh, hd).transpose(1, 2)
        w = sh(q) @ sh(k).transpose(-1, -2) + alibi + amask
        w = torch.softmax(w, dim=-1)
        o = (w @ sh(v)).transpose(1, 2).reshape(B, S, H)
        o = F.linear(o, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        h = (h + o) * ymask.unsqueeze(-1)
        h = F.layer_norm(h, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
        f = F.linear(h, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"])
        f = F.linear(F.gelu(f), sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
        h = (h + f) * ymask.unsqueeze(-1)
        h = F.layer_norm(h, (H,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
        return h * ymask.unsqueeze(-1)

    for n in range(L):
        if n < L // 2:
            h = layer(h * ymask.unsqueeze(-1), n)
            if n < L // 2 - 1:
                skips.append(h)
        else:
            s = skips.pop()
            p = f"{pre}skip_connections_layers.{n - L // 2}."
            h = F.linear(torch.cat([h, s], dim=-1), sd[p + "weight"], sd[p + "bias"])
            h = layer(h * ymask.unsqueeze(-1), n)
    h = (h * ymask.unsqueeze(-1)).transpose(1, 2)
    out = F.conv1d(h, sd[pre + "proj_out.weight"], sd[pre + "proj_out.bias"]) * ymask.unsqueeze(1)
    return out[:, :, 1:]


def cfg_velocity(sd, cfg, x, z, cond, lengths, t, gradient_scale, speech_prompt, f=None):
    """voicebox.py:51-72 ; returns dphi_dt [B,80,S]."""
    f = f or (lambda *a: estimator_forward(sd, cfg, *a))
    B = z.shape[0]
    t = t.reshape(1).expand(B).reshape(B, 1, 1) if t.numel() == 1 else t.view(B, 1, 1)
    if not speech_prompt:
        cond = cond * 0
    if gradient_scale > 0:
        xx = torch.cat([cfg["n_tokens"] * torch.ones_like(x), x], 0)
        v = f(xx, torch.cat([z, z], 0), torch.cat([torch.zeros_like(cond), cond], 0), torch.cat([t, t], 0),
              torch.cat([lengths, lengths], 0))
        vu, vc = torch.chunk(v, 2, 0)
        return vc + gradient_scale * (vc - vu)
    return f(x, z, cond, t, lengths)


def generate(sd, cfg, x, cond, lengths, n_timesteps, noise, solver="euler", gradient_scale=0.0,
             speech_prompt=False, prompt_lengths=None, f=None, trace=None):
    """voicebox.py:140-150 + solve_euler :74-99 / solve_heun :101-138.

    `noise` is the list of N(0,1) tensors the reference would draw with randn_like, in call order:
    noise[0] = z0, then one per prompt re-noising."""
    sigma = cfg["sigma_min"]
    it = iter(noise)
    z = next(it).clone()
    n = (n_timesteps + 1) // 2 if solver == "heun" else n_timesteps
    t_span = torch.linspace(0, 1, n + 1)
    t, dt = t_span[0], t_span[1] - t_span[0]
    P = int(prompt_lengths[0]) if speech_prompt else 0

    def renoise(zz, t):
        eps = next(it)
        pr = (1 - (1 - sigma) * t) * eps + t * cond
        zz[:, :, :P] = pr[:, :, :P]
        return zz

    def vel(zz, tt):
        v = cfg_velocity(sd, cfg, x, zz, cond, lengths, tt, gradient_scale, speech_prompt, f)
        if trace is not None:      # per-NFE velocities (tests print the error growth over chained NFEs)
            trace.append(v.clone())
        return v
    steps = 1
    while steps <= len(t_span) - 1:
        v = vel(z, t)
        z_hat = z + dt * v
        t = t + dt
        if speech_prompt:
            z_hat = renoise(z_hat, t)
        if solver == "heun" and steps < len(t_span) - 1:
            v2 = vel(z_hat, t)
            z_hat = z + dt * (v + v2) / 2
            if speech_prompt:
                z_hat = renoise(z_hat, t)
        z = z_hat
        if steps < len(t_span) - 1:
            dt = t_span[steps + 1] - t
        steps += 1
    return z


def noise_count(n_timesteps, solver, speech_prompt):
    n = (n_timesteps + 1) // 2 if solver == "heun" else n_timesteps
    if not speech_prompt:
        return 1
    return 1 + (2 * n - 1 if solver == "heun" else n)


def random_state_dict(cfg, seed=0, pre="estimator."):
    """Random weights with the reference's key names/shapes (weight-norm parametrisation kept)."""
    g = torch.Generator().manual_seed(seed)
    H, I, E, F_ = cfg["hidden_size"], cfg["intermediate_size"], cfg["embedding_dim"], cfg["n_feats"]
    sd = {}

    def lin(name, o, i, shape=None):
        sd[name + ".weight"] = (torch.randn(shape or (o, i), generator=g) / math.sqrt(i))
        sd[name + ".bias"] = torch.randn(o, generator=g) * 0.05

    def ln(name):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(H, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(H, generator=g)

    sd[pre + "embed.weight"] = torch.randn(cfg["n_tokens"] + 1, E, generator=g) * 0.05
    lin(pre + "proj_in", H, 2 * F_ + E, (H, 2 * F_ + E, 1))
    W, G = cfg["convpos_width"], cfg["convpos_groups"]
    for i in range(cfg["convpos_depth"]):
        p = f"{pre}pos_conv_embeds.{i}.conv"
        sd[p + ".parametrizations.weight.original0"] = 1 + 0.2 * torch.rand(1, 1, W, generator=g)
        sd[p + ".parametrizations.weight.original1"] = torch.randn(H, H // G, W, generator=g)
        sd[p + ".bias"] = torch.randn(H, generator=g) * 0.05
    ln(pre + "layer_norm")
    for n in range(cfg["num_hidden_layers"]):
        p = f"{pre}layers.{n}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "attention." + nm, H, H)
        ln(p + "layer_norm")
        lin(p + "feed_forward.intermediate_dense", I, H)
        lin(p + "feed_forward.output_dense", H, I)
        ln(p + "final_layer_norm")
    for n in range(cfg["num_hidden_layers"] // 2):
        lin(f"{pre}skip_connections_layers.{n}", H, 2 * H)
    lin(pre + "proj_out", F_, H, (F_, H, 1))
    return sd
