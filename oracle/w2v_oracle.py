"""Oracle: XLS-R (wav2vec 2.0, pre-LN "stable layer norm") unit extractor restated functionally on CPU
torch (TEST INFRASTRUCTURE).  **parity unpinned** against the reference's actual tokenizer:

The reference calls third-party `seamless_communication @ 90e2b57` `UnitExtractor("xlsr2_1b_v2", kmeans_10k)`
`.predict(wave, 34)` (src/inference.py:59,111-113; setup.py:49) on top of `fairseq2`; neither package nor any
checkpoint is in /root/reference or installable here, and the reference has no tests or fixtures at this
boundary.  This file restates the published algorithm [RECALLED, SURVEY.md §8 a1]:
  layer_norm over the whole waveform -> 7 x (Conv1d + LayerNorm(512) + GELU) -> LayerNorm(512) -> Linear
  512->1280 -> x + GELU(grouped weight-normed conv k128 g16, last frame trimmed) -> pre-LN encoder layers
  0..out_layer_idx (early exit, no final LayerNorm) -> k-means: argmin(|x|^2 - 2 x C + |c|^2).
What IS pinned: the network part is checked against the independent HF `transformers.Wav2Vec2Model`
implementation of the same architecture (tests/test_oracle_cpu.py), and the k-means stage against a
brute-force fp64 argmin.  State-dict keys are HF Wav2Vec2Model's.
"""
import math

import torch
import torch.nn.functional as F

XLSR_1B = dict(conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2),
               hidden_size=1280, num_attention_heads=16, intermediate_size=5120, num_hidden_layers=48,
               num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5, n_units=10000)


def n_frames(n, cfg):
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        n = (n - k) // s + 1
    return n


def posconv_weight(sd):
    p = "encoder.pos_conv_embed.conv."
    if p + "weight" in sd:
        return sd[p + "weight"]
    g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
    return g * v / v.norm(dim=(0, 1), keepdim=True)


def features(sd, cfg, wave, out_layer_idx):
    """wave f32 [n] -> hidden f32 [frames, hidden] = output of encoder layer `out_layer_idx` (0-based)."""
    eps = cfg["layer_norm_eps"]
    x = F.layer_norm(wave, wave.shape).view(1, 1, -1)
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
        p = f"feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=s)
        x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps).transpose(1, 2)
        x = F.gelu(x)
    x = x.transpose(1, 2)[0]                                                                   # [T, 512]
    x = F.layer_norm(x, (x.shape[-1],), sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], eps)
    x = F.linear(x, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    H, nh = cfg["hidden_size"], cfg["num_attention_heads"]
    hd = H // nh
    kw = cfg["num_conv_pos_embeddings"]
    pc = F.conv1d(x.T[None], posconv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"], padding=kw // 2,
                  groups=cfg["num_conv_pos_embedding_groups"])
    if kw % 2 == 0:
        pc = pc[:, :, :-1]
    x = x + F.gelu(pc)[0].T
    T = x.shape[0]
    for n in range(out_layer_idx + 1):
        p = f"encoder.layers.{n}."
        h = F.layer_norm(x, (H,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps)
        q = F.linear(h, sd[p + "attention.q_proj.weight"], sd[p + "attention.q_proj.bias"]) * hd ** -0.5
        k = F.linear(h, sd[p + "attention.k_proj.weight"], sd[p + "attention.k_proj.bias"])
        v = F.linear(h, sd[p + "attention.v_proj.weight"], sd[p + "attention.v_proj.bias"])
        sh = lambda z: z.view(T, nh, hd).transpose(0, 1)
        w = torch.softmax(sh(q) @ sh(k).transpose(1, 2), dim=-1)
        o = (w @ sh(v)).transpose(0, 1).reshape(T, H)
        x = x + F.linear(o, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        h = F.layer_norm(x, (H,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
        f = F.linear(F.gelu(F.linear(h, sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"])),
                     sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"])
        x = x + f
    return x


def kmeans_assign(x, centroids):
    """x [T, D], centroids [n_units, D] (the .npy layout) -> (ids int64 [T], dist [T, n_units])."""
    C = centroids.T                                            # [D, n_units]
    cn = (C ** 2).sum(0, keepdim=True)
    dist = x.pow(2).sum(1, keepdim=True) - 2 * torch.matmul(x, C) + cn
    return dist.argmin(dim=-1), dist


def predict(sd, cfg, centroids, wave, out_layer_idx):
    return kmeans_assign(features(sd, cfg, wave, out_layer_idx), centroids)[0]


def random_state_dict(cfg, seed=0, n_layers=None):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {}
    cin = 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"] = r(c, cin, k, sc=(cin * k) ** -0.5)
        sd[p + "conv.bias"] = r(c, sc=0.05)
        sd[p + "layer_norm.weight"] = 1 + 0.1 * r(c)
        sd[p + "layer_norm.bias"] = 0.1 * r(c)
        cin = c
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    sd["feature_projection.layer_norm.weight"] = 1 + 0.1 * r(cin)
    sd["feature_projection.layer_norm.bias"] = 0.1 * r(cin)
    sd["feature_projection.projection.weight"] = r(H, cin, sc=cin ** -0.5)
    sd["feature_projection.projection.bias"] = r(H, sc=0.05)
    kw, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = 1 + 0.2 * torch.rand(1, 1, kw, generator=g)
    sd["encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = r(H, H // G, kw)
    sd["encoder.pos_conv_embed.conv.bias"] = r(H, sc=0.05)
    for n in range(n_layers if n_layers is not None else cfg["num_hidden_layers"]):
        p = f"encoder.layers.{n}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{nm}.weight"] = r(H, H, sc=H ** -0.5)
            sd[p + f"attention.{nm}.bias"] = r(H, sc=0.05)
        for nm in ("layer_norm", "final_layer_norm"):
            sd[p + nm + ".weight"] = 1 + 0.1 * r(H)
            sd[p + nm + ".bias"] = 0.1 * r(H)
        sd[p + "feed_forward.intermediate_dense.weight"] = r(I, H, sc=H ** -0.5)
        sd[p + "feed_forward.intermediate_dense.bias"] = r(I, sc=0.05)
        sd[p + "feed_forward.output_dense.weight"] = r(H, I, sc=I ** -0.5)
        sd[p + "feed_forward.output_dense.bias"] = r(H, sc=0.05)
    return sd
