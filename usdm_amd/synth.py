"""Random-init builders for the four models at their real sizes, generated ON the GPU.

There is no network: checkpoints (naver-ai/USDM-DailyTalk, naver-ai/xlsr-token-Voicebox,
nvidia/bigvgan_22khz_80band, xlsr2_1b_v2 + kmeans_10k) cannot be fetched, so benchmarks and the smoke
test use random weights of the exact architectures (BASELINE.json: "random-init 7B weights")."""
import torch

from .llm import MISTRAL_7B_USDM, USDMForCausalLM
from .unit_extractor import XLSR_1B, UnitExtractor
from .voicebox.model import Voicebox
from .voicebox.vocoder.env import AttrDict
from .voicebox.vocoder.models import BigVGAN

VOICEBOX_CFG = dict(n_feats=80, n_tokens=10000, embedding_dim=1280, hidden_size=1024, intermediate_size=4096,
                    num_attention_heads=16, num_hidden_layers=24, convpos_width=31, convpos_groups=16, convpos_depth=2,
                    attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=1e-4)

BIGVGAN_22K_80 = dict(
    resblock="1", upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
    upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta", snake_logscale=True,
    num_mels=80, sampling_rate=22050, hop_size=256, n_fft=1024, win_size=1024, fmin=0, fmax=8000)


def _gen(dev, seed):
    return torch.Generator(device=dev).manual_seed(seed)


def w2v_state_dict(dev, cfg=None, n_layers=35, seed=1):
    """Random XLS-R weights (HF Wav2Vec2Model key names), fp32, generated on the device."""
    cfg = dict(cfg or XLSR_1B)
    g = _gen(dev, seed)
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev, generator=g) * sc
    sd, cin = {}, 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"], sd[p + "conv.bias"] = r(c, cin, k, sc=(cin * k) ** -0.5), r(c, sc=0.05)
        sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"] = 1 + 0.1 * r(c), 0.1 * r(c)
        cin = c
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"] = 1 + 0.1 * r(cin), 0.1 * r(cin)
    sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"] = r(H, cin, sc=cin ** -0.5), r(H, sc=0.05)
    kw, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    sd["encoder.pos_conv_embed.conv.weight"] = r(H, H // G, kw, sc=(kw * H // G) ** -0.5)
    sd["encoder.pos_conv_embed.conv.bias"] = r(H, sc=0.05)
    for n in range(n_layers):
        p = f"encoder.layers.{n}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{nm}.weight"], sd[p + f"attention.{nm}.bias"] = r(H, H, sc=H ** -0.5), r(H, sc=0.05)
        for nm in ("layer_norm", "final_layer_norm"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1 + 0.1 * r(H), 0.1 * r(H)
        sd[p + "feed_forward.intermediate_dense.weight"], sd[p + "feed_forward.intermediate_dense.bias"] = r(I, H, sc=H ** -0.5), r(I, sc=0.05)
        sd[p + "feed_forward.output_dense.weight"], sd[p + "feed_forward.output_dense.bias"] = r(H, I, sc=I ** -0.5), r(H, sc=0.05)
    return sd


def w2v_centroids(dev, cfg=None, cseed=2):
    cfg = dict(cfg or XLSR_1B)
    return torch.randn(cfg["n_units"], cfg["hidden_size"], device=dev, generator=_gen(dev, cseed))


def make_unit_extractor(dev, cfg=None, n_layers=35, seed=1, cseed=2):
    """XLS-R 1B up to encoder layer 34 + 10 000 centroids, fp32, random."""
    cfg = dict(cfg or XLSR_1B)
    return UnitExtractor(None, None, device=dev, config=cfg, state_dict=w2v_state_dict(dev, cfg, n_layers, seed),
                         centroids=w2v_centroids(dev, cfg, cseed))


def random_llm_state_dict(cfg, dev, seed=3, norm_jitter=0.1):
    """Full HF-named Mistral state dict, bf16, generated ON the device in a fixed key order (tests hand a .cpu() copy of the
    SAME tensors to the CPU oracle: generating 7.3 B normals on the host would take minutes)."""
    c, d = cfg, cfg["head_dim"]
    H, I, V = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
    nh, nkv = c["num_attention_heads"], c["num_key_value_heads"]
    g = _gen(dev, seed)
    bf = torch.bfloat16
    r = lambda o, i, sc: (torch.randn(o, i, device=dev, generator=g) * sc).to(bf)
    n = lambda: (1 + norm_jitter * torch.randn(H, device=dev, generator=g)).to(bf)
    sd = {"model.embed_tokens.weight": r(V, H, 1.0), "lm_head.weight": r(V, H, H ** -0.5), "model.norm.weight": n()}
    for l in range(c["num_hidden_layers"]):
        p = f"model.layers.{l}."
        sd[p + "self_attn.q_proj.weight"] = r(nh * d, H, H ** -0.5)
        sd[p + "self_attn.k_proj.weight"] = r(nkv * d, H, H ** -0.5)
        sd[p + "self_attn.v_proj.weight"] = r(nkv * d, H, H ** -0.5)
        sd[p + "self_attn.o_proj.weight"] = r(H, nh * d, (nh * d) ** -0.5)
        sd[p + "mlp.gate_proj.weight"] = r(I, H, H ** -0.5)
        sd[p + "mlp.up_proj.weight"] = r(I, H, H ** -0.5)
        sd[p + "mlp.down_proj.weight"] = r(H, I, I ** -0.5)
        sd[p + "input_layernorm.weight"] = n()
        sd[p + "post_attention_layernorm.weight"] = n()
    return sd


def make_llm(dev, cfg=None, seed=3, **kw):
    return USDMForCausalLM.random_init(dict(cfg or MISTRAL_7B_USDM), dev, seed=seed, **kw)


def make_voicebox(dev, cfg=None, seed=4):
    torch.manual_seed(seed)
    with torch.device(dev):
        m = Voicebox(**(cfg or VOICEBOX_CFG))
    with torch.no_grad():
        m.estimator.embed.weight.mul_(0.05)
    return m.eval()


def make_bigvgan(dev, h=None, seed=5, compute_dtype=torch.float32):
    torch.manual_seed(seed)
    with torch.device(dev):
        m = BigVGAN(AttrDict(h or BIGVGAN_22K_80), compute_dtype=compute_dtype)
    m.remove_weight_norm()
    with torch.no_grad():   # keep activations O(1) so the sin^2 path does real work (default init is std 0.01)
        for p_ in m.parameters():
            if p_.dim() == 3:
                fan = p_.shape[1] * p_.shape[2]
                p_.normal_(0.0, fan ** -0.5)
    m.invalidate()
    return m.eval()
