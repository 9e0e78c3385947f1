"""CPU: the local-checkpoint boundary: hub names resolving inside a cache directory, streaming tensor source
over sharded safetensors, Voicebox save_pretrained -> from_pretrained with the weight-norm parametrisation keys, BigVGAN
from_pretrained for checkpoints with and without weight norm, Mistral config reading, wav2vec2 key conversion.
(Reference: src/inference.py:105-129, src/decoder/voicebox/util/model_util.py:57-69, vocoder/models.py:234-313.)"""
import json
import os

import pytest
import torch


def test_resolve_local_accepts_hub_cache_layout_and_plain_directories(tmp_path):
    from usdm_amd.checkpoints import resolve_local
    cache = str(tmp_path)
    snap = os.path.join(cache, "models--naver-ai--xlsr-token-Voicebox", "snapshots", "abc123")
    os.makedirs(snap)
    open(os.path.join(snap, "config.json"), "w").write("{}")
    assert resolve_local(cache, "naver-ai/xlsr-token-Voicebox", ("config.json",)) == snap
    plain = os.path.join(cache, "bigvgan_22khz_80band")
    os.makedirs(plain)
    open(os.path.join(plain, "config.json"), "w").write("{}")
    assert resolve_local(cache, "nvidia/bigvgan_22khz_80band", ("config.json",)) == plain
    with pytest.raises(FileNotFoundError) as e:
        resolve_local(cache, "naver-ai/USDM-DailyTalk", ("config.json",))
    assert "USDM-DailyTalk" in str(e.value)


def test_tensor_source_streams_sharded_safetensors_and_bin(tmp_path):
    from safetensors.torch import save_file
    from usdm_amd.checkpoints import TensorSource
    g = torch.Generator().manual_seed(0)
    sd = {f"model.layers.{i}.w": torch.randn(4, 6, generator=g).to(torch.bfloat16) for i in range(5)}
    d = str(tmp_path / "sharded")
    os.makedirs(d)
    names = sorted(sd)
    save_file({k: sd[k] for k in names[:2]}, os.path.join(d, "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in names[2:]}, os.path.join(d, "model-00002-of-00002.safetensors"))
    wm = {k: ("model-00001-of-00002.safetensors" if k in names[:2] else "model-00002-of-00002.safetensors") for k in names}
    json.dump({"metadata": {}, "weight_map": wm}, open(os.path.join(d, "model.safetensors.index.json"), "w"))
    src = TensorSource(d)
    assert sorted(src.keys()) == names and all(torch.equal(src(k), sd[k]) for k in names)
    with pytest.raises(KeyError):
        src("nope")
    d2 = str(tmp_path / "single")
    os.makedirs(d2)
    torch.save(sd, os.path.join(d2, "pytorch_model.bin"))
    src2 = TensorSource(d2)
    assert all(torch.equal(src2(k), sd[k]) for k in names)
    f3 = str(tmp_path / "wrapped.pt")
    torch.save({"generator": sd}, f3)                     # BigVGAN-style wrapper
    assert all(torch.equal(TensorSource(f3)(k), sd[k]) for k in names)


def test_voicebox_save_pretrained_from_pretrained_roundtrip_keeps_weight_norm_keys(tmp_path):
    from oracle import voicebox_oracle as VO
    from tests.golden.configs import SMALL_VB
    from usdm_amd.voicebox.model import Voicebox
    kw = {k: SMALL_VB[k] for k in SMALL_VB if k != "sigma_min"}
    m = Voicebox(**kw, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=1e-4)
    sd = VO.random_state_dict(SMALL_VB, 3)
    m.load_state_dict(sd, strict=True)
    d = str(tmp_path / "models--naver-ai--xlsr-token-Voicebox" / "snapshots" / "r1")
    m.save_pretrained(d)
    assert json.load(open(os.path.join(d, "config.json")))["n_tokens"] == SMALL_VB["n_tokens"]
    m2 = Voicebox.from_pretrained("naver-ai/xlsr-token-Voicebox", cache_dir=str(tmp_path))     # the reference's call shape
    sd2 = m2.state_dict()
    assert set(sd2) == set(sd)
    assert "estimator.pos_conv_embeds.0.conv.parametrizations.weight.original0" in sd2
    assert all(torch.equal(sd2[k], sd[k]) for k in sd)
    assert m2.n_tokens == m.n_tokens and m2.sigma_min == m.sigma_min


def test_bigvgan_from_pretrained_with_and_without_weight_norm(tmp_path, capsys):
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64)
    # (a) a checkpoint WITH weight norm (weight_g / weight_v), as nvidia/bigvgan_22khz_80band ships
    m = BigVGAN(AttrDict(h))
    da = str(tmp_path / "with_wn")
    os.makedirs(da)
    json.dump(h, open(os.path.join(da, "config.json"), "w"))
    torch.save({"generator": m.state_dict()}, os.path.join(da, "bigvgan_generator.pt"))
    assert any(k.endswith("weight_g") for k in m.state_dict())
    ma = BigVGAN.from_pretrained(da)
    assert all(torch.equal(v, ma.state_dict()[k]) for k, v in m.state_dict().items())
    ma.remove_weight_norm()                                   # what initialize_decoder does next
    # (b) a checkpoint whose weight norm was already removed: the try/except of models.py:270-275
    m.remove_weight_norm()
    stripped = m.state_dict()
    assert not any(k.endswith("weight_g") for k in stripped)
    db = str(tmp_path / "stripped")
    os.makedirs(db)
    json.dump(h, open(os.path.join(db, "config.json"), "w"))
    torch.save({"generator": stripped}, os.path.join(db, "bigvgan_generator.pt"))
    mb = BigVGAN.from_pretrained(db)
    assert "does not contain weight norm" in capsys.readouterr().out
    assert all(torch.equal(v, mb.state_dict()[k]) for k, v in stripped.items())
    # folded weights of (a) after removal == stripped weights of (b)
    assert all(torch.allclose(ma.state_dict()[k], stripped[k], atol=1e-6) for k in stripped)
    assert mb.h.sampling_rate == 22050 and mb.h.hop_size == 256


def test_read_mistral_config_and_w2v_key_conversion(tmp_path):
    from usdm_amd.checkpoints import convert_w2v_keys, read_mistral_config
    cfg = dict(model_type="mistral", vocab_size=42003, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
               num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768,
               sliding_window=4096, torch_dtype="bfloat16")
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    c = read_mistral_config(str(tmp_path))
    from usdm_amd.llm import MISTRAL_7B_USDM
    assert all(c[k] == v for k, v in MISTRAL_7B_USDM.items()) and c["sliding_window"] == 4096
    t = torch.zeros(1)
    f2 = {"encoder_frontend.feature_extractor.layers.3.conv.weight": t, "encoder_frontend.post_extract_layer_norm.bias": t,
          "encoder_frontend.model_dim_proj.weight": t, "encoder_frontend.pos_encoder.conv.weight_g": t,
          "encoder.layers.7.self_attn.q_proj.weight": t, "encoder.layers.7.self_attn.output_proj.bias": t,
          "encoder.layers.7.self_attn_layer_norm.weight": t, "encoder.layers.7.ffn.inner_proj.weight": t,
          "encoder.layers.7.ffn.output_proj.weight": t, "encoder.layers.7.ffn_layer_norm.bias": t}
    got = set(convert_w2v_keys(f2))
    assert got == {"feature_extractor.conv_layers.3.conv.weight", "feature_projection.layer_norm.bias", "feature_projection.projection.weight",
                   "encoder.pos_conv_embed.conv.parametrizations.weight.original0", "encoder.layers.7.attention.q_proj.weight",
                   "encoder.layers.7.attention.out_proj.bias", "encoder.layers.7.layer_norm.weight",
                   "encoder.layers.7.feed_forward.intermediate_dense.weight", "encoder.layers.7.feed_forward.output_dense.weight",
                   "encoder.layers.7.final_layer_norm.bias"}
    hf = {"wav2vec2.encoder.layers.0.attention.k_proj.weight": t, "feature_extractor.conv_layers.0.layer_norm.weight": t}
    assert set(convert_w2v_keys(hf)) == {"encoder.layers.0.attention.k_proj.weight", "feature_extractor.conv_layers.0.layer_norm.weight"}
