"""GPU: real checkpoint directories end to end.

  * USDMForCausalLM.from_pretrained on a directory WRITTEN BY HF transformers (MistralForCausalLM.save_pretrained, sharded
    safetensors + index): tokens equal to the CPU oracle on the same weights;
  * UnitExtractor on a directory written by HF Wav2Vec2Model.save_pretrained + kmeans .npy: ids equal to k-means over HF's own
    hidden_states[35];
  * initialize_decoder on a cache in the huggingface_hub layout;
  * `python -m usdm_amd.inference` (main()) on a synthetic --model_cache_dir holding all four checkpoints + a tokenizer:
    the reference CLI, wav in -> wav out."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LLM_CFG = dict(vocab_size=42003, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)


def _write_llm(d, seed=5, shard="3MB", rig_eos=False):
    import transformers
    from oracle import mistral_oracle as MO
    sd = MO.random_state_dict(LLM_CFG, seed=seed)
    if rig_eos:
        # A random model practically never emits an EOS, so every round would run into max_length (a TOTAL length in the
        # reference, inference.py:64-66) and leave no room for the next round.  Rig the output head so the rounds end after a few
        # tokens: only a handful of text ids keep a (random) logit, and the unit round's EOS 28705 gets a 2x larger one.
        lm = sd["lm_head.weight"].float()
        keep = torch.tensor([3 + ord(c) - 32 for c in "aeot "] + [98, 28705])     # characters (ids of _write_tokenizer), "\n" = 98
        mask = torch.zeros(32000, dtype=torch.bool)
        mask[keep] = True
        lm[:32000][~mask] = 0
        lm[28705] *= 2.0
        sd["lm_head.weight"] = lm.to(torch.bfloat16)
    hf_cfg = transformers.MistralConfig(**{k: LLM_CFG[k] for k in LLM_CFG}, sliding_window=4096, tie_word_embeddings=False,
                                        attn_implementation="eager")
    m = transformers.MistralForCausalLM(hf_cfg).to(torch.bfloat16)
    m.load_state_dict(sd, strict=True)
    m.save_pretrained(d, max_shard_size=shard, safe_serialization=True)
    return sd


def test_llm_from_pretrained_streams_hf_sharded_safetensors(dev, tmp_path):
    from oracle import mistral_oracle as MO
    from usdm_amd.llm import USDMForCausalLM
    d = str(tmp_path / "models--naver-ai--USDM-DailyTalk" / "snapshots" / "rev0")
    sd = _write_llm(d)
    assert os.path.exists(os.path.join(d, "model.safetensors.index.json")), os.listdir(d)       # really sharded
    m = USDMForCausalLM.from_pretrained("naver-ai/USDM-DailyTalk", device=dev, cache_dir=str(tmp_path), torch_dtype=torch.bfloat16,
                                        attn_implementation="flash_attention_2", device_map="auto", low_cpu_mem_usage=True,
                                        ctx_max=128).to(dev).eval()
    ids = torch.randint(0, 32000, (21,), generator=torch.Generator().manual_seed(1))
    bad = [[i] for i in range(32000, 42003)]
    ref = MO.greedy_generate(sd, LLM_CFG, ids, 12, bad_words_ids=bad)
    out = m.generate(input_ids=ids[None].to(dev), max_new_tokens=12, do_sample=True, top_k=1, bad_words_ids=bad)[0].tolist()
    assert out[:21 + 4] == ref[:21 + 4]                       # (later tokens may differ at bf16 near-ties; covered elsewhere)
    big = USDMForCausalLM.from_pretrained(d, device=dev, ctx_max=8192)   # beyond the 4096 sliding window: the kernels bound the key range
    assert big.window == 4096 and big.ctx_max == 8192 and m.window == 0


def _write_w2v(d, n_layers=4):
    import transformers
    cfg = transformers.Wav2Vec2Config(hidden_size=256, num_hidden_layers=n_layers, num_attention_heads=4, intermediate_size=512,
                                      feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True,
                                      num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=4, layer_norm_eps=1e-5,
                                      hidden_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
                                      mask_time_prob=0.0, apply_spec_augment=False)
    torch.manual_seed(3)
    m = transformers.Wav2Vec2Model(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    m.save_pretrained(d, safe_serialization=True)
    return m


def test_unit_extractor_from_hf_wav2vec2_directory_and_kmeans_npy(dev, tmp_path):
    from usdm_amd.unit_extractor import UnitExtractor
    d = str(tmp_path / "xlsr2_1b_v2")
    hf = _write_w2v(d, n_layers=5)          # 5 layers: hidden_states[4] is then NOT the last entry (HF applies the final LayerNorm to that one)
    cen = torch.randn(300, 256, generator=torch.Generator().manual_seed(4)) * 0.5
    np.save(str(tmp_path / "kmeans_10k.npy"), cen.numpy())
    wave = torch.randn(12000, generator=torch.Generator().manual_seed(5)) * 0.1
    # the reference's call, resolved inside the cache directory (src/inference.py:111-113)
    ue = UnitExtractor("xlsr2_1b_v2", "https://dl.fbaipublicfiles.com/seamlessM4T/models/unit_extraction/kmeans_10k.npy", device=dev,
                       cache_dir=str(tmp_path))
    got = ue.predict(wave.to(dev), 3).cpu()
    with torch.no_grad():
        x = torch.nn.functional.layer_norm(wave, wave.shape)           # upstream normalises the whole waveform first
        hs = hf(x[None], output_hidden_states=True).hidden_states[4][0]  # [index 4] = output of encoder layer index 3
    dist = (hs ** 2).sum(1, keepdim=True) - 2 * hs @ cen.T + (cen.T ** 2).sum(0)
    want = dist.argmin(-1)
    top2 = torch.topk(dist, 2, largest=False).values
    bad = got != want
    print("ids equal to HF Wav2Vec2Model + k-means:", float((~bad).float().mean()))
    assert got.shape == want.shape and float((~bad).float().mean()) >= 0.97
    assert bool(((top2[:, 1] - top2[:, 0])[bad] <= 1e-3 * dist.abs().max()).all())


def _write_decoders(cache):
    from oracle import bigvgan_oracle as BO, voicebox_oracle as VO
    from tests.golden.configs import SMALL_VB
    from usdm_amd.voicebox.model import Voicebox
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    vcfg = dict(SMALL_VB, n_tokens=10000)
    kw = {k: vcfg[k] for k in vcfg if k != "sigma_min"}
    vb = Voicebox(**kw, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=1e-4)
    vb.load_state_dict(VO.random_state_dict(vcfg, 4))
    vb.save_pretrained(os.path.join(cache, "models--naver-ai--xlsr-token-Voicebox", "snapshots", "r0"))
    h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64)
    dv = os.path.join(cache, "models--nvidia--bigvgan_22khz_80band", "snapshots", "r0")
    os.makedirs(dv)
    json.dump(h, open(os.path.join(dv, "config.json"), "w"))
    voc = BigVGAN(AttrDict(h))                                          # WITH weight norm, like the published checkpoint
    torch.save({"generator": voc.state_dict()}, os.path.join(dv, "bigvgan_generator.pt"))
    return vcfg, h


def test_initialize_decoder_from_hub_cache_layout(dev, tmp_path):
    from usdm_amd.voicebox.util.model_util import initialize_decoder, reconstruct_speech
    _write_decoders(str(tmp_path))
    vb, voc = initialize_decoder(str(tmp_path), dev)
    assert not vb.training and not voc.training
    assert not any(k.endswith("weight_g") for k in voc.state_dict())     # weight norm removed (model_util.py:68)
    audio = reconstruct_speech(torch.randint(0, 10000, (12,)).to(dev), dev, None, None, vb, voc, n_timesteps=2)
    assert audio.dtype == np.float32 and audio.shape == (256 * ((12 * 441) // 256),) and np.isfinite(audio).all()


def _write_tokenizer(d):
    """A fast tokenizer with USDM's id layout (src/train_pt.py:104-128): characters, then <|continue|> 32000, <|correspond|> 32001,
    <|unit i|> 32002+i, <pad> 42002."""
    from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for c in [chr(i) for i in range(32, 127)] + ["\n"]:
        vocab[c] = len(vocab)
    while len(vocab) < 32000:
        vocab[f"<filler{len(vocab)}>"] = len(vocab)
    vocab["<|continue|>"], vocab["<|correspond|>"] = 32000, 32001
    for i in range(10000):
        vocab[f"<|unit{i}|>"] = 32002 + i
    vocab["<pad>"] = 42002
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(r"<\|unit\d+\|>|<\|correspond\|>|<\|continue\|>|[\s\S]"), behavior="isolated")
    tok.decoder = decoders.Fuse()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", pad_token="<pad>", model_max_length=1024)
    fast.save_pretrained(d)


def test_cli_main_on_a_synthetic_model_cache_dir(dev, tmp_path):
    """`python -m usdm_amd.inference --input_path u.wav --reference_path r.wav --model_cache_dir C --output_path o.wav`
    (src/inference.py:92-134) with every checkpoint loaded from disk."""
    from scipy.io.wavfile import read, write
    import usdm_amd.inference as inf
    cache = str(tmp_path / "cache")
    os.makedirs(cache)
    _write_decoders(cache)
    _write_w2v(os.path.join(cache, "xlsr2_1b_v2"), n_layers=35)
    np.save(os.path.join(cache, "kmeans_10k.npy"), (torch.randn(10000, 256, generator=torch.Generator().manual_seed(6)) * 0.5).numpy())
    llm_dir = os.path.join(cache, "models--naver-ai--USDM-DailyTalk", "snapshots", "r0")
    _write_llm(llm_dir, seed=7, shard="30MB", rig_eos=True)
    _write_tokenizer(llm_dir)
    t = torch.arange(20000) / 16000.0
    wav = (0.2 * torch.sin(2 * torch.pi * 300 * t) + 0.02 * torch.randn(20000, generator=torch.Generator().manual_seed(8))).numpy().astype(np.float32)
    user, ref, out = (str(tmp_path / n) for n in ("user.wav", "ref.wav", "out.wav"))
    write(user, 16000, wav)
    write(ref, 22050, wav[:18000])
    os.environ.pop("USDM_MODEL_CACHE_DIR", None)
    rc = inf.main(["--input_path", user, "--reference_path", ref, "--model_cache_dir", cache, "--output_path", out])
    assert rc == 0
    sr, data = read(out)
    assert sr == 22050 and data.dtype == np.float32 and data.ndim == 1 and data.size % 256 == 0 and data.size > 0
    assert np.isfinite(data).all() and np.abs(data).max() <= 1.0
    # and without a reference (speaker-unconditional generation)
    out2 = str(tmp_path / "out2.wav")
    assert inf.main(["--input_path", user, "--model_cache_dir", cache, "--output_path", out2]) == 0
    assert read(out2)[0] == 22050
