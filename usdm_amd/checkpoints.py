"""Local checkpoint plumbing for the drop-ins (there is no network: hub names resolve to directories on disk).

The reference loads four checkpoints by hub name into `--model_cache_dir` (src/inference.py:105-129,
src/decoder/voicebox/util/model_util.py:57-69).  A cache directory populated by the reference therefore has the
huggingface_hub layout  <cache>/models--<org>--<name>/snapshots/<rev>/...  ; hand-made caches use  <cache>/<name>/ .
resolve_local() accepts both, so `python -m usdm_amd.inference --model_cache_dir <the reference's cache>` works unchanged.

TensorSource streams tensors by name out of (sharded) safetensors or torch .bin files without materialising the whole
state dict on the host: the 7B is 14.6 GB in bf16 and goes to the GPU tensor by tensor (each rank of a tensor-parallel
job slices its shard on the way).
"""
import glob
import json
import os

import torch


def resolve_local(cache_dir, repo_id, must_contain=()):
    """Directory of hub repo `org/name` inside cache_dir, or raise FileNotFoundError listing what was tried."""
    org, _, name = repo_id.rpartition("/")
    cands = [os.path.join(cache_dir, name), os.path.join(cache_dir, repo_id) if org else None,
             os.path.join(cache_dir, repo_id.replace("/", "--"))]
    snap = os.path.join(cache_dir, "models--" + repo_id.replace("/", "--"), "snapshots")
    if os.path.isdir(snap):
        revs = sorted((os.path.join(snap, r) for r in os.listdir(snap)), key=os.path.getmtime, reverse=True)
        cands = revs + cands
    if os.path.isdir(cache_dir) and all(os.path.exists(os.path.join(cache_dir, f)) for f in must_contain) and must_contain:
        cands.append(cache_dir)                      # the directory itself is the checkpoint
    tried = []
    for c in cands:
        if not c:
            continue
        tried.append(c)
        if os.path.isdir(c) and all(os.path.exists(os.path.join(c, f)) for f in must_contain):
            return c
    raise FileNotFoundError(f"{repo_id}: no local copy under {cache_dir} (hub downloads are not available); tried " + ", ".join(tried))


class TensorSource:
    """name -> tensor (CPU), lazily, from a checkpoint directory or file:
       model.safetensors.index.json + shards | model.safetensors | pytorch_model.bin.index.json + shards | pytorch_model.bin | a file."""

    def __init__(self, path):
        self.path = path
        self._where, self._open, self._bins = {}, {}, {}
        files = []
        if os.path.isdir(path):
            for idx in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
                p = os.path.join(path, idx)
                if os.path.exists(p):
                    with open(p) as f:
                        for k, fn in json.load(f)["weight_map"].items():
                            self._where[k] = os.path.join(path, fn)
                    break
            else:
                files = [p for p in (os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")) if os.path.exists(p)][:1]
                if not files:
                    files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or sorted(glob.glob(os.path.join(path, "*.bin")))
                if not files:
                    raise FileNotFoundError(f"{path}: no safetensors / .bin weights found")
        else:
            files = [path]
        for fn in files:
            for k in self._keys_of(fn):
                self._where[k] = fn

    def _keys_of(self, fn):
        if fn.endswith(".safetensors"):
            return list(self._st(fn).keys())
        return list(self._bin(fn).keys())

    def _st(self, fn):
        if fn not in self._open:
            from safetensors import safe_open
            self._open[fn] = safe_open(fn, framework="pt", device="cpu")       # memory-mapped: get_tensor reads one tensor
        return self._open[fn]

    def _bin(self, fn):
        if fn not in self._bins:
            try:
                sd = torch.load(fn, map_location="cpu", mmap=True, weights_only=True)
            except Exception:  # noqa: BLE001 - legacy (non-zipfile) checkpoints cannot be mmapped
                sd = torch.load(fn, map_location="cpu", weights_only=True)
            for wrap in ("state_dict", "model", "generator"):
                if isinstance(sd, dict) and wrap in sd and isinstance(sd[wrap], dict) and len(sd) <= 4:
                    sd = sd[wrap]
            self._bins = {fn: sd}                                               # keep ONE .bin shard resident at a time
        return self._bins[fn]

    def keys(self):
        return self._where.keys()

    def __contains__(self, k):
        return k in self._where

    def __call__(self, name):
        fn = self._where.get(name)
        if fn is None:
            raise KeyError(f"{name} not found in {self.path}")
        if fn.endswith(".safetensors"):
            return self._st(fn).get_tensor(name)
        return self._bin(fn)[name]

    def state_dict(self):
        return {k: self(k) for k in self.keys()}


def read_mistral_config(path):
    """HF config.json -> the dict usdm_amd.llm.USDMForCausalLM takes."""
    with open(os.path.join(path, "config.json")) as f:
        c = json.load(f)
    if c.get("model_type", "mistral") != "mistral":
        raise ValueError(f"{path}: model_type {c.get('model_type')!r} is not the Mistral architecture of USDM")
    out = {k: c[k] for k in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                             "num_key_value_heads")}
    out["head_dim"] = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
    out["rms_norm_eps"] = c.get("rms_norm_eps", 1e-5)
    out["rope_theta"] = float(c.get("rope_theta", 10000.0))
    out["max_position_embeddings"] = c.get("max_position_embeddings", 32768)
    out["sliding_window"] = c.get("sliding_window", 4096)      # Mistral-7B-v0.1: 4096; an explicit null = full causal attention
    if c.get("tie_word_embeddings"):
        raise NotImplementedError("tied input/output embeddings are not the USDM checkpoint layout")
    return out


# fairseq2 wav2vec2 parameter names (seamless_communication's `xlsr2_1b_v2` card loads into fairseq2's Wav2Vec2Model) ->
# the HF Wav2Vec2Model names this package uses.  [RECALLED from the upstream sources, which are not in this environment:
# unverifiable here, see DESIGN.md section 2 "parity unpinned".]
_F2_RULES = [
    ("encoder_frontend.feature_extractor.layers.", "feature_extractor.conv_layers."),
    ("encoder_frontend.post_extract_layer_norm.", "feature_projection.layer_norm."),
    ("encoder_frontend.model_dim_proj.", "feature_projection.projection."),
    ("encoder_frontend.pos_encoder.conv.", "encoder.pos_conv_embed.conv."),
    (".self_attn_layer_norm.", ".layer_norm."),
    (".self_attn.output_proj.", ".attention.out_proj."),
    (".self_attn.", ".attention."),
    (".ffn_layer_norm.", ".final_layer_norm."),
    (".ffn.inner_proj.", ".feed_forward.intermediate_dense."),
    (".ffn.output_proj.", ".feed_forward.output_dense."),
]


def convert_w2v_keys(sd):
    """Accept HF Wav2Vec2Model names as they are (optionally under a `wav2vec2.` prefix); map fairseq2 names onto them.
    Weight-norm of the positional conv may come as weight_g / weight_v or parametrizations.weight.original0 / 1."""
    out = {}
    for k, v in sd.items():
        if k.startswith("wav2vec2."):
            k = k[len("wav2vec2."):]
        for a, b in _F2_RULES:
            if a in k:
                k = k.replace(a, b)
        k = k.replace("pos_conv_embed.conv.weight_g", "pos_conv_embed.conv.parametrizations.weight.original0")
        k = k.replace("pos_conv_embed.conv.weight_v", "pos_conv_embed.conv.parametrizations.weight.original1")
        out[k] = v
    return out
