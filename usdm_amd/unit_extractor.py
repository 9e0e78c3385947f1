"""XLS-R + k-means speech tokenizer on MI355X: drop-in for
`seamless_communication.models.unit_extractor.UnitExtractor` as the reference uses it
(src/inference.py:59,111-113; src/decoder/voicebox/util/model_util.py:79-80):

    UnitExtractor(model, kmeans, device=...).predict(wave: FloatTensor[n] on device, out_layer_idx) -> LongTensor[frames]

The upstream package (seamless_communication @ 90e2b57 on fairseq2) and its checkpoints are not available
offline, so `model` / `kmeans` are local paths (a torch state dict with HF Wav2Vec2Model key names and a
[n_units, dim] .npy), or tensors passed directly.  All arithmetic is fp32 on the exact-f32 MFMA path
(usdm_gemm F32) so unit ids can match an fp32 reference bit for bit wherever the top-2 centroid margin
exceeds fp32 summation noise; `last_margin` reports that margin per frame.
"""
import os

import numpy as np
import torch

from . import ops
from ._lib import ACT_GELU
from .graph import GraphedPlan
from .plancache import LRU, Arena, bucket

SAMPLE_BUCKET = 16000   # 1 s of 16 kHz audio: lengths inside one bucket share ONE workspace
MAX_ARENAS = 3          # workspaces kept (least recently used goes first)
MAX_PLANS = 4           # exact-length plans (+ hipGraphs) kept per workspace

XLSR_1B = dict(conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2),
               hidden_size=1280, num_attention_heads=16, intermediate_size=5120, num_hidden_layers=48,
               num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, layer_norm_eps=1e-5, n_units=10000)


def _frames(n, cfg):
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        n = (n - k) // s + 1
    return n


class UnitExtractor:
    def __init__(self, model_name_or_card, kmeans_uri, device=None, dtype=torch.float32, config=None,
                 state_dict=None, centroids=None, cache_dir=None):
        if dtype != torch.float32:
            raise NotImplementedError("the tokenizer path is fp32 (bit-exact ids are the contract); got %s" % dtype)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("UnitExtractor (usdm_amd) runs on the MI355X only; there is no CPU fallback")
        self.cfg = dict(config or XLSR_1B)
        # The reference passes a model CARD name and a URL (src/inference.py:111-113).  Offline they resolve to files of the same
        # name inside cache_dir (or $USDM_MODEL_CACHE_DIR): <dir>/xlsr2_1b_v2[.pt|.safetensors|/] and <dir>/kmeans_10k.npy.
        cache_dir = cache_dir or os.environ.get("USDM_MODEL_CACHE_DIR")
        if cache_dir and state_dict is None and isinstance(model_name_or_card, str) and not os.path.exists(model_name_or_card):
            for c in (model_name_or_card, model_name_or_card + ".pt", model_name_or_card + ".safetensors", model_name_or_card + ".bin"):
                if os.path.exists(os.path.join(cache_dir, c)):
                    model_name_or_card = os.path.join(cache_dir, c)
                    break
        if cache_dir and centroids is None and isinstance(kmeans_uri, str) and not os.path.exists(kmeans_uri):
            c = os.path.join(cache_dir, os.path.basename(kmeans_uri))
            if os.path.exists(c):
                kmeans_uri = c
        if state_dict is None:
            if not (isinstance(model_name_or_card, str) and os.path.exists(model_name_or_card)):
                raise FileNotFoundError(f"{model_name_or_card}: model cards / URLs cannot be fetched offline; pass a local "
                                        "checkpoint (file or directory; HF Wav2Vec2Model or fairseq2 key names) or state_dict=")
            from .checkpoints import TensorSource, convert_w2v_keys
            state_dict = convert_w2v_keys(TensorSource(model_name_or_card).state_dict())
            cj = os.path.join(model_name_or_card, "config.json") if os.path.isdir(model_name_or_card) else None
            if config is None and cj and os.path.exists(cj):     # an HF Wav2Vec2 directory carries its own sizes
                import json
                with open(cj) as f:
                    hc = json.load(f)
                self.cfg.update({k: (tuple(hc[k]) if isinstance(hc[k], list) else hc[k]) for k in XLSR_1B if k in hc})
        if centroids is None:
            if not (isinstance(kmeans_uri, str) and os.path.exists(kmeans_uri)):
                raise FileNotFoundError(f"{kmeans_uri}: URLs cannot be fetched offline; pass a local .npy or centroids=")
            centroids = torch.from_numpy(np.load(kmeans_uri))
        from .checkpoints import convert_w2v_keys
        self._pack(convert_w2v_keys(state_dict), centroids)
        self._plans = LRU(MAX_ARENAS)     # (sample bucket, layer) -> (Arena, LRU of exact-length plans)
        self.last_io = None
        self.last_margin = None

    # ------------------------------------------------------------------ load-time packing
    def _pack(self, sd, centroids):
        cfg, dev = self.cfg, self.device
        g = lambda n: sd[n].detach().to(dev, torch.float32)
        P = {"convs": []}
        w0 = g("feature_extractor.conv_layers.0.conv.weight")  # [512,1,10]
        P["conv0"] = dict(w=w0.reshape(w0.shape[0], -1).contiguous(), b=g("feature_extractor.conv_layers.0.conv.bias").contiguous(),
                          g=g("feature_extractor.conv_layers.0.layer_norm.weight").contiguous(),
                          be=g("feature_extractor.conv_layers.0.layer_norm.bias").contiguous())
        for i in range(1, len(cfg["conv_kernel"])):
            p = f"feature_extractor.conv_layers.{i}."
            w = g(p + "conv.weight")  # [Cout, Cin, k] -> [Cout][k*Cin]
            P["convs"].append(dict(w=w.permute(0, 2, 1).contiguous().reshape(w.shape[0], -1), b=g(p + "conv.bias").contiguous(),
                                   g=g(p + "layer_norm.weight").contiguous(), be=g(p + "layer_norm.bias").contiguous(),
                                   k=cfg["conv_kernel"][i], s=cfg["conv_stride"][i]))
        P["fp_ln"] = (g("feature_projection.layer_norm.weight").contiguous(), g("feature_projection.layer_norm.bias").contiguous())
        P["fp_w"], P["fp_b"] = g("feature_projection.projection.weight").contiguous(), g("feature_projection.projection.bias").contiguous()
        H, G, kw = cfg["hidden_size"], cfg["num_conv_pos_embedding_groups"], cfg["num_conv_pos_embeddings"]
        cg = H // G
        pc = "encoder.pos_conv_embed.conv."
        if pc + "weight" in sd:
            w = g(pc + "weight")
        else:
            g0, v = g(pc + "parametrizations.weight.original0"), g(pc + "parametrizations.weight.original1")
            w = g0 * v / v.norm(dim=(0, 1), keepdim=True)
        P["pos_w"] = w.reshape(G, cg, cg, kw).permute(0, 1, 3, 2).contiguous().reshape(G, cg, kw * cg)
        P["pos_b"] = g(pc + "bias").contiguous()
        P["layers"] = []
        n = 0
        while f"encoder.layers.{n}.attention.q_proj.weight" in sd:
            p = f"encoder.layers.{n}."
            P["layers"].append(dict(
                ln1=(g(p + "layer_norm.weight").contiguous(), g(p + "layer_norm.bias").contiguous()),
                wqk=torch.cat([g(p + "attention.q_proj.weight"), g(p + "attention.k_proj.weight")], 0).contiguous(),
                bqk=torch.cat([g(p + "attention.q_proj.bias"), g(p + "attention.k_proj.bias")], 0).contiguous(),
                wv=g(p + "attention.v_proj.weight").contiguous(), bv=g(p + "attention.v_proj.bias").contiguous(),
                wo=g(p + "attention.out_proj.weight").contiguous(), bo=g(p + "attention.out_proj.bias").contiguous(),
                ln2=(g(p + "final_layer_norm.weight").contiguous(), g(p + "final_layer_norm.bias").contiguous()),
                w1=g(p + "feed_forward.intermediate_dense.weight").contiguous(), b1=g(p + "feed_forward.intermediate_dense.bias").contiguous(),
                w2=g(p + "feed_forward.output_dense.weight").contiguous(), b2=g(p + "feed_forward.output_dense.bias").contiguous()))
            n += 1
        C = centroids.detach().to(dev, torch.float32).contiguous()   # [n_units, D]
        P["C"] = C
        P["csq"] = (C.T ** 2).sum(0).contiguous()                    # (C**2).sum(0) of the [D, n_units] matrix, as upstream
        self.P = P

    # ------------------------------------------------------------------ plan
    def _build(self, n, out_layer_idx, arena):
        cfg, P, dev = self.cfg, self.P, self.device
        if out_layer_idx >= len(P["layers"]):
            raise ValueError(f"out_layer_idx {out_layer_idx} but only {len(P['layers'])} encoder layers are loaded")
        eps = cfg["layer_norm_eps"]
        plan = ops.Plan()
        arena.begin()
        Z = lambda *s, dt=torch.float32: arena.zeros(*s, dtype=dt)
        io = dict(wave=Z(n))
        wn = Z(n)
        ops.wave_layernorm(io["wave"], wn, n, eps, plan=plan)
        C0 = cfg["conv_dim"][0]
        T = (n - cfg["conv_kernel"][0]) // cfg["conv_stride"][0] + 1
        cur = Z(T, C0)
        c0 = P["conv0"]
        ops.w2v_conv0(wn, c0["w"], c0["b"], c0["g"], c0["be"], cur, n=n, T=T, C=C0, k=cfg["conv_kernel"][0], stride=cfg["conv_stride"][0],
                      eps=eps, plan=plan)
        for cv in P["convs"]:
            Tn = (T - cv["k"]) // cv["s"] + 1
            nxt = Z(Tn, C0)
            ops.gemm(cur, cv["w"], M=Tn, N=C0, Kc=C0, taps=cv["k"], rowsA=T, a_row_mul=cv["s"], a_row_off=0, a_row_step=1,
                     bias=cv["b"], out32=nxt, plan=plan)
            ops.norm(nxt, cv["g"], cv["be"], rows=Tn, C=C0, eps=eps, act=ACT_GELU, out32=nxt, plan=plan)
            cur, T = nxt, Tn
        H, nh, I = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"]
        hd = H // nh
        feat = Z(T, C0)
        ops.norm(cur, *P["fp_ln"], rows=T, C=C0, eps=eps, out32=feat, plan=plan)
        x = Z(T, H)
        ops.gemm(feat, P["fp_w"], M=T, N=H, Kc=C0, bias=P["fp_b"], out32=x, plan=plan)
        G, kw = cfg["num_conv_pos_embedding_groups"], cfg["num_conv_pos_embeddings"]
        cg = H // G
        x2 = Z(T, H)
        ops.gemm(x, P["pos_w"], M=T, N=cg, Kc=cg, taps=kw, lda=H, rowsA=T, a_row_off=-(kw // 2), a_row_step=1, groups=G,
                 a_gstride=cg, w_gstride=cg * kw * cg, c_gcol=cg, bias=P["pos_b"], act=ACT_GELU, residual=x, ldr=H, out32=x2, ldc=H, plan=plan)
        x = x2
        Sp = (T + 15) // 16 * 16
        xn, qk, vt = Z(T, H), Z(T, 2 * H), Z(H, Sp)
        sc, ao, ff = Z(T, nh * Sp), Z(T, H), Z(T, I)
        for l in range(out_layer_idx + 1):
            w = P["layers"][l]
            ops.norm(x, *w["ln1"], rows=T, C=H, eps=eps, out32=xn, plan=plan)
            ops.gemm(xn, w["wqk"], M=T, N=2 * H, Kc=H, bias=w["bqk"], out32=qk, plan=plan)
            ops.gemm(xn, w["wv"], M=T, N=H, Kc=H, bias=w["bv"], out32=vt, ldc=Sp, transpose_out=True, plan=plan)
            # scores[h] = (Q_h K_h^T) * hd^-0.5 -> sc[t][h*Sp + j]
            ops.gemm(qk, qk[:, H:], M=T, N=T, Kc=hd, lda=2 * H, ldw=2 * H, rowsA=T, groups=nh, a_gstride=hd, w_gstride=hd, c_gcol=Sp,
                     alpha=hd ** -0.5, out32=sc, ldc=nh * Sp, plan=plan)
            ops.softmax_segments(sc, rows=T, nseg=nh, n=T, npad=Sp, ldrow=nh * Sp, ldseg=Sp, plan=plan)
            ops.gemm(sc, vt, M=T, N=hd, Kc=Sp, lda=nh * Sp, ldw=Sp, rowsA=T, groups=nh, a_gstride=Sp, w_gstride=hd * Sp, c_gcol=hd,
                     out32=ao, ldc=H, plan=plan)
            ops.gemm(ao, w["wo"], M=T, N=H, Kc=H, bias=w["bo"], residual=x, ldr=H, out32=x, plan=plan)
            ops.norm(x, *w["ln2"], rows=T, C=H, eps=eps, out32=xn, plan=plan)
            ops.gemm(xn, w["w1"], M=T, N=I, Kc=H, bias=w["b1"], act=ACT_GELU, out32=ff, plan=plan)
            ops.gemm(ff, w["w2"], M=T, N=H, Kc=I, bias=w["b2"], residual=x, ldr=H, out32=x, plan=plan)
        nu = P["C"].shape[0]
        dots = Z(T, nu)
        ops.gemm(x, P["C"], M=T, N=nu, Kc=H, out32=dots, plan=plan)
        io["ids"] = Z(T, dt=torch.int64)
        io["margin"] = Z(T)
        io["features"] = x
        ops.kmeans_argmin(x, dots, P["csq"], io["ids"], T=T, D=H, n_units=nu, ldd=nu, margin=io["margin"], plan=plan)
        return plan, io

    @torch.no_grad()
    def predict(self, inp, out_layer_idx, sample_rate=16000):
        """wave FloatTensor[n] (16 kHz, on the GPU) -> unit ids LongTensor[frames]."""
        if not torch.is_tensor(inp):
            raise NotImplementedError("decoding audio files is host I/O outside the hot path; pass a waveform tensor")
        if inp.dim() != 1:
            raise ValueError("expected a mono waveform of shape [n]")
        x = inp.to(self.device, torch.float32)
        if x.numel() % 2 == 1:
            # upstream's Collater(pad_value=2, pad_to_multiple=2) appends one sample of value 2 [RECALLED, unverifiable here]
            x = torch.cat([x, torch.full((1,), 2.0, device=self.device)])
        n = x.numel()
        if _frames(n, self.cfg) < 1:
            raise ValueError(f"waveform of {n} samples is shorter than the 400-sample receptive field")
        nb = bucket(n, SAMPLE_BUCKET)

        def new_arena():
            a = Arena(self.device)
            if nb != n:
                self._build(nb, out_layer_idx, a)          # reserve the workspace at the bucket's capacity
            return a, LRU(MAX_PLANS)
        arena, plans = self._plans.get_or_build((nb, out_layer_idx), new_arena)

        def new_plan():
            plan, io = self._build(n, out_layer_idx, arena)
            return GraphedPlan(plan), io
        gp, io = plans.get_or_build(n, new_plan)
        arena.take(gp)                                       # another length ran on this workspace last -> back to all-zero
        io["wave"].copy_(x)
        gp.run()
        self.last_margin, self.last_io = io["margin"], io   # (tests / diagnostics: features, margins of the last call)
        return io["ids"].clone()
