"""BigVGAN generator drop-in (reference: src/decoder/voicebox/vocoder/models.py:132-313).

Same constructor (`BigVGAN(h, use_cuda_kernel=False)`), `.forward(mel[B,80,T]) -> [B,1,256T]`,
`.remove_weight_norm()`, `.h`, local-directory `from_pretrained`, and the reference's state-dict keys
(`conv_pre.weight_g/_v`, `ups.i.0.*`, `resblocks.n.convs{1,2}.l.*`, `resblocks.n.activations.k.act.{alpha,beta}`,
`...upsample.filter`, `...downsample.lowpass.filter`, `activation_post.*`, `conv_post.*`).

The nn.Modules below only HOLD parameters.  forward() packs them once (bf16, channels-last tap-major)
and replays a pre-built plan of HIP launches: tap-GEMMs on the MFMA cores for every conv / transposed
conv, and one fused kernel per anti-aliased SnakeBeta.  There is no torch arithmetic on the path.
"""
import json
import math
import os

import torch
from torch import nn
from torch.nn.utils import remove_weight_norm, weight_norm

from ... import ops
from ..._lib import ACT_TANH
from ...graph import GraphedPlan
from ...plancache import LRU, Arena, bucket
from .env import AttrDict

T_BUCKET = 64        # mel frames: lengths inside one bucket share ONE workspace (exact-length plans are re-parameterised on it)
MAX_ARENAS = 3       # workspaces kept per model (~1.3 GB at 896 frames, f32); least recently used goes first
MAX_PLANS = 4        # exact-length plans (+ hipGraphs) kept per workspace

LRELU_SLOPE = 0.1


def load_hparams_from_json(path) -> AttrDict:
    with open(path) as f:
        return AttrDict(json.loads(f.read()))


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """Host computation of the low-pass taps (reference: alias_free_torch/filter.py:28-57)."""
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if kernel_size % 2 == 0:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.view(1, 1, kernel_size)


class _Filter(nn.Module):
    def __init__(self, ratio, kernel_size):
        super().__init__()
        self.register_buffer("filter", kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, kernel_size))


class _Down(nn.Module):
    def __init__(self, ratio, kernel_size):
        super().__init__()
        self.lowpass = _Filter(ratio, kernel_size)


class SnakeBeta(nn.Module):
    """Parameter holder for activations.SnakeBeta (reference: vocoder/activations.py:62-120)."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.alpha = nn.Parameter(init.clone() * alpha)
        self.beta = nn.Parameter(init.clone() * alpha)


class Snake(nn.Module):
    """Parameter holder for activations.Snake (reference: vocoder/activations.py:9-59): x + sin^2(x a)/(a + 1e-9)."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.alpha = nn.Parameter(init.clone() * alpha)


def _make_act(name, channels, logscale):
    if name == "snakebeta":
        return SnakeBeta(channels, alpha_logscale=logscale)
    if name == "snake":
        return Snake(channels, alpha_logscale=logscale)
    raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")


def _act_params(act, dev):
    """(alpha, beta) device tensors for usdm_aa_snake; Snake is SnakeBeta with beta = alpha."""
    a = act.alpha.detach().float().to(dev).contiguous()
    b = act.beta.detach().float().to(dev).contiguous() if hasattr(act, "beta") else a
    return a, b


class Activation1d(nn.Module):
    """Parameter holder for alias_free_torch.act.Activation1d (2x up, SnakeBeta, 2x down, 12 taps)."""

    def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise NotImplementedError("usdm_aa_snake implements the 2x / 12-tap configuration BigVGAN uses")
        self.act = activation
        self.upsample = _Filter(up_ratio, up_kernel_size)
        self.downsample = _Down(down_ratio, down_kernel_size)


class AMPBlock1(nn.Module):
    """Parameter holder for AMPBlock1 (reference: vocoder/models.py:28-85)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5), activation=None):
        super().__init__()
        self.h = h
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([
            weight_norm(nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)))
            for d in dilation])
        self.convs2 = nn.ModuleList([
            weight_norm(nn.Conv1d(channels, channels, kernel_size, 1, dilation=1, padding=get_padding(kernel_size, 1)))
            for _ in dilation])
        self.num_layers = len(self.convs1) + len(self.convs2)
        self.activations = nn.ModuleList([
            Activation1d(activation=_make_act(activation, channels, h.snake_logscale)) for _ in range(self.num_layers)])

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            remove_weight_norm(l)


class AMPBlock2(nn.Module):
    """Parameter holder for AMPBlock2 (reference: vocoder/models.py:88-128): x = x + conv_d(act(x)) per dilation."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3), activation=None):
        super().__init__()
        self.h = h
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs = nn.ModuleList([
            weight_norm(nn.Conv1d(channels, channels, kernel_size, 1, dilation=d, padding=get_padding(kernel_size, d)))
            for d in dilation])
        self.num_layers = len(self.convs)
        self.activations = nn.ModuleList([
            Activation1d(activation=_make_act(activation, channels, h.snake_logscale)) for _ in range(self.num_layers)])

    def remove_weight_norm(self):
        for l in self.convs:
            remove_weight_norm(l)


def _pad32(c):
    return (c + 31) // 32 * 32


def _folded_weight(conv):
    """Effective weight of a (possibly weight-normed) conv holder."""
    if hasattr(conv, "weight_g"):
        g, v = conv.weight_g.detach(), conv.weight_v.detach()
        n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
        return (g * v / n).float()
    return conv.weight.detach().float()


def _pack_conv(w, cin_pad, dtype):
    """[Cout, Cin, k] -> [Cout][k*cin_pad] of `dtype` (tap-major, channels-last K)."""
    cout, cin, k = w.shape
    p = torch.zeros(cout, k, cin_pad, dtype=torch.float32, device=w.device)
    p[:, :, :cin] = w.permute(0, 2, 1)
    return p.reshape(cout, k * cin_pad).to(dtype).contiguous()


def _pack_convT(w, k, u, cin_pad, dtype):
    """ConvTranspose1d weight [Cin, Cout, k] -> per output phase p a 2-tap matrix [Cout][2*cin_pad]
    (taps ordered by ascending input offset) plus the first input offset."""
    cin, cout, _ = w.shape
    pad = (k - u) // 2
    mats, offs = [], []
    for p in range(u):
        js = sorted([j for j in range(k) if (p + pad - j) % u == 0], key=lambda j: (p + pad - j) // u)
        o = [(p + pad - j) // u for j in js]
        if len(js) != 2 or o[1] - o[0] != 1:
            raise NotImplementedError("ConvTranspose1d packing assumes kernel == 2*stride (BigVGAN upsamplers)")
        m = torch.zeros(cout, 2, cin_pad, dtype=torch.float32, device=w.device)
        for t, j in enumerate(js):
            m[:, t, :cin] = w[:, :, j].T
        mats.append(m.reshape(cout, 2 * cin_pad).to(dtype).contiguous())
        offs.append(o[0])
    return mats, offs


class BigVGAN(nn.Module):
    def __init__(self, h: AttrDict, use_cuda_kernel: bool = False, compute_dtype=torch.float32):
        super().__init__()
        self.h = h
        # MFMA operand type of the convolutions: float32 = exact-f32 matrix cores (the reference's
        # arithmetic, default); bfloat16 = 16x faster matrix rate, ~20-25 dB SNR on random weights
        # because sin^2(x e^a) amplifies operand rounding (measured).
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = compute_dtype
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        self.conv_pre = weight_norm(nn.Conv1d(h.num_mels, h.upsample_initial_channel, 7, 1, padding=3))
        resblock = AMPBlock1 if h.resblock == "1" else AMPBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(nn.ModuleList([weight_norm(nn.ConvTranspose1d(
                h.upsample_initial_channel // (2 ** i), h.upsample_initial_channel // (2 ** (i + 1)), k, u,
                padding=(k - u) // 2))]))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = h.upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(resblock(h, ch, k, d, activation=h.activation))
        self.activation_post = Activation1d(activation=_make_act(h.activation, ch, h.snake_logscale))
        self.conv_post = weight_norm(nn.Conv1d(ch, 1, 7, 1, padding=3))
        for m in list(self.ups.modules()) + [self.conv_post]:
            if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d)):
                m.weight_v.data.normal_(0.0, 0.01) if hasattr(m, "weight_v") else m.weight.data.normal_(0.0, 0.01)
        self._packed = None
        self._plans = LRU(MAX_ARENAS)      # (frame bucket, device) -> (Arena, LRU of exact-length plans)

    # ------------------------------------------------------------------ reference API
    def remove_weight_norm(self):
        print("Removing weight norm...")
        for l in self.ups:
            for l_i in l:
                remove_weight_norm(l_i)
        for l in self.resblocks:
            l.remove_weight_norm()
        remove_weight_norm(self.conv_pre)
        remove_weight_norm(self.conv_post)
        self.invalidate()

    def invalidate(self):
        """Drop packed weights / plans (call after changing parameters)."""
        self._packed, self._plans = None, LRU(MAX_ARENAS)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, cache_dir=None, map_location="cpu", use_cuda_kernel=False, **kw):
        """Local-directory loading of `config.json` + `bigvgan_generator.pt['generator']`
        (reference: models.py:234-313; hub download is not available offline)."""
        d = pretrained_model_name_or_path
        if not os.path.isdir(d) and cache_dir is not None:      # the reference's call: hub name + cache_dir (model_util.py:64-67)
            from ...checkpoints import resolve_local
            d = resolve_local(cache_dir, d, ("config.json",))
        if not os.path.isdir(d):
            raise FileNotFoundError(f"{d}: only local directories can be loaded (no network); expected config.json + bigvgan_generator.pt")
        h = load_hparams_from_json(os.path.join(d, "config.json"))
        model = cls(h, use_cuda_kernel=use_cuda_kernel)
        ckpt = torch.load(os.path.join(d, "bigvgan_generator.pt"), map_location=map_location)
        try:
            model.load_state_dict(ckpt["generator"])
        except RuntimeError:
            print("[INFO] the pretrained checkpoint does not contain weight norm. Loading the checkpoint after removing weight norm!")
            model.remove_weight_norm()
            model.load_state_dict(ckpt["generator"])
        return model

    # ------------------------------------------------------------------ packing
    def _pack(self, dev):
        h, cd = self.h, self.compute_dtype
        P = {"taps": self.activation_post.upsample.filter.detach().view(-1).float().cpu().tolist(),
             "taps_dn": self.activation_post.downsample.lowpass.filter.detach().view(-1).float().cpu().tolist()}
        mels_pad = _pad32(h.num_mels)
        P["pre_w"] = _pack_conv(_folded_weight(self.conv_pre).to(dev), mels_pad, cd)
        P["pre_b"] = self.conv_pre.bias.detach().float().to(dev).contiguous()
        P["ups"], P["blocks"] = [], []
        ch = h.upsample_initial_channel
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            conv = self.ups[i][0]
            mats, offs = _pack_convT(_folded_weight(conv).to(dev), k, u, _pad32(ch), cd)
            P["ups"].append(dict(mats=mats, offs=offs, bias=conv.bias.detach().float().to(dev).contiguous(), u=u, cin=ch, cout=ch // 2))
            ch //= 2
            cp = _pad32(ch)
            for j in range(self.num_kernels):
                blk = self.resblocks[i * self.num_kernels + j]
                layers = []
                for l, d in enumerate(blk.dilation):
                    if isinstance(blk, AMPBlock1):
                        layers.append(dict(
                            w1=_pack_conv(_folded_weight(blk.convs1[l]).to(dev), cp, cd), b1=blk.convs1[l].bias.detach().float().to(dev).contiguous(),
                            w2=_pack_conv(_folded_weight(blk.convs2[l]).to(dev), cp, cd), b2=blk.convs2[l].bias.detach().float().to(dev).contiguous(),
                            a1=_act_params(blk.activations[2 * l].act, dev), a2=_act_params(blk.activations[2 * l + 1].act, dev), d=d))
                    else:
                        layers.append(dict(
                            w1=_pack_conv(_folded_weight(blk.convs[l]).to(dev), cp, cd), b1=blk.convs[l].bias.detach().float().to(dev).contiguous(),
                            a1=_act_params(blk.activations[l].act, dev), d=d))
                P["blocks"].append(dict(k=blk.kernel_size, layers=layers, two=isinstance(blk, AMPBlock1)))
        P["post_a"] = _act_params(self.activation_post.act, dev)
        P["post_w"] = _pack_conv(_folded_weight(self.conv_post).to(dev), _pad32(ch), cd)
        P["post_b"] = self.conv_post.bias.detach().float().to(dev).contiguous()
        P["logscale"] = bool(h.snake_logscale)
        return P

    def _build_plan(self, T, dev, arena):
        h, P = self.h, self._packed
        plan = ops.Plan()
        arena.begin()
        z32 = lambda *s: arena.zeros(*s, dtype=torch.float32)
        f32 = self.compute_dtype == torch.float32
        z16 = lambda *s: arena.zeros(*s, dtype=self.compute_dtype)  # MFMA operand buffers
        ok = lambda t: dict(out32=t) if f32 else dict(out16=t)  # route a kernel output to an operand buffer
        taps, taps_dn, ls = P["taps"], P["taps_dn"], P["logscale"]
        mels_pad = _pad32(h.num_mels)
        io = dict(mel=z32(1, h.num_mels, T), scale=1.0, shift=0.0)
        mel16 = z16(T, mels_pad)
        io["mel16"] = mel16
        c0 = h.upsample_initial_channel
        cur16 = z16(T, _pad32(c0))
        ops.gemm(mel16, P["pre_w"], M=T, N=c0, Kc=mels_pad, taps=7, rowsA=T, a_row_off=-3, a_row_step=1, bias=P["pre_b"],
                 ldc=_pad32(c0), plan=plan, **ok(cur16))
        Tc, ch = T, c0
        last32 = None
        for i, up in enumerate(P["ups"]):
            u, cin, cout = up["u"], up["cin"], up["cout"]
            cinp, cp = _pad32(cin), _pad32(cout)
            Tn = Tc * u
            x32 = z32(Tn, cp)
            for p in range(u):
                ops.gemm(cur16, up["mats"][p], M=Tc, N=cout, Kc=cinp, taps=2, rowsA=Tc, a_row_off=up["offs"][p], a_row_step=1,
                         bias=up["bias"], out32=x32, ldc=cp, c_row_mul=u, c_row_off=p, plan=plan)
            a16, xt32 = z16(Tn, cp), z32(Tn, cp)
            ys = []
            for j in range(self.num_kernels):
                blk = P["blocks"][i * self.num_kernels + j]
                k = blk["k"]
                y32 = z32(Tn, cp)
                src = x32
                for lay in blk["layers"]:
                    d = lay["d"]
                    ops.aa_snake(src, lay["a1"][0], lay["a1"][1], taps, taps_dn, T=Tn, C=cp, Creal=cout, logscale=ls, plan=plan, **ok(a16))
                    if not blk["two"]:   # AMPBlock2: x = x + conv_d(act(x))
                        ops.gemm(a16, lay["w1"], M=Tn, N=cout, Kc=cp, taps=k, rowsA=Tn, a_row_off=-get_padding(k, d), a_row_step=d,
                                 bias=lay["b1"], residual=src, ldr=cp, out32=y32, ldc=cp, plan=plan)
                        src = y32
                        continue
                    ops.gemm(a16, lay["w1"], M=Tn, N=cout, Kc=cp, taps=k, rowsA=Tn, a_row_off=-get_padding(k, d), a_row_step=d,
                             bias=lay["b1"], out32=xt32, ldc=cp, plan=plan)
                    ops.aa_snake(xt32, lay["a2"][0], lay["a2"][1], taps, taps_dn, T=Tn, C=cp, Creal=cout, logscale=ls, plan=plan, **ok(a16))
                    ops.gemm(a16, lay["w2"], M=Tn, N=cout, Kc=cp, taps=k, rowsA=Tn, a_row_off=-get_padding(k, 1), a_row_step=1,
                             bias=lay["b2"], residual=src, ldr=cp, out32=y32, ldc=cp, plan=plan)
                    src = y32
                ys.append(y32)
            if self.num_kernels != 3:
                raise NotImplementedError("usdm_sum3_scale assumes three AMP blocks per stage")
            last = i == len(P["ups"]) - 1
            nxt16 = None if last else z16(Tn, cp)
            last32 = z32(Tn, cp) if last else None
            if last:
                ops.sum3_scale(ys[0], ys[1], ys[2], 1.0 / 3.0, out32=last32, plan=plan)
            else:
                ops.sum3_scale(ys[0], ys[1], ys[2], 1.0 / 3.0, plan=plan, **ok(nxt16))
            cur16, Tc, ch = nxt16, Tn, cout
        cp = _pad32(ch)
        a16 = z16(Tc, cp)
        ops.aa_snake(last32, P["post_a"][0], P["post_a"][1], taps, taps_dn, T=Tc, C=cp, Creal=ch, logscale=ls, plan=plan, **ok(a16))
        out = z32(1, 1, Tc)
        ops.gemm(a16, P["post_w"], M=Tc, N=1, Kc=cp, taps=7, rowsA=Tc, a_row_off=-3, a_row_step=1, bias=P["post_b"], act=ACT_TANH,
                 out32=out, ldc=1, plan=plan)
        io["out"] = out
        return plan, io

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, _scale=1.0, _shift=0.0):
        """mel [B, num_mels, T] f32 on the GPU -> waveform [B, 1, T*hop] (models.py:189-211).
        `_scale/_shift` fold the mel de-normalisation of model_util.py:103 into the layout kernel."""
        if not x.is_cuda:
            raise RuntimeError("BigVGAN (usdm_amd) runs on the MI355X only; there is no CPU fallback")
        if x.dim() != 3 or x.shape[1] != self.h.num_mels:
            raise ValueError(f"expected mel of shape [B, {self.h.num_mels}, T], got {tuple(x.shape)}")
        dev = x.device
        if self._packed is None:
            self._packed = self._pack(dev)
        B, _, T = x.shape
        Tb = bucket(T, T_BUCKET)

        def new_arena():
            a = Arena(dev)
            if Tb != T:
                self._build_plan(Tb, dev, a)      # reserve the workspace at the bucket's capacity
            return a, LRU(MAX_PLANS)
        arena, plans = self._plans.get_or_build((Tb, dev.index, self.compute_dtype), new_arena)

        def new_plan():
            plan, io = self._build_plan(T, dev, arena)
            return GraphedPlan(plan), io
        gp, io = plans.get_or_build(T, new_plan)
        arena.take(gp)                            # another length ran on this workspace last -> back to all-zero
        outs = []
        for b in range(B):
            ops.cf_to_cl(x[b:b + 1].contiguous().float(), B=1, C=self.h.num_mels, T=T, Cpad=io["mel16"].shape[1],
                         scale=_scale, shift=_shift,
                         **(dict(out32=io["mel16"]) if self.compute_dtype == torch.float32 else dict(out16=io["mel16"])))
            gp.run()
            outs.append(io["out"].clone())        # a NEW tensor, as the reference returns (the plan's buffer is reused)
        return outs[0] if B == 1 else torch.cat(outs, 0)
