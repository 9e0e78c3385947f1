"""GPU parity of the stand-alone kernels (norm, fused AA-snake, attention) against fp64/fp32 CPU math
and against the golden vector of the reference's Activation1d."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _r(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _close(out, ref, tol, what=""):
    ref = ref.double()
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= tol * (ref.abs().max().item() + 1e-12), f"{what}: err {err}"


@pytest.mark.parametrize("C", [512, 1024, 1280, 4096])
def test_layernorm_and_rms(dev, C):
    from usdm_amd import ops
    rows = 37
    x, r = _r((rows, C), 1), _r((rows, C), 2)
    g, b = _r((C,), 3), _r((C,), 4)
    o32 = torch.zeros(rows, C, device=dev)
    o16 = torch.zeros(rows, C, device=dev, dtype=torch.bfloat16)
    s32 = torch.zeros(rows, C, device=dev)
    ops.norm(x.to(dev), g.to(dev), b.to(dev), rows=rows, C=C, res=r.to(dev), out32=o32, out16=o16, sum32=s32)
    ref = torch.nn.functional.layer_norm((x + r).double(), (C,), g.double(), b.double(), 1e-5)
    _close(o32, ref, 2e-6, "ln f32")
    _close(o16.float(), ref, 1e-2, "ln bf16")
    _close(s32, x + r, 1e-7, "sum")
    # further f32 addends (split-K partials): x + res + res2[0] + res2[1], summed in that order
    r2 = _r((2, rows, C), 5)
    ops.norm(x.to(dev), g.to(dev), b.to(dev), rows=rows, C=C, res=r.to(dev), res2=r2.to(dev), n_res2=2, res2_stride=rows * C,
             out32=o32, sum32=s32)
    tot = ((x + r) + r2[0]) + r2[1]
    assert torch.equal(s32.cpu(), tot), "res2 sum order"
    _close(o32, torch.nn.functional.layer_norm(tot.double(), (C,), g.double(), b.double(), 1e-5), 2e-6, "ln res2")
    # gelu + valid_len mask
    vl = torch.tensor([20, 5], dtype=torch.int32, device=dev)
    ops.norm(x[:36].contiguous().to(dev), g.to(dev), b.to(dev), rows=36, C=C, act=1, valid_len=vl, rows_per_batch=18, out32=o32)
    ref = torch.nn.functional.gelu(torch.nn.functional.layer_norm(x[:36].double(), (C,), g.double(), b.double(), 1e-5))
    ref[18 + 5:36] = 0
    _close(o32[:36], ref, 2e-6, "ln gelu mask")
    # RMSNorm, HF bf16 semantics
    xb = x.to(torch.bfloat16)
    gb = g.to(torch.bfloat16)
    ops.norm(xb.to(dev), gb.float().to(dev), None, rows=rows, C=C, rms=True, round_bf16=True, eps=1e-5, out16=o16)
    xf = xb.float()
    ref = gb * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16)
    mism = (o16.cpu() != ref).float().mean().item()
    assert mism < 2e-3, mism  # rsqrt ulp differences may flip a few bf16 roundings


def test_aa_snake_golden_and_large(dev):
    from oracle import bigvgan_oracle as BO
    from usdm_amd import ops
    d = np.load(os.path.join(G, "bigvgan_act.npz"))
    taps = BO.aa_filter_taps()
    x = torch.from_numpy(d["x"])  # [1,C,T]
    C, T = x.shape[1], x.shape[2]
    Cp = 32
    xcl = torch.zeros(T, Cp)
    xcl[:, :C] = x[0].T
    out = torch.full((T, Cp), 7.0, device=dev)
    ops.aa_snake(xcl.to(dev), torch.from_numpy(d["alpha"]).to(dev), torch.from_numpy(d["beta"]).to(dev), taps, taps,
                 T=T, C=Cp, Creal=C, out32=out)
    _close(out[:, :C].T, torch.from_numpy(d["y"])[0], 2e-5, "act golden")
    assert out[:, C:].abs().max().item() == 0.0
    # larger, ragged T, several chunk sizes
    for (C, T, L) in [(96, 1000, 0), (32, 5000, 32), (64, 7, 8), (768, 345, 16)]:
        x = _r((1, C, T), 5, 1.5)
        al, be = _r((C,), 6, 0.4), _r((C,), 7, 0.4)
        ref = BO.activation1d(x.double(), al.double(), be.double(), taps.double())[0].T
        o32 = torch.zeros(T, C, device=dev)
        o16 = torch.zeros(T, C, device=dev, dtype=torch.bfloat16)
        ops.aa_snake(x[0].T.contiguous().to(dev), al.to(dev), be.to(dev), taps, taps, T=T, C=C, out32=o32, out16=o16, L=L)
        _close(o32, ref, 2e-5, f"act C{C} T{T}")
        _close(o16.float(), ref, 1e-2, f"act bf16 C{C} T{T}")


def _attn_ref(q, k, v, mode, slopes=None, kv_len=None, scale=1.0, q_pos0=0):
    B, H, Sq, D = q.shape
    Hkv, Skv = k.shape[1], k.shape[2]
    k = k.repeat_interleave(H // Hkv, 1)
    v = v.repeat_interleave(H // Hkv, 1)
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    qi = torch.arange(Sq).view(-1, 1) + q_pos0
    kj = torch.arange(Skv).view(1, -1)
    if mode == 0:
        bias = -(slopes.double().view(1, H, 1, 1)) * (qi - kj).abs().double()
        bias[..., 0] = 0
        s = s + bias
        for b in range(B):
            s[b, :, :, kv_len[b]:] = -float("inf")
    else:
        s = s.masked_fill(kj > qi, -float("inf"))
    return (torch.softmax(s, -1) @ v.double())  # [B,H,Sq,D]


@pytest.mark.parametrize("mode,dh,B,H,Hkv,Sq", [(0, 64, 2, 4, 4, 300), (0, 64, 1, 16, 16, 1118), (1, 128, 1, 8, 2, 200),
                                                (1, 128, 1, 4, 1, 577), (0, 128, 1, 2, 2, 130), (1, 64, 2, 2, 2, 64)])
def test_attention(dev, mode, dh, B, H, Hkv, Sq):
    _attention_case(dev, mode, dh, B, H, Hkv, Sq)


@pytest.mark.parametrize("v16", ["0", "1"])
@pytest.mark.parametrize("B,H,Sq", [(2, 4, 300), (1, 16, 1118), (1, 8, 17), (3, 8, 129), (2, 16, 64)])
def test_attention_voicebox_form_both_kernels(dev, monkeypatch, v16, B, H, Sq):
    """The bidirectional ALiBi form (d = 64, MHA) through the 16-query-wave kernel (default) and the 32-query-wave one
    (USDM_ATTN_V16=0): ragged kv_len, query counts that end inside a wave / inside a workgroup, fewer than 8 heads."""
    monkeypatch.setenv("USDM_ATTN_V16", v16)
    _attention_case(dev, 0, 64, B, H, H, Sq)


def _attention_case(dev, mode, dh, B, H, Hkv, Sq):
    from usdm_amd import ops
    Spad = (Sq + 63) // 64 * 64
    bf = torch.bfloat16
    q = _r((B, H, Sq, dh), 1, 0.5).to(bf)
    k = _r((B, Hkv, Sq, dh), 2, 0.5).to(bf)
    v = _r((B, Hkv, Sq, dh), 3, 1.0).to(bf)
    slopes = torch.tensor([2.0 ** (-(i + 1) / 2) for i in range(H)])
    kv_len = torch.tensor([Sq, max(1, Sq - 37), 1][:B], dtype=torch.int32)
    scale = 1.0 if mode == 0 else dh ** -0.5
    qd = torch.zeros(B, H, Spad, dh, dtype=bf, device=dev); qd[:, :, :Sq] = q.to(dev)
    kd = torch.zeros(B, Hkv, Spad, dh, dtype=bf, device=dev); kd[:, :, :Sq] = k.to(dev)
    vt = torch.zeros(B, Hkv, dh, Spad, dtype=bf, device=dev); vt[:, :, :, :Sq] = v.transpose(-1, -2).to(dev)
    o = torch.zeros(B, Sq, H * dh, dtype=bf, device=dev)
    ops.attention(qd, kd, vt, o, mode=mode, dh=dh, B=B, Hq=H, Hkv=Hkv, Sq=Sq, Skv=Sq, Skv_alloc=Spad,
                  q_strides=(H * Spad * dh, Spad * dh, dh), k_strides=(Hkv * Spad * dh, Spad * dh, dh),
                  v_strides=(Hkv * dh * Spad, dh * Spad, Spad), o_strides=(Sq * H * dh, H * dh),
                  scale=scale, kv_len=kv_len.to(dev) if mode == 0 else None, slopes=slopes.to(dev) if mode == 0 else None)
    ref = _attn_ref(q.float(), k.float(), v.float(), mode, slopes, kv_len, scale)
    ref = ref.permute(0, 2, 1, 3).reshape(B, Sq, H * dh)
    _close(o.float(), ref, 2e-2, f"attention mode{mode} dh{dh}")
