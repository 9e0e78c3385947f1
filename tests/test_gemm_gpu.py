"""GPU parity of the universal tap-GEMM (usdm_gemm) against plain fp64 torch on the CPU."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _tol(dtype, K):
    return (2e-2 if dtype == torch.bfloat16 else 2e-5)


def _check(out, ref, dtype, K, what):
    ref = ref.double()
    err = (out.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    assert err <= _tol(dtype, K) * scale, f"{what}: max err {err} vs scale {scale}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(70, 50, 64), (300, 200, 96), (1118 * 2, 1024, 1024), (129, 257, 32 * 7)])
def test_linear(dev, dtype, M, N, K):
    from usdm_amd import ops
    if dtype == torch.float32 and K % 16:
        pytest.skip()
    A = _rand((M, K), dtype, 1)
    W = _rand((N, K), dtype, 2)
    b = _rand((N,), torch.float32, 3)
    R = _rand((M, N), torch.float32, 4)
    out = torch.full((M, N), float("nan"), device=dev)
    out16 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), residual=R.to(dev), ldr=N, out32=out, out16=out16)
    ref = A.double() @ W.double().T + b.double() + R.double()
    _check(out, ref, dtype, K, "linear f32 out")
    _check(out16.float(), ref, torch.bfloat16, K, "linear bf16 out")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_and_transpose(dev, dtype):
    from usdm_amd import ops
    M, N, K = 200, 96, 64
    A, W = _rand((M, K), dtype, 1, 0.3), _rand((N, K), dtype, 2, 0.3)
    b = _rand((N,), torch.float32, 3)
    out = torch.zeros((N, M), device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), act=1, out32=out, ldc=M, transpose_out=True)
    ref = torch.nn.functional.gelu(A.double() @ W.double().T + b.double()).T
    _check(out, ref, dtype, K, "gelu+transpose")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("k,dil,stride", [(3, 1, 1), (7, 3, 1), (11, 5, 1), (3, 1, 2), (2, 1, 2)])
def test_conv1d_taps(dev, dtype, k, dil, stride):
    """Conv1d on channels-last activations == tap-GEMM (vocoder/models.py:33-49 shapes, XLS-R strides)."""
    from usdm_amd import ops
    T, Cin, Cout = 333, 64, 48
    x = _rand((1, Cin, T), dtype, 5)
    w = _rand((Cout, Cin, k), dtype, 6, 0.2)
    b = _rand((Cout,), torch.float32, 7)
    pad = (k * dil - dil) // 2 if stride == 1 else 0
    ref = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=stride, padding=pad, dilation=dil)[0].T
    Tout = ref.shape[0]
    A = x[0].T.contiguous().to(dev)                      # [T, Cin]
    Wp = w.permute(0, 2, 1).contiguous().reshape(Cout, k * Cin).to(dev)  # [Cout][tap][Cin]
    out = torch.zeros((Tout, Cout), device=dev)
    ops.gemm(A, Wp, M=Tout, N=Cout, Kc=Cin, taps=k, rowsA=T, a_row_mul=stride, a_row_off=-pad, a_row_step=dil,
             bias=b.to(dev), out32=out)
    _check(out, ref, dtype, k * Cin, f"conv k{k} d{dil} s{stride}")


def test_conv_transpose_phases(dev):
    """ConvTranspose1d(k=2u, stride u, pad (k-u)//2) as u two-tap phase GEMMs (vocoder/models.py:157-162)."""
    from usdm_amd import ops
    dtype = torch.bfloat16
    for (k, u) in [(8, 4), (4, 2)]:
        T, Cin, Cout = 100, 64, 32
        x = _rand((1, Cin, T), dtype, 8)
        w = _rand((Cin, Cout, k), dtype, 9, 0.2)
        b = _rand((Cout,), torch.float32, 10)
        pad = (k - u) // 2
        ref = torch.nn.functional.conv_transpose1d(x.double(), w.double(), b.double(), stride=u, padding=pad)[0].T
        A = x[0].T.contiguous().to(dev)
        out = torch.zeros((T * u, Cout), device=dev)
        for p in range(u):
            js = [j for j in range(k) if (p + pad - j) % u == 0]
            js.sort(key=lambda j: (p + pad - j) // u)  # ascending input offset
            offs = [(p + pad - j) // u for j in js]
            assert len(js) == 2 and offs[1] - offs[0] == 1
            Wp = torch.stack([w[:, :, j].T for j in js], dim=1).contiguous().reshape(Cout, 2 * Cin).to(dev)
            ops.gemm(A, Wp, M=T, N=Cout, Kc=Cin, taps=2, rowsA=T, a_row_off=offs[0], a_row_step=1,
                     bias=b.to(dev), out32=out, c_row_mul=u, c_row_off=p)
        _check(out, ref, dtype, 2 * Cin, f"convT k{k} u{u}")


def test_grouped_conv_batch(dev):
    """Grouped Conv1d k31 g16 pad 15 over a batch of 2 (networks.py:70-76)."""
    from usdm_amd import ops
    dtype = torch.bfloat16
    B, T, G, Cg, k = 2, 150, 4, 64, 31
    C_ = G * Cg
    x = _rand((B, C_, T), dtype, 11)
    w = _rand((C_, Cg, k), dtype, 12, 0.1)
    b = _rand((C_,), torch.float32, 13)
    ref = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), padding=k // 2, groups=G).permute(0, 2, 1)
    A = x.permute(0, 2, 1).contiguous().to(dev)  # [B, T, C]
    Wp = w.reshape(G, Cg, Cg, k).permute(0, 1, 3, 2).contiguous().reshape(G, Cg, k * Cg).to(dev)
    out = torch.zeros((B, T, C_), device=dev)
    ops.gemm(A, Wp, M=T, N=Cg, Kc=Cg, taps=k, lda=C_, rowsA=T, a_row_off=-(k // 2), a_row_step=1,
             groups=G, batch=B, a_gstride=Cg, w_gstride=Cg * k * Cg, a_bstride=T * C_, c_gcol=Cg, c_bstride=T,
             bias=b.to(dev), act=1, out32=out, ldc=C_)
    _check(out, torch.nn.functional.gelu(ref), dtype, k * Cg, "grouped conv")


def test_two_source_k(dev):
    """Linear over cat[h, skip] without materialising the concat (networks.py:364)."""
    from usdm_amd import ops
    M, H = 100, 64
    buf = _rand((2, M, H), torch.bfloat16, 14).to(dev)
    W = _rand((H, 2 * H), torch.bfloat16, 15)
    out = torch.zeros((M, H), device=dev)
    ops.gemm(buf, W.to(dev), M=M, N=H, Kc=H, taps=2, lda=H, rowsA=M, a_tap_stride=M * H, out32=out)
    ref = torch.cat([buf[0].cpu(), buf[1].cpu()], -1).double() @ W.double().T
    _check(out, ref, torch.bfloat16, 2 * H, "two-source")


def test_swiglu_and_qkv(dev):
    from usdm_amd import ops
    M, K, F = 70, 64, 64
    A = _rand((M, K), torch.bfloat16, 16)
    Wg, Wu = _rand((F, K), torch.bfloat16, 17, 0.3), _rand((F, K), torch.bfloat16, 18, 0.3)
    Wp = torch.stack([Wg.reshape(F // 16, 16, K), Wu.reshape(F // 16, 16, K)], 1).reshape(2 * F, K).contiguous()
    out = torch.zeros((M, F), device=dev)
    ops.gemm(A.to(dev), Wp.to(dev), M=M, N=2 * F, Kc=K, act=3, out32=out, ldc=F)
    g, u = A.double() @ Wg.double().T, A.double() @ Wu.double().T
    _check(out, torch.nn.functional.silu(g) * u, torch.bfloat16, K, "swiglu")
    # qkv head split
    B, S, Hh, D = 2, 37, 2, 64
    Spad = 64
    A = _rand((B * S, K), torch.bfloat16, 19)
    W = _rand((3 * Hh * D, K), torch.bfloat16, 20, 0.3)
    q = torch.zeros((B, Hh, Spad, D), device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    v = torch.zeros((B, Hh, D, Spad), device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), M=B * S, N=3 * Hh * D, Kc=K, qkv=dict(S=S, Spad=Spad, H=Hh, D=D, q=q, k=k, v=v))
    ref = (A.double() @ W.double().T).reshape(B, S, 3, Hh, D)
    _check(q[:, :, :S].float(), ref[:, :, 0].permute(0, 2, 1, 3), torch.bfloat16, K, "q")
    _check(k[:, :, :S].float(), ref[:, :, 1].permute(0, 2, 1, 3), torch.bfloat16, K, "k")
    _check(v[:, :, :, :S].float(), ref[:, :, 2].permute(0, 2, 3, 1), torch.bfloat16, K, "v^T")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N", [(300, 200), (129, 50), (257, 130)])
def test_epilogue_variants(dev, dtype, M, N):
    """Every store path of the epilogue: activation + residual (rolled loop), ragged N with activation (scalar path),
    transposed output with bias + activation + residual (transposed LDS tile), bf16 residual, alpha."""
    from usdm_amd import ops
    K = 64
    A, W = _rand((M, K), dtype, 1, 0.3), _rand((N, K), dtype, 2, 0.3)
    b = _rand((N,), torch.float32, 3)
    R = _rand((M, N), torch.float32, 4)
    pre = 0.5 * (A.double() @ W.double().T) + b.double()
    # row-major, GELU then residual
    out = torch.zeros((M, N), device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), alpha=0.5, act=1, residual=R.to(dev), ldr=N, out32=out)
    _check(out, torch.nn.functional.gelu(pre) + R.double(), dtype, K, "gelu+residual")
    # tanh, bf16 output, no residual
    out16 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), alpha=0.5, act=4, out16=out16)
    _check(out16.float(), torch.tanh(pre), torch.bfloat16, K, "tanh bf16")
    # transposed output [N][M] with bias, GELU and a residual laid out like the output's logical [M][N]
    outT = torch.zeros((N, M), device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), alpha=0.5, act=1, residual=R.to(dev), ldr=N, out32=outT, ldc=M,
             transpose_out=True)
    _check(outT, (torch.nn.functional.gelu(pre) + R.double()).T, dtype, K, "transpose+gelu+residual")
    # bf16 residual, plain
    Rb = R.to(torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), alpha=0.5, residual=Rb.to(dev), ldr=N, out32=out)
    _check(out, pre + Rb.double(), dtype, K, "bf16 residual")


def test_qkv_epilogue_full_shape(dev):
    """Head-split QKV epilogue at the Voicebox layer shape (several tiles per part, bias, V^T through the transposed tile)."""
    from usdm_amd import ops
    B, S, Hh, D, K = 2, 1118, 16, 64, 128
    Spad = (S + 63) // 64 * 64
    A = _rand((B * S, K), torch.bfloat16, 21)
    W = _rand((3 * Hh * D, K), torch.bfloat16, 22, 0.2)
    b = _rand((3 * Hh * D,), torch.float32, 23)
    q = torch.zeros((B, Hh, Spad, D), device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    v = torch.zeros((B, Hh, D, Spad), device=dev, dtype=torch.bfloat16)
    ops.gemm(A.to(dev), W.to(dev), M=B * S, N=3 * Hh * D, Kc=K, bias=b.to(dev), qkv=dict(S=S, Spad=Spad, H=Hh, D=D, q=q, k=k, v=v))
    ref = (A.double() @ W.double().T + b.double()).reshape(B, S, 3, Hh, D)
    _check(q[:, :, :S].float(), ref[:, :, 0].permute(0, 2, 1, 3), torch.bfloat16, K, "q")
    _check(k[:, :, :S].float(), ref[:, :, 1].permute(0, 2, 1, 3), torch.bfloat16, K, "k")
    _check(v[:, :, :, :S].float(), ref[:, :, 2].permute(0, 2, 3, 1), torch.bfloat16, K, "v^T")
    assert float(v[:, :, :, S:].abs().max()) == 0.0 and float(q[:, :, S:].abs().max()) == 0.0   # padding untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,S", [(300, 200, 512, 2), (2236, 1024, 4096, 2), (129, 130, 448, 3), (499, 1280, 5120, 4)])
def test_split_k(dev, dtype, M, N, K, S):
    """split-K partial outputs: their sum equals the unsplit GEMM; bias and residual are applied exactly once."""
    from usdm_amd import ops
    A, W = _rand((M, K), dtype, 1, 0.3), _rand((N, K), dtype, 2, 0.3)
    b = _rand((N,), torch.float32, 3)
    R = _rand((M, N), torch.float32, 4)
    parts = torch.full((S, M, N), float("nan"), device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, bias=b.to(dev), residual=R.to(dev), ldr=N, out32=parts, split_k=S,
             c_split_stride=M * N)
    ref = A.double() @ W.double().T + b.double() + R.double()
    _check(parts.sum(0), ref, dtype, K, f"split_k={S}")


@pytest.fixture
def tile_env():
    import os
    old = os.environ.get("USDM_GEMM_TILE")
    yield lambda t: os.environ.__setitem__("USDM_GEMM_TILE", str(t))
    if old is None:
        os.environ.pop("USDM_GEMM_TILE", None)
    else:
        os.environ["USDM_GEMM_TILE"] = old


@pytest.mark.parametrize("tile", [12, 13, 14])
def test_pingpong_tiles_bit_identical(dev, tile, tile_env):
    """The 8-wave ping-pong tiles (256x128 / 288x128 / 128x128, USDM_GEMM_TILE=12 / 13 / 14) accumulate every output in the same K order as the
    128x128 LDS-DMA tile (4): plain, residual, split-K and head-split epilogues must be BIT-identical, ragged M / N / K (a K that
    ends in half a 64-deep step - with the step counts 3k + 5 at which that half step falls on a steady-state round of the
    ring -, a split whose last step is short) included; the packed GELU epilogue within one bf16 ulp."""
    from usdm_amd import ops

    def run(t, f):
        tile_env(t)
        return f()

    for (M, N, K) in [(600, 520, 352), (2236, 1024, 1024), (257, 384, 64), (1118, 4096, 320), (600, 520, 480), (300, 256, 288)]:
        A, W = _rand((M, K), torch.bfloat16, 31, 0.3).to(dev), _rand((N, K), torch.bfloat16, 32, 0.3).to(dev)
        b, R = _rand((N,), torch.float32, 33).to(dev), _rand((M, N), torch.float32, 34).to(dev)

        def plain():
            o32 = torch.full((M, N), float("nan"), device=dev); o16 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            ops.gemm(A, W, M=M, N=N, Kc=K, bias=b, residual=R, ldr=N, out32=o32, out16=o16)
            return o32, o16

        def split():
            p = torch.full((3, M, N), float("nan"), device=dev)
            ops.gemm(A, W, M=M, N=N, Kc=K, bias=b, residual=R, ldr=N, out32=p, split_k=3, c_split_stride=M * N)
            return (p,)

        def gelu():
            o16 = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
            ops.gemm(A, W, M=M, N=N, Kc=K, bias=b, act=1, out16=o16)
            return (o16,)

        def transposed():
            oT = torch.zeros((N, M), device=dev)
            ops.gemm(A, W, M=M, N=N, Kc=K, bias=b, out32=oT, ldc=M, transpose_out=True)
            return (oT,)

        for name, f in (("plain", plain), ("split", split), ("transposed", transposed)):
            if name == "split" and K < 3 * 64:
                continue
            for x, y in zip(run(4, f), run(tile, f)):
                assert torch.equal(x, y), f"{name} {M}x{N}x{K}: tile {tile} differs from tile 4"
        (g4,), (gt,) = run(4, gelu), run(tile, gelu)
        ulp = (g4.view(torch.int16).int() - gt.view(torch.int16).int()).abs().max().item()
        assert ulp <= 1, f"gelu {M}x{N}x{K}: {ulp} bf16 ulps"
        ref = torch.nn.functional.gelu(A.double().cpu() @ W.double().cpu().T + b.double().cpu())
        _check(gt.float(), ref, torch.bfloat16, K, "packed gelu")

    # batched operand with an output row offset and a K that ends in half a step (the Voicebox input projection, networks.py:196)
    S1, Nn, K, Bx = 1118, 1024, 1440, 2     # 22.5 steps of 64: the half step is the last one issued by a steady-state round
    A, W = _rand((Bx * S1, K), torch.bfloat16, 38, 0.3).to(dev), _rand((Nn, K), torch.bfloat16, 39, 0.3).to(dev)
    b = _rand((Nn,), torch.float32, 40).to(dev)

    def batched():
        o32 = torch.zeros((Bx * (S1 + 1), Nn), device=dev); o16 = torch.zeros((Bx * (S1 + 1), Nn), device=dev, dtype=torch.bfloat16)
        ops.gemm(A, W, M=S1, N=Nn, Kc=K, lda=K, rowsA=S1, batch=Bx, a_bstride=S1 * K, c_bstride=S1 + 1, c_row_off=1, bias=b,
                 out32=o32, out16=o16, ldc=Nn)
        return o32, o16

    for x, y in zip(run(4, batched), run(tile, batched)):
        assert torch.equal(x, y), f"batched: tile {tile} differs from tile 4 ({(x != y).float().mean().item():.4f} of the elements)"
    ref = (A.double().cpu().reshape(Bx, S1, K) @ W.double().cpu().T + b.double().cpu())
    _check(run(tile, batched)[0].reshape(Bx, S1 + 1, Nn)[:, 1:], ref, torch.bfloat16, K, "batched vs reference")

    # two sources concatenated along K (the Voicebox skip Linear, networks.py:364): taps = 2, a_tap_stride, no row shift
    Mm, Hh2 = 600, 256
    buf = _rand((3, Mm, Hh2), torch.bfloat16, 43, 0.3).to(dev)
    W2 = _rand((Hh2, 2 * Hh2), torch.bfloat16, 44, 0.3).to(dev)

    def two_source():
        o32 = torch.zeros((Mm, Hh2), device=dev); o16 = torch.zeros((Mm, Hh2), device=dev, dtype=torch.bfloat16)
        ops.gemm(buf, W2, M=Mm, N=Hh2, Kc=Hh2, taps=2, lda=Hh2, rowsA=Mm, a_tap_stride=2 * Mm * Hh2, bias=b[:Hh2].contiguous(), out32=o32, out16=o16)
        return o32, o16

    for x, y in zip(run(4, two_source), run(tile, two_source)):    # (4 is routed to the register-staged tile for multi-tap operands)
        assert torch.equal(x, y), f"two-source K: tile {tile} differs"
    ref = torch.cat([buf[0].cpu(), buf[2].cpu()], -1).double() @ W2.double().cpu().T + b[:Hh2].double().cpu()
    _check(run(tile, two_source)[0], ref, torch.bfloat16, 2 * Hh2, "two-source K vs reference")

    # head-split epilogue: two sequences whose boundary falls inside a tile at an even and at an odd position
    for S in (1118, 333):
        B, Hh, D, K = 2, 4, 64, 128
        Spad = (S + 63) // 64 * 64
        A, W = _rand((B * S, K), torch.bfloat16, 35).to(dev), _rand((3 * Hh * D, K), torch.bfloat16, 36, 0.2).to(dev)
        b = _rand((3 * Hh * D,), torch.float32, 37).to(dev)

        def qkv():
            q = torch.zeros((B, Hh, Spad, D), device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q)
            v = torch.zeros((B, Hh, D, Spad), device=dev, dtype=torch.bfloat16)
            ops.gemm(A, W, M=B * S, N=3 * Hh * D, Kc=K, bias=b, qkv=dict(S=S, Spad=Spad, H=Hh, D=D, q=q, k=k, v=v))
            return q, k, v

        for x, y in zip(run(4, qkv), run(tile, qkv)):
            assert torch.equal(x, y), f"qkv S={S}: tile {tile} differs from tile 4"


@pytest.mark.parametrize("M", [16, 38, 100])
def test_short_prefill_tiles_bit_identical(dev, M, tile_env):
    """Round 4: short 7B prefills (the 33 - 100 new rows behind a reused prefix) pick the 64x64 tile with four DMA stages (8) and the
    128x128 ping-pong tile (14) instead of the 2-stage 64x64 tile (5).  The exact prefix reuse rests on a prefill row
    not depending on how many rows were prefilled with it, i.e. on EVERY tile accumulating K in the same order: the LLM epilogues
    (bf16-rounded residual add, SwiGLU pairs, plain bf16) must be bit-identical across the tiles a short and a long prefill use."""
    from usdm_amd import ops
    from usdm_amd._lib import ACT_SWIGLU
    bf = torch.bfloat16
    for (N, K) in [(768, 1088), (512, 2048), (1536, 1024)]:
        A, W = _rand((M, K), bf, 51, 0.3).to(dev), _rand((N, K), bf, 52, 0.3).to(dev)
        R = _rand((M, N), bf, 53).to(dev)

        def residual():
            o = torch.zeros((M, N), device=dev, dtype=bf)
            ops.gemm(A, W, M=M, N=N, Kc=K, residual=R, ldr=N, round_bf16=True, out16=o)
            return o

        def swiglu():
            o = torch.zeros((M, N // 2), device=dev, dtype=bf)
            ops.gemm(A, W, M=M, N=N, Kc=K, act=ACT_SWIGLU, round_bf16=True, out16=o, ldc=N // 2)
            return o

        def plain():
            o = torch.zeros((M, N), device=dev, dtype=bf)
            ops.gemm(A, W, M=M, N=N, Kc=K, out16=o)
            return o
        for name, f in (("residual", residual), ("swiglu", swiglu), ("plain", plain)):
            tile_env(4)
            ref = f()
            for t in (5, 8, 10, 14, 12):
                tile_env(t)
                assert torch.equal(ref, f()), f"{name} {M}x{N}x{K}: tile {t} differs from tile 4"
    os.environ.pop("USDM_GEMM_TILE", None)
    # and the automatic choice for the 7B's short-prefill shapes
    for (N, K, want) in [(6144, 4096, 8), (4096, 14336, 8), (28672, 4096, 14)]:
        A, W = torch.zeros(M, K, device=dev, dtype=bf), torch.zeros(8, K, device=dev, dtype=bf)      # (the query reads no memory)
        o = torch.zeros(M, 8, device=dev, dtype=bf)
        assert ops.gemm(A, W, M=M, N=N, Kc=K, out16=o, ldc=N, tile_query=True) == want


def test_pingpong_selected_for_big_linear(dev):
    """The launcher's own choice on the Voicebox / LLM-prefill shapes is the ping-pong tile (kernel name in the plan's record is
    not visible from here, so check through the override-free result being bit-identical AND the documented rule's inputs)."""
    from usdm_amd import ops
    M, N, K = 2236, 4096, 1024
    A, W = _rand((M, K), torch.bfloat16, 41, 0.3).to(dev), _rand((N, K), torch.bfloat16, 42, 0.3).to(dev)
    o = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    ops.gemm(A, W, M=M, N=N, Kc=K, out16=o)
    _check(o.float(), A.double().cpu() @ W.double().cpu().T, torch.bfloat16, K, "big linear")


@pytest.mark.parametrize("M", [2236, 1118, 300])
def test_folded_layernorm_producer_and_consumers(dev, M, tile_env):
    """LayerNorm folded into the neighbouring GEMM epilogues (usdm_gemm stats_out / ln_mode; the post-LN block of the reference,
    networks.py:236-266).  Producer: per-tile row sums / M2 equal those of the f32 output it stored.  Consumer 1 (GELU epilogue):
    GELU(LN(x) W0^T + b) from the un-normalised bf16 rows and gamma-folded weights.  Consumer 2: residual = LN(x) computed on the
    fly from the un-normalised f32 rows, also under split-K (only split 0 adds it).  All three on every ping-pong tile."""
    from usdm_amd import ops
    from usdm_amd._lib import ACT_GELU, UsdmError
    H, I = 1024, 512
    bf = torch.bfloat16
    for tile in (12, 13, 14):
        tile_env(tile)
        # ---- producer: x1 = A Wo^T + b + res, stats of x1
        A, Wo = _rand((M, H), bf, 41, 0.3).to(dev), _rand((H, H), bf, 42, 0.05).to(dev)
        b, res = _rand((H,), torch.float32, 43).to(dev), _rand((M, H), torch.float32, 44, 2.0).to(dev) + 0.7
        x32 = torch.zeros(M, H, device=dev); x16 = torch.zeros(M, H, device=dev, dtype=bf)
        nt = H // 128
        st = torch.full((M, nt, 2), float("nan"), device=dev)
        ops.gemm(A, Wo, M=M, N=H, Kc=H, bias=b, residual=res, ldr=H, out32=x32, out16=x16, stats_out=st)
        xs = x32.double().view(M, nt, 128)
        assert torch.allclose(st[:, :, 0].double(), xs.sum(-1), rtol=1e-5, atol=1e-3), f"tile {tile}: row sums"
        m2 = ((xs - xs.mean(-1, keepdim=True)) ** 2).sum(-1)      # M2 about the tile's own mean (merged pairwise by the consumers)
        assert torch.allclose(st[:, :, 1].double(), m2, rtol=1e-5, atol=1e-3), f"tile {tile}: per-tile M2"
        assert torch.equal(x16, x32.to(bf))
        # ---- consumer 1: GELU(LN(x1) W1^T + b1)
        gam, bet = (1 + 0.2 * _rand((H,), torch.float32, 45)).to(dev), (0.3 * _rand((H,), torch.float32, 46)).to(dev)
        W1, b1 = _rand((I, H), torch.float32, 47, 0.05).to(dev), _rand((I,), torch.float32, 48).to(dev)
        w1g = (W1 * gam[None]).to(bf).contiguous()
        c1, d1 = w1g.float().sum(1).contiguous(), (b1 + W1 @ bet).contiguous()
        f16 = torch.zeros(M, I, device=dev, dtype=bf)
        lnk = dict(stats=st, nt=nt, C=H, eps=1e-5)
        ops.gemm(x16, w1g, M=M, N=I, Kc=H, bias=d1, act=ACT_GELU, out16=f16, ln=dict(mode=1, c=c1, **lnk))
        ln = torch.nn.functional.layer_norm(x32.double(), (H,), gam.double(), bet.double(), 1e-5)
        ref = torch.nn.functional.gelu(ln @ W1.double().T + b1.double())
        err = (f16.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 2e-2, f"tile {tile}: folded LN + GELU rel err {err}"
        # ---- consumer 2: y = F W2^T + b2 + LN(x1), split-K 3 (bias and residual applied once)
        F_, W2, b2 = _rand((M, I), bf, 49, 0.3).to(dev), _rand((H, I), bf, 50, 0.05).to(dev), _rand((H,), torch.float32, 51).to(dev)
        for S in (1, 3):
            parts = torch.full((max(S, 1), M, H), float("nan"), device=dev)
            ops.gemm(F_, W2, M=M, N=H, Kc=I, bias=b2, residual=x32, ldr=H, out32=parts, split_k=S if S > 1 else 0, c_split_stride=M * H,
                     ln=dict(mode=2, gamma=gam, beta=bet, **lnk))
            ref2 = F_.double() @ W2.double().T + b2.double() + ln
            err2 = (parts.sum(0).double() - ref2).abs().max().item() / ref2.abs().max().item()
            assert err2 <= 1e-4, f"tile {tile}, split {S}: folded-LN residual rel err {err2}"
    # a tile without the folded epilogues must refuse, not silently ignore
    tile_env(5)
    with pytest.raises(UsdmError):
        ops.gemm(A, Wo, M=M, N=H, Kc=H, bias=b, out32=x32, stats_out=st)
