"""Thin per-kernel wrappers over the C-ABI (torch tensors in, device pointers out).

These exist for the parity tests and for the Python drop-ins; they add no arithmetic.
"""
import ctypes as C
import ctypes as C_
import os

import torch

from . import _lib
from ._lib import (BF16, F32, AttnArgs, AttnDecodeArgs, DecodeState, GemmArgs, GemvArgs, GemvBatchArgs, NormArgs, RopeArgs, SampleArgs, SnakeArgs,
                   VbInputArgs, VbSolverArgs, check, lib)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """A recorded sequence of C-ABI launches with their argument structs pre-built.

    Python builds the structs once per shape; run() is then a tight loop of ctypes calls on the
    current HIP stream (and is what gets captured into a hipGraph by usdm_amd.graph.GraphedPlan)."""

    def __init__(self):
        self.calls = []
        self.keep = []  # tensors that must outlive the plan (workspaces, packed weights)

    def add(self, what, fn, *args):
        self.calls.append((what, fn, args))

    def hold(self, *tensors):
        self.keep.extend(tensors)
        return tensors[0] if len(tensors) == 1 else tensors

    def run(self):
        st = _stream()
        for what, fn, args in self.calls:
            rc = fn(*args, st)
            if rc != 0:
                check(rc, what)

    def __len__(self):
        return len(self.calls)


def _go(plan, what, fn, *args):
    if plan is not None:
        plan.add(what, fn, *args)
    else:
        check(fn(*args, _stream()), what)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.UsdmError("usdm_amd kernels run on the GPU only (no CPU fallback); got a CPU tensor")


def gemm(A, W, *, M, N, Kc, taps=1, lda=None, rowsA=None, a_row_mul=1, a_row_off=0, a_row_step=0,
         a_tap_stride=0, ldw=None, groups=1, batch=1, a_gstride=0, w_gstride=0, a_bstride=0, c_gcol=0,
         c_bstride=0, bias=None, alpha=1.0, act=0, round_bf16=False, residual=None, ldr=0,
         out32=None, out16=None, ldc=None, c_row_mul=1, c_row_off=0, transpose_out=False,
         qkv=None, split_k=0, c_split_stride=0, stats_out=None, ln=None, plan=None, tile_query=False):
    """Raw launch of usdm_gemm; see include/usdm_hip.h for the meaning of every field."""
    _need_cuda(A, W, bias, residual, out32, out16)
    a = GemmArgs()
    a.dtype = _dt(A)
    assert W.dtype == A.dtype
    a.M, a.N, a.taps, a.Kc = M, N, taps, Kc
    a.A, a.lda = _ptr(A), (lda if lda is not None else A.stride(-2))
    a.rowsA = rowsA if rowsA is not None else M
    a.a_row_mul, a.a_row_off, a.a_row_step, a.a_tap_stride = a_row_mul, a_row_off, a_row_step, a_tap_stride
    a.W, a.ldw = _ptr(W), (ldw if ldw is not None else taps * Kc)
    a.groups, a.batch = groups, batch
    a.a_gstride, a.w_gstride, a.a_bstride, a.c_gcol, a.c_bstride = a_gstride, w_gstride, a_bstride, c_gcol, c_bstride
    a.bias, a.alpha, a.act, a.round_bf16 = _ptr(bias), alpha, act, int(round_bf16)
    a.residual = _ptr(residual)
    a.res_dtype = _dt(residual) if residual is not None else F32
    a.ldr = ldr
    a.C32, a.C16 = _ptr(out32), _ptr(out16)
    a.ldc = ldc if ldc is not None else N
    a.c_row_mul, a.c_row_off, a.transpose_out = c_row_mul, c_row_off, int(transpose_out)
    a.split_k, a.c_split_stride = split_k, c_split_stride
    a.stats_out = _ptr(stats_out)
    if ln is not None:      # folded LayerNorm (include/usdm_hip.h): dict(mode, stats, nt, C, eps, c= | gamma=, beta=)
        _need_cuda(ln["stats"], ln.get("c"), ln.get("gamma"), ln.get("beta"))
        a.ln_stats, a.ln_nt, a.ln_mode, a.ln_C, a.ln_eps = _ptr(ln["stats"]), ln["nt"], ln["mode"], ln["C"], ln.get("eps", 1e-5)
        a.ln_c, a.ln_gamma, a.ln_beta = _ptr(ln.get("c")), _ptr(ln.get("gamma")), _ptr(ln.get("beta"))
        if ln.get("guard") is not None:      # device word OR-ed with 1 when a row's |mean| / sigma exceeds guard_ratio
            _need_cuda(ln["guard"])
            a.ln_guard, a.ln_guard_ratio = _ptr(ln["guard"]), float(ln["guard_ratio"])
    tile = os.environ.get("USDM_GEMM_TILE")      # benchmarks / tile-equivalence tests: the library itself reads no environment
    if tile is not None:
        a.tile_sel = int(tile) + 1
    if qkv is not None:
        a.epi = _lib.EPI_QKV_HEADS
        a.qkv_S, a.qkv_Spad, a.qkv_H, a.qkv_D = qkv["S"], qkv["Spad"], qkv["H"], qkv["D"]
        a.qkv_q, a.qkv_k, a.qkv_v = _ptr(qkv["q"]), _ptr(qkv["k"]), _ptr(qkv["v"])
    if tile_query:
        return lib.usdm_gemm_tile_for(C.byref(a))
    _go(plan, "usdm_gemm", lib.usdm_gemm, C.byref(a))


def norm(x, gamma, beta=None, *, rows, C, eps=1e-5, res=None, rms=False, act=0, round_bf16=False, premask=False,
         valid_len=None, rows_per_batch=0, out32=None, out16=None, sum32=None, sum16=None,
         ldx=None, ldr=None, ldo=None, lds=None, res2=None, n_res2=0, res2_stride=0, plan=None):
    """usdm_norm: LayerNorm/RMSNorm over the last axis (see include/usdm_hip.h)."""
    _need_cuda(x, gamma, beta, res, out32, out16, sum32, sum16, valid_len)
    a = NormArgs()
    a.x, a.x_dtype, a.ldx = _ptr(x), _dt(x), (ldx if ldx is not None else C)
    a.res, a.res_dtype, a.ldr = _ptr(res), (_dt(res) if res is not None else F32), (ldr if ldr is not None else C)
    a.gamma, a.beta, a.eps = _ptr(gamma), _ptr(beta), eps
    a.rows, a.C = rows, C
    a.rms, a.act, a.round_bf16, a.premask = int(rms), act, int(round_bf16), int(premask)
    a.valid_len, a.rows_per_batch = _ptr(valid_len), rows_per_batch
    a.out32, a.out16, a.ldo = _ptr(out32), _ptr(out16), (ldo if ldo is not None else C)
    a.sum32, a.sum16, a.lds = _ptr(sum32), _ptr(sum16), (lds if lds is not None else C)
    if res2 is not None:     # further f32 addends (split-K partials): res2[i] for i < n_res2, res2_stride elements apart
        _need_cuda(res2)
        a.res2, a.n_res2, a.res2_stride = _ptr(res2), (n_res2 or 1), res2_stride
    _go(plan, "usdm_norm", lib.usdm_norm, C_.byref(a))


def aa_snake(x, alpha, beta, fup, fdn, *, T, C, Creal=None, logscale=True, out32=None, out16=None, ldx=None, ldo=None, L=0, plan=None):
    """usdm_aa_snake: fused Activation1d(SnakeBeta) on channels-last f32 [T][C]."""
    _need_cuda(x, alpha, beta, out32, out16)
    a = SnakeArgs()
    a.x, a.ldx = _ptr(x), (ldx if ldx is not None else C)
    a.T, a.C, a.Creal, a.L = T, C, (Creal if Creal is not None else C), L
    a.alpha, a.beta, a.logscale = _ptr(alpha), _ptr(beta), int(logscale)
    for j in range(12):
        a.fup[j] = float(fup[j])
        a.fdn[j] = float(fdn[j])
    a.out32, a.out16, a.ldo = _ptr(out32), _ptr(out16), (ldo if ldo is not None else C)
    _go(plan, "usdm_aa_snake", lib.usdm_aa_snake, C_.byref(a))


def attention(q, k, vt, o, *, mode, dh, B, Hq, Hkv, Sq, Skv, Skv_alloc, q_strides, k_strides, v_strides, o_strides,
              scale=1.0, q_pos0=0, kv_len=None, slopes=None, alibi_col0_zero=True, window=0, plan=None):
    """usdm_attention (see include/usdm_hip.h for layouts)."""
    _need_cuda(q, k, vt, o, kv_len, slopes)
    a = AttnArgs()
    a.mode, a.dh, a.B, a.Hq, a.Hkv, a.Sq, a.Skv, a.Skv_alloc = mode, dh, B, Hq, Hkv, Sq, Skv, Skv_alloc
    a.q_pos0, a.alibi_col0_zero, a.scale = q_pos0, int(alibi_col0_zero), scale
    a.q, (a.q_bs, a.q_hs, a.q_rs) = _ptr(q), q_strides
    a.k, (a.k_bs, a.k_hs, a.k_rs) = _ptr(k), k_strides
    a.vt, (a.v_bs, a.v_hs, a.v_ds) = _ptr(vt), v_strides
    a.o, (a.o_bs, a.o_rs) = _ptr(o), o_strides
    a.kv_len, a.slopes = _ptr(kv_len), _ptr(slopes)
    a.window = int(window)
    a.variant = 1 if os.environ.get("USDM_ATTN_V16", "1") == "0" else 0      # (tools/attn_bench.py: the 32-query-wave kernel)
    if os.environ.get("USDM_ATTN_ORDER") == "0":
        a.head_order = -1
    _go(plan, "usdm_attention", lib.usdm_attention, C_.byref(a))


def sum3_scale(a, b, c, scale, *, out32=None, out16=None, plan=None):
    _need_cuda(a, b, c, out32, out16)
    n = a.numel()
    _go(plan, "usdm_sum3_scale", lib.usdm_sum3_scale, _ptr(a), _ptr(b), _ptr(c), C.c_float(scale), C.c_int64(n),
        _ptr(out32), _ptr(out16))


def cf_to_cl(x, *, B, C, T, Cpad, scale=1.0, shift=0.0, out32=None, out16=None, plan=None):
    """channels-first f32 [B][C][T] -> channels-last [B][T][Cpad]."""
    _need_cuda(x, out32, out16)
    _go(plan, "usdm_cf_to_cl", lib.usdm_cf_to_cl, _ptr(x), C_.c_int32(B), C_.c_int32(C), C_.c_int32(T), C_.c_int32(Cpad),
        C_.c_float(scale), C_.c_float(shift), _ptr(out32), _ptr(out16))


def vb_build_input(ids, y, cond, table, out, *, B_in, dup, S, E, F, null_id, use_cond, ldo, plan=None):
    _need_cuda(ids, y, cond, table, out)
    a = VbInputArgs()
    a.ids, a.y, a.cond, a.table = _ptr(ids), _ptr(y), _ptr(cond), _ptr(table)
    a.B_in, a.dup, a.S, a.E, a.F, a.null_id, a.use_cond = B_in, dup, S, E, F, null_id, int(use_cond)
    a.out, a.ldo, a.out_dtype = _ptr(out), ldo, _dt(out)
    assert table.dtype == out.dtype
    _go(plan, "usdm_vb_build_input", lib.usdm_vb_build_input, C_.byref(a))


def softmax_alibi(x, *, rows, rows_per_batch, nheads, n, npad, ldrow, ldseg, slopes=None, kv_len=None, col0_zero=True, plan=None):
    _need_cuda(x, slopes, kv_len)
    _go(plan, "usdm_softmax_alibi", lib.usdm_softmax_alibi, _ptr(x), C_.c_int32(rows), C_.c_int32(rows_per_batch), C_.c_int32(nheads),
        C_.c_int32(n), C_.c_int32(npad), C_.c_int64(ldrow), C_.c_int32(ldseg), _ptr(slopes), _ptr(kv_len), C_.c_int32(int(col0_zero)))


def vb_time_token(t, freqs, h32, h16, *, Bx, H, rows_per_batch, t_stride=1, plan=None):
    _need_cuda(t, freqs, h32, h16)
    _go(plan, "usdm_vb_time_token", lib.usdm_vb_time_token, _ptr(t), C_.c_int32(t_stride), _ptr(freqs), C_.c_int32(Bx),
        C_.c_int32(H), C_.c_int64(rows_per_batch), _ptr(h32), _ptr(h16))


def vb_solver_step(vout, z, *, B, F, S, mode, dt, cfg=False, gs=0.0, v1=None, eps=None, cond=None, P=0, c_eps=0.0,
                   c_cond=0.0, z_in=None, z_commit=None, t_cur=None, t_count=0, t_next=0.0, plan=None):
    _need_cuda(vout, z, v1, eps, cond, z_in, z_commit, t_cur)
    a = VbSolverArgs()
    a.vout, a.z, a.v1, a.eps, a.cond = _ptr(vout), _ptr(z), _ptr(v1), _ptr(eps), _ptr(cond)
    a.z_in, a.z_commit, a.t_cur = _ptr(z_in), _ptr(z_commit), _ptr(t_cur)
    a.B, a.F, a.S, a.P, a.cfg, a.mode, a.t_count = B, F, S, P, int(cfg), mode, t_count
    a.gs, a.dt, a.c_eps, a.c_cond, a.t_next = gs, dt, c_eps, c_cond, t_next
    _go(plan, "usdm_vb_solver_step", lib.usdm_vb_solver_step, C_.byref(a))


def copy_bytes(dst, src, nbytes, plan=None):
    _need_cuda(dst, src)
    _go(plan, "usdm_copy_bytes", lib.usdm_copy_bytes, _ptr(dst), _ptr(src), C_.c_int64(nbytes))


def process_unit(units, rep, hop):
    """usdm_process_unit: int64 [n] on the GPU -> int64 [floor(n*rep/hop)]."""
    _need_cuda(units)
    if units.dtype != torch.int64 or units.dim() != 1:
        raise TypeError("units must be a 1-D int64 tensor")
    n = units.numel()
    nframes = (n * rep) // hop
    out = torch.empty(nframes, dtype=torch.int64, device=units.device)
    if n == 0 or nframes == 0:
        return out
    check(lib.usdm_process_unit(_ptr(units.contiguous()), C_.c_int32(n), C_.c_int32(rep), C_.c_int32(hop), _ptr(out),
                                C_.c_int32(nframes), _stream()), "usdm_process_unit")
    return out


def gemv(W, x, *, N, K, ldw=None, norm_w=None, eps=1e-5, act=0, round_bf16=True, residual=None, y16=None, y32=None,
         ban=None, part_val=None, part_idx=None, idx_offset=0, x_delta=None, x_out=None, skip=None, p2p=None, p2p_site=0,
         p2p_mode=0, merge=None, cmb=None, plan=None, only_args=False):
    """usdm_gemv: batch-1 weight-streaming GEMV (see include/usdm_hip.h).  p2p: a usdm_amd.p2p.P2PComm (fused all-reduce).
    only_args=True: return the filled usdm_gemv_args instead of launching (a phase of usdm_gemv_chain)."""
    _need_cuda(W, x, norm_w, residual, y16, y32, ban, part_val, part_idx, x_delta, x_out, skip)
    if x_out is not None and x_out.data_ptr() == x.data_ptr():
        raise ValueError("usdm_gemv: x_out must not alias x")
    a = GemvArgs()
    a.W, a.ldw, a.N, a.K = _ptr(W), (ldw if ldw is not None else K), N, K
    a.x, a.norm_w, a.eps = _ptr(x), _ptr(norm_w), eps
    a.act, a.round_bf16 = act, int(round_bf16)
    a.residual, a.y16, a.y32 = _ptr(residual), _ptr(y16), _ptr(y32)
    a.ban, a.part_val, a.part_idx, a.idx_offset = _ptr(ban), _ptr(part_val), _ptr(part_idx), idx_offset
    a.x_delta, a.x_out, a.skip = _ptr(x_delta), _ptr(x_out), _ptr(skip)
    if merge is not None:       # (pm, pl, po, NS): x is merged from the decode-attention partials in the prologue
        pm, pl, po, ns = merge
        _need_cuda(pm, pl, po)
        a.mrg_pm, a.mrg_pl, a.mrg_po, a.mrg_ns = _ptr(pm), _ptr(pl), _ptr(po), ns
    if cmb is not None:         # (granules int64 [K/2], err int32 [1]): the hand-off form of `merge` (one combine per head, see usdm_hip.h)
        gran, err = cmb
        _need_cuda(gran, err)
        if merge is None or gran.numel() * gran.element_size() < (K // 2) * 8:
            raise ValueError("usdm_gemv: cmb needs merge=(pm, pl, po, NS) and K/2 8-byte granules")
        a.cmb_gran, a.cmb_err, a.cmb_timeout_ms = _ptr(gran), _ptr(err), 200
    if p2p is not None and p2p_mode:
        p2p.check_site(p2p_site, N)
        a.p2p, a.p2p_site, a.p2p_mode = p2p.dev_ptr, p2p_site, p2p_mode
    if only_args:
        return a
    _go(plan, "usdm_gemv", lib.usdm_gemv, C_.byref(a))


def gemv_engine(phases, sync, gran, timeout_ms=2000, plan=None):
    """usdm_gemv_engine: the chained projections on the loader / consumer engine (LDS-DMA weight ring, granule hand-offs).
    gran: >= 24576 int64 words of device scratch."""
    _need_cuda(sync, gran)
    if not (1 <= len(phases) <= 4) or sync.numel() < 8 or sync.element_size() != 4 or gran.numel() * gran.element_size() < 3 * 8192 * 8:
        raise ValueError("gemv_engine: 1..4 phases, an 8-word sync block and 192 KB of granule space")
    c = _lib.GemvChainArgs()
    for i, ph in enumerate(phases):
        c.ph[i] = ph
    c.nph, c.sync, c.timeout_ms, c.gran = len(phases), _ptr(sync), timeout_ms, _ptr(gran)
    _go(plan, "usdm_gemv_engine", _lib.exp().usdm_gemv_engine, C_.byref(c))


def gemv_chain(phases, sync, timeout_ms=2000, plan=None):
    """usdm_gemv_chain: up to 4 consecutive decode projections in one persistent launch.  phases: usdm_gemv_args from
    gemv(..., only_args=True); sync: int32/uint32 device tensor of >= 8 words, zero-initialised once by the caller."""
    _need_cuda(sync)
    if not (1 <= len(phases) <= 4) or sync.numel() < 8 or sync.element_size() != 4:
        raise ValueError("gemv_chain: 1..4 phases and an 8-word sync block")
    c = _lib.GemvChainArgs()
    for i, ph in enumerate(phases):
        c.ph[i] = ph
    c.nph, c.sync, c.timeout_ms = len(phases), _ptr(sync), timeout_ms
    _go(plan, "usdm_gemv_chain", _lib.exp().usdm_gemv_chain, C_.byref(c))


def p2p_reduce(p2p, site, n, h, skip=None, plan=None):
    """usdm_allreduce_p2p_reduce: second half of the split form: h = bf16(h + bf16(sum over ranks of slot[site]))."""
    _need_cuda(h, skip)
    p2p.check_site(site, n)
    _go(plan, "usdm_allreduce_p2p_reduce", lib.usdm_allreduce_p2p_reduce, C_.c_void_p(p2p.dev_ptr), C_.c_int32(site), C_.c_int32(n),
        _ptr(h), _ptr(skip))


def argmax_p2p(part_val, part_idx, nparts, st, p2p, site, phase=0, embed=None, h_out=None, Hd=0, plan=None):
    """usdm_argmax_p2p: vocab-parallel token pick across ranks + decode-state update + epoch advance."""
    _need_cuda(part_val, part_idx, embed, h_out)
    p2p.check_site(site, 2)
    _go(plan, "usdm_argmax_p2p", lib.usdm_argmax_p2p, _ptr(part_val), _ptr(part_idx), C_.c_int32(nparts), C_.byref(st),
        C_.c_void_p(p2p.dev_ptr), C_.c_int32(site), C_.c_int32(phase), _ptr(embed), C_.c_int32(Hd), _ptr(h_out))


def decode_state(next_token, out_tokens, step, pos, *, id_offset=0, advance_pos=True, batch=0, done=None, eos=None):
    """batch > 1: next_token / step / pos are [batch] and out_tokens is [batch][max_out].
    done [1] / eos [8] = {n_eos, min_new, ids...}: device words of the optional device-side end of sequence."""
    st = DecodeState()
    st.next_token, st.out_tokens, st.step, st.pos = _ptr(next_token), _ptr(out_tokens), _ptr(step), _ptr(pos)
    st.max_out = out_tokens.shape[-1] if batch > 1 else out_tokens.numel()
    st.id_offset, st.advance_pos, st.batch = id_offset, int(advance_pos), batch
    st.done, st.eos = _ptr(done), _ptr(eos)
    return st


def argmax_final(part_val, part_idx, nparts, st, embed=None, h_out=None, Hd=0, nseg=1, seg_stride=0, plan=None):
    _need_cuda(part_val, part_idx, embed, h_out)
    if nseg > 1:      # [segment (rank)][sequence][nparts]: the gathered partials of a tensor-parallel batched step
        _go(plan, "usdm_argmax_final_seg", lib.usdm_argmax_final_seg, _ptr(part_val), _ptr(part_idx), C_.c_int32(nparts), C_.c_int32(nseg),
            C_.c_int64(seg_stride), C_.byref(st), _ptr(embed), C_.c_int32(Hd), _ptr(h_out))
        return
    _go(plan, "usdm_argmax_final", lib.usdm_argmax_final, _ptr(part_val), _ptr(part_idx), C_.c_int32(nparts), C_.byref(st),
        _ptr(embed), C_.c_int32(Hd), _ptr(h_out))


def sample_params_tensor(device, n=1):
    """Device block holding n usdm_sample_params (24 bytes each: temperature, top_k, top_p, reserved, seed)."""
    sz = C.sizeof(_lib.SampleParams)
    return torch.zeros(sz if n == 1 else (n, sz), dtype=torch.uint8, device=device)


def set_sample_params(t, temperature, top_k, top_p, seed):
    """t: one 24-byte block (a row of sample_params_tensor(device, n) for slot b of a batch)"""
    p = _lib.SampleParams(float(temperature), int(top_k), float(top_p), 0, int(seed) & 0xFFFFFFFFFFFFFFFF)
    t.copy_(torch.frombuffer(bytearray(bytes(p)), dtype=torch.uint8))


def sample_final(logits, st, *, temperature=1.0, top_k=0, top_p=1.0, seed=0, probs_out=None, embed=None, h_out=None, Hd=0,
                 dev_params=None, plan=None):
    """usdm_sample_final: temperature / top-k / top-p sampling of one token from ban-masked f32 logits.
    dev_params (sample_params_tensor): the knobs are read from device memory instead (graph-replayable per request).
    Batched state (decode_state(batch=B)): logits [B][V], dev_params [B][24], h_out [B][Hd]; one workgroup per sequence."""
    _need_cuda(logits, probs_out, embed, h_out, dev_params)
    a = SampleArgs()
    a.logits, a.V, a.temperature, a.top_k, a.top_p = _ptr(logits), logits.shape[-1], temperature, top_k, top_p
    a.logits_bs = logits.stride(0) if logits.dim() == 2 else logits.numel()
    a.seed, a.probs_out, a.dev_params = seed, _ptr(probs_out), _ptr(dev_params)
    _go(plan, "usdm_sample_final", lib.usdm_sample_final, C_.byref(a), C_.byref(st), _ptr(embed), C_.c_int32(Hd), _ptr(h_out))


def embed_rows(table, out, *, Hd, ids=None, next_token=None, n=1, plan=None):
    _need_cuda(table, out, ids, next_token)
    _go(plan, "usdm_embed_rows", lib.usdm_embed_rows, _ptr(table), _ptr(ids), _ptr(next_token), C_.c_int32(n), C_.c_int32(Hd), _ptr(out))


def rope_cache(qkv, cos, sin, kcache, vcache, *, ld, S, pos0, Hq, Hkv, ctx_max, max_pos, vt=None, vt_ld=0, plan=None):
    _need_cuda(qkv, cos, sin, kcache, vcache, vt)
    a = RopeArgs()
    a.qkv, a.ld, a.S, a.pos0, a.Hq, a.Hkv, a.ctx_max, a.max_pos = _ptr(qkv), ld, S, pos0, Hq, Hkv, ctx_max, max_pos
    a.cos, a.sin, a.kcache, a.vcache, a.vt, a.vt_ld = _ptr(cos), _ptr(sin), _ptr(kcache), _ptr(vcache), _ptr(vt), vt_ld
    _go(plan, "usdm_rope_cache", lib.usdm_rope_cache, C_.byref(a))


def gemv_batch(W, x, *, nb, N, K, x_bs, y_bs=0, res_bs=0, part_bs=0, ldw=None, norm_w=None, eps=1e-5, act=0, round_bf16=True,
               residual=None, y16=None, y32=None, ban=None, part_val=None, part_idx=None, idx_offset=0, form=0, ks=None, plan=None):
    """usdm_gemv_batch: the decode projection over nb <= 16 input vectors (x is [nb][x_bs], outputs [nb][y_bs]).
    form 0: VALU kernel for nb <= 4, matrix-core kernel above; 1: matrix cores; -1: VALU; 3 / 5: A/B forms of the matrix-core kernel.
    ks = (part f32 [gemv_batch_ks_floats(N, K)], counters int32 [ceil(N / 16)], zero): K split over workgroups (K > 4096)."""
    _need_cuda(W, x, norm_w, residual, y16, y32, ban, part_val, part_idx)
    b = GemvBatchArgs()
    a = b.g
    a.W, a.ldw, a.N, a.K = _ptr(W), (ldw if ldw is not None else K), N, K
    a.x, a.norm_w, a.eps = _ptr(x), _ptr(norm_w), eps
    a.act, a.round_bf16 = act, int(round_bf16)
    a.residual, a.y16, a.y32 = _ptr(residual), _ptr(y16), _ptr(y32)
    a.ban, a.part_val, a.part_idx, a.idx_offset = _ptr(ban), _ptr(part_val), _ptr(part_idx), idx_offset
    b.nb, b.x_bs, b.y_bs, b.res_bs, b.part_bs, b.form = nb, x_bs, y_bs, res_bs, part_bs, form
    if ks is not None:
        part, cnt = ks
        _need_cuda(part, cnt)
        if part.dtype != torch.float32 or cnt.dtype != torch.int32 or cnt.numel() < -(-N // 16):
            raise ValueError("usdm_gemv_batch: ks = (float32 partials, int32 counters [ceil(N / 16)])")
        b.ks_part, b.ks_cnt, b.ks_part_floats = _ptr(part), _ptr(cnt), part.numel()
    _go(plan, "usdm_gemv_batch", lib.usdm_gemv_batch, C_.byref(b))


def gemv_batch_ks_floats(N, K):
    """floats of ks_part usdm_gemv_batch wants for this shape (0: the shape is not split over workgroups)"""
    return int(lib.usdm_gemv_batch_ks_floats(C_.c_int32(N), C_.c_int32(K)))


def attn_decode(qkv, pos, cos, sin, kcache, vcache, pm, pl, po, out, *, Hq, Hkv, ctx_max, NS, scale, counters=None, batch=0,
                qkv_bs=0, out_bs=0, cache_bs=0, skip=None, defer_merge=False, window=0, cmb_gran=None, plan=None):
    _need_cuda(qkv, pos, cos, sin, kcache, vcache, pm, pl, po, out, counters)
    a = AttnDecodeArgs()
    a.qkv, a.pos, a.Hq, a.Hkv, a.ctx_max, a.NS, a.scale = _ptr(qkv), _ptr(pos), Hq, Hkv, ctx_max, NS, scale
    a.cos, a.sin, a.kcache, a.vcache = _ptr(cos), _ptr(sin), _ptr(kcache), _ptr(vcache)
    a.pm, a.pl, a.po, a.out, a.counters = _ptr(pm), _ptr(pl), _ptr(po), _ptr(out), _ptr(counters)
    a.batch, a.qkv_bs, a.out_bs, a.cache_bs, a.skip = batch, qkv_bs, out_bs, cache_bs, _ptr(skip)
    a.defer_merge = int(defer_merge)
    a.window = int(window)
    if cmb_gran is not None:
        _need_cuda(cmb_gran)
        if cmb_gran.numel() * cmb_gran.element_size() < Hq * 64 * 8:
            raise ValueError("usdm_attn_decode: cmb_gran holds Hq*64 8-byte granules")
        a.cmb_gran = _ptr(cmb_gran)
    _go(plan, "usdm_attn_decode", lib.usdm_attn_decode, C_.byref(a))


def residual_add(h, delta, n, plan=None):
    _need_cuda(h, delta)
    _go(plan, "usdm_residual_add", lib.usdm_residual_add, _ptr(h), _ptr(delta), C_.c_int32(n))


def gemv_nblocks(N, act=0):
    return lib.usdm_gemv_nblocks(C_.c_int32(N), C_.c_int32(act))


def wave_layernorm(x, y, n, eps=1e-5, plan=None):
    _need_cuda(x, y)
    _go(plan, "usdm_wave_layernorm", lib.usdm_wave_layernorm, _ptr(x), C_.c_int32(n), C_.c_float(eps), _ptr(y))


def w2v_conv0(x, w, b, g, be, out, *, n, T, C, k, stride, eps=1e-5, plan=None):
    _need_cuda(x, w, b, g, be, out)
    _go(plan, "usdm_w2v_conv0", lib.usdm_w2v_conv0, _ptr(x), C_.c_int32(n), C_.c_int32(T), C_.c_int32(C), C_.c_int32(k),
        C_.c_int32(stride), _ptr(w), _ptr(b), _ptr(g), _ptr(be), C_.c_float(eps), _ptr(out))


def softmax_segments(x, *, rows, nseg, n, npad, ldrow, ldseg, plan=None):
    _need_cuda(x)
    _go(plan, "usdm_softmax_segments", lib.usdm_softmax_segments, _ptr(x), C_.c_int32(rows), C_.c_int32(nseg), C_.c_int32(n),
        C_.c_int32(npad), C_.c_int64(ldrow), C_.c_int32(ldseg))


def kmeans_argmin(x, dots, csq, ids, *, T, D, n_units, ldd, margin=None, plan=None):
    _need_cuda(x, dots, csq, ids, margin)
    _go(plan, "usdm_kmeans_argmin", lib.usdm_kmeans_argmin, _ptr(x), C_.c_int32(T), C_.c_int32(D), _ptr(dots), C_.c_int64(ldd),
        _ptr(csq), C_.c_int32(n_units), _ptr(ids), _ptr(margin))


def stft_frames(x, window, frames, *, n, n_fft, hop, pad, T, plan=None):
    _need_cuda(x, window, frames)
    _go(plan, "usdm_stft_frames", lib.usdm_stft_frames, _ptr(x), C_.c_int32(n), C_.c_int32(n_fft), C_.c_int32(hop), C_.c_int32(pad),
        _ptr(window), _ptr(frames), C_.c_int32(T))


def stft_mag(re_im, out, *, ld, T, nbins, eps, ldo, nbins_pad, plan=None):
    _need_cuda(re_im, out)
    _go(plan, "usdm_stft_mag", lib.usdm_stft_mag, _ptr(re_im), C_.c_int64(ld), C_.c_int32(T), C_.c_int32(nbins), C_.c_float(eps),
        _ptr(out), C_.c_int64(ldo), C_.c_int32(nbins_pad))


def frame_signal(x, frames, *, n, frame_len, hop, offset, T, plan=None):
    _need_cuda(x, frames)
    _go(plan, "usdm_frame_signal", lib.usdm_frame_signal, _ptr(x), C_.c_int32(n), C_.c_int32(frame_len), C_.c_int32(hop),
        C_.c_int32(offset), _ptr(frames), C_.c_int32(T))


def mask_time(valid_len, *, B, T, C, layout, off=0, x32=None, x16=None, plan=None):
    _need_cuda(valid_len, x32, x16)
    _go(plan, "usdm_mask_time", lib.usdm_mask_time, _ptr(x32), _ptr(x16), C_.c_int32(B), C_.c_int32(T), C_.c_int32(C),
        C_.c_int32(layout), _ptr(valid_len), C_.c_int32(off))
