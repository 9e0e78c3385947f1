"""CPU, world_size 2 over gloo: the tensor-parallel algebra used by usdm_amd.llm (shard_weights + f32 partial
sums + all_reduce + vocab-parallel arg-max gather) reproduces the unsharded computation."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mistral_oracle as MO
    from usdm_amd.llm import shard_weights
    cfg = dict(vocab_size=203, hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
               num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_theta=10000.0)
    sd = MO.random_state_dict(cfg, seed=7, dtype=torch.float32)
    W = shard_weights(lambda n: sd[n], cfg, rank, world, "cpu", dtype=torch.float32)
    lw = W["layers"][0]
    x = torch.randn(5, 256, generator=torch.Generator().manual_seed(1))
    d, Hq, Hkv, I = 64, 4 // world, 2 // world, 512 // world
    # attention block: local heads, o_proj partial sums all-reduced
    qkv = x @ lw["qkv"].T
    q, k, v = qkv[:, :Hq * d], qkv[:, Hq * d:(Hq + Hkv) * d], qkv[:, (Hq + Hkv) * d:]
    sh = lambda t, h: t.view(5, h, d).transpose(0, 1)
    att = torch.softmax(sh(q, Hq) @ sh(k, Hkv).repeat_interleave(Hq // Hkv, 0).transpose(1, 2) * d ** -0.5, -1)
    o = (att @ sh(v, Hkv).repeat_interleave(Hq // Hkv, 0)).transpose(0, 1).reshape(5, Hq * d)
    part = o @ lw["o"].T
    dist.all_reduce(part)
    # reference (unsharded)
    p = "model.layers.0."
    qf, kf, vf = x @ sd[p + "self_attn.q_proj.weight"].T, x @ sd[p + "self_attn.k_proj.weight"].T, x @ sd[p + "self_attn.v_proj.weight"].T
    attf = torch.softmax(sh(qf, 4) @ sh(kf, 2).repeat_interleave(2, 0).transpose(1, 2) * d ** -0.5, -1)
    of = (attf @ sh(vf, 2).repeat_interleave(2, 0)).transpose(0, 1).reshape(5, 256)
    ref = of @ sd[p + "self_attn.o_proj.weight"].T
    ok1 = torch.allclose(part, ref, atol=1e-4)
    # MLP block: packed gate/up rows (blocks of 16+16), down partial sums
    gu = x @ lw["gu"].T
    gu = gu.view(5, I // 16, 2, 16)
    act = (torch.nn.functional.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(5, I)
    part = act @ lw["down"].T
    dist.all_reduce(part)
    ref = (torch.nn.functional.silu(x @ sd[p + "mlp.gate_proj.weight"].T) * (x @ sd[p + "mlp.up_proj.weight"].T)) @ sd[p + "mlp.down_proj.weight"].T
    ok2 = torch.allclose(part, ref, atol=1e-4)
    # vocab-parallel arg-max: local (max, global id) gathered, ties -> lowest id
    logits = x[0] @ W["lm_head"].T
    loc = torch.stack([logits.max(), (logits.argmax() + W["v0"]).float()])
    allv = [torch.zeros(2) for _ in range(world)]
    dist.all_gather(allv, loc)
    best = max(allv, key=lambda t: (t[0].item(), -t[1].item()))
    ok3 = int(best[1].item()) == int((x[0] @ sd["lm_head.weight"].T).argmax())
    out_q.put((rank, bool(ok1), bool(ok2), bool(ok3)))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_tp2_gloo_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(ok1 and ok2 and ok3 for _, ok1, ok2, ok3 in res), res


def _emulate_lm_head_partials(logits_loc, v0, nparts):
    """What usdm_gemv's lm_head mode leaves in this rank's partial buffers: one (max, global id) per block of 16 rows,
    ties -> lowest id; slots past the shard's last block keep the (-inf, 0x7fffffff) fill (llm.py _alloc)."""
    from usdm_amd.llm import NO_CANDIDATE_IDX
    pv = torch.full((nparts,), float("-inf"))
    pi = torch.full((nparts,), NO_CANDIDATE_IDX, dtype=torch.int32)
    n = logits_loc.numel()
    for b in range((n + 15) // 16):
        blk = logits_loc[16 * b:16 * b + 16]
        j = int(torch.argmax(blk))            # first maximum = lowest id
        pv[b], pi[b] = blk[j], v0 + 16 * b + j
    return pv, pi


def _final_argmax(pv, pi):
    """argmax_final_kernel's rule: largest value, ties -> lowest id."""
    best = max(range(pv.numel()), key=lambda i: (pv[i].item(), -int(pi[i])))
    return int(pi[best])


@pytest.mark.parametrize("tp", [2, 4, 8])
def test_vocab_parallel_partials_have_equal_size_on_every_rank(tp):
    """with V = 42 003 and tp = 8 the last rank owns 5246 rows instead of 5251; the arg-max partial buffers
    that are all-gathered must still have ONE size, and the unused tail slots must never win (all logits negative here, so
    a zero-initialised slot WOULD win)."""
    from usdm_amd.llm import vocab_shard
    V = 42003
    logits = -torch.rand(V, generator=torch.Generator().manual_seed(tp)) - 0.5     # all negative
    logits[V - 3] = -0.25                                                           # the winner lives in the LAST rank's shard
    shards = [vocab_shard(V, r, tp) for r in range(tp)]
    assert len({s[3] for s in shards}) == 1 and len({s[0] for s in shards}) == 1     # nparts and Vloc equal on every rank
    assert shards[0][1] == 0 and shards[-1][2] == V and all(shards[r][2] == shards[r + 1][1] for r in range(tp - 1))
    nparts = shards[0][3]
    gathered_v, gathered_i = [], []
    for Vloc, v0, v1, _ in shards:
        assert v1 - v0 <= Vloc and (v1 - v0 + 15) // 16 <= nparts
        pv, pi = _emulate_lm_head_partials(logits[v0:v1], v0, nparts)
        assert pv.numel() == pi.numel() == nparts
        gathered_v.append(pv); gathered_i.append(pi)
    got = _final_argmax(torch.cat(gathered_v), torch.cat(gathered_i))
    assert got == int(torch.argmax(logits)) == V - 3
    if tp == 8:
        assert shards[-1][2] - shards[-1][1] == 5246 and shards[0][0] == 5251 and nparts == 329


def _gather_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from usdm_amd.llm import vocab_shard
    V = 4099                                                    # odd split: rank 1 is one row short
    logits = -torch.rand(V, generator=torch.Generator().manual_seed(5)) - 0.5
    logits[V - 1] = -0.1
    Vloc, v0, v1, nparts = vocab_shard(V, rank, world)
    pv, pi = _emulate_lm_head_partials(logits[v0:v1], v0, nparts)
    allv, alli = torch.empty(nparts * world), torch.empty(nparts * world, dtype=torch.int32)
    dist.all_gather_into_tensor(allv, pv)                       # the call llm._gather_partials makes (equal counts required)
    dist.all_gather_into_tensor(alli, pi)
    out_q.put((rank, _final_argmax(allv, alli) == int(torch.argmax(logits)), True, True))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_tp2_gloo_gather_of_partials_with_uneven_vocab_split():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(30)
    assert all(r[1] for r in res), res


def _p2p_wiring_worker(rank, world, port, out_q):
    """a rank whose transport is broken must not leave the group's collective sequence on its own."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from usdm_amd import p2p as P
    # 1. no GPU here: creating the exchange buffer fails on EVERY rank -> the same (None, reason) on every rank
    c, why = P.P2PComm.try_from_process_group(dist.group.WORLD, 3, 64)
    r1 = (c is None, "create/export" in (why or ""))
    # 2. uneven failure: rank 0's wiring "works" (stand-ins for the HIP calls), rank 1's import raises
    class Fake(P.P2PComm):
        def __init__(self, rank, world, n_sites, max_elems, timeout_ms=None):
            self.rank, self.world, self._h, self.closed = rank, world, None, False
        def export_handle(self):
            return b"h%d" % self.rank
        def import_handle(self, peer, handle):
            if self.rank == 1:
                raise RuntimeError("hipIpcOpenMemHandle: invalid argument")
        def commit(self):
            pass
        def close(self):
            self.closed = True
    c2, why2 = Fake.try_from_process_group(dist.group.WORLD, 3, 64)
    r2 = (c2 is None, "rank 1: import/commit" in (why2 or ""))
    with pytest.raises(P.P2PError):
        Fake.from_process_group(dist.group.WORLD, 3, 64)
    # 3. agree(): one rank's failure becomes everybody's verdict
    v = P.agree(dist.group.WORLD, "boom" if rank == 1 else None)
    r3 = v == "rank 1: boom"
    # 4. the group is still in step: a plain collective after all of the above completes with the right value
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)
    out_q.put((rank, r1, r2, r3, float(t.item())))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_p2p_wiring_failures_keep_every_rank_in_the_collective_sequence():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_wiring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(2)]
    for p in procs:
        p.join(30)
    for rank, r1, r2, r3, tot in res:
        assert r1 == (True, True) and r2 == (True, True) and r3 and tot == 3.0, res


def test_peer_only_error_word_raises():
    """a rank whose own waits all succeeded but whose error word carries the PEER bit (another rank timed out and wrote
    it through its mapping of this rank's buffer) must raise at its next host sync: the word is decoded on the host, no GPU needed."""
    from usdm_amd.p2p import P2PComm, P2PError
    assert P2PComm.describe_error(0, 3, 7) is None
    own = P2PComm.describe_error(1, 3, 7)                       # USDM_P2P_ERR_TIMEOUT_ROWS on this rank
    peer = P2PComm.describe_error(8 | 2, 3, 7)                  # a peer's TIMEOUT_PICK, marked PEER
    assert "rank 3" in own and "this rank" in own and "PEER" not in own
    assert "PEER" in peer and "0xa" in peer and "invalid" in peer

    class Fake(P2PComm):
        def __init__(self):
            self.rank = 3
        def status(self):
            return 8 | 4, 11
        def __del__(self):
            pass
    with pytest.raises(P2PError, match="PEER"):
        Fake().raise_if_failed()
