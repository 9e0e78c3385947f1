"""Host side of the one-shot peer-to-peer all-reduce (C-ABI: usdm_allreduce_p2p_*, include/usdm_hip.h).

P2PComm owns this rank's uncached exchange buffer and the mappings of every peer's buffer:
  * ranks in different processes (the production form, one process per GPU): the 64-byte hipIpc handles travel over any
    host channel - here torch.distributed.all_gather_object on the job's process group (gloo or nccl);
  * logical ranks inside one process (single-GPU validation): buffers are attached by address.
"""
import ctypes as C
import os

from ._lib import check, lib

lib.usdm_allreduce_p2p_bytes.restype = C.c_int64
lib.usdm_allreduce_p2p_base.restype = C.c_void_p
lib.usdm_allreduce_p2p_dev.restype = C.c_void_p

HANDLE_BYTES = 64
MAX_RANKS = 8


class P2PError(RuntimeError):
    pass


class P2PComm:
    def __init__(self, rank, world, n_sites, max_elems, timeout_ms=None):
        if timeout_ms is None:
            timeout_ms = int(os.environ.get("USDM_P2P_TIMEOUT_MS", "10000"))
        self.rank, self.world, self.n_sites, self.max_elems = rank, world, n_sites, max_elems
        self._h = C.c_void_p()
        check(lib.usdm_allreduce_p2p_create(C.c_int32(rank), C.c_int32(world), C.c_int32(n_sites), C.c_int32(max_elems),
                                            C.c_int32(timeout_ms), C.byref(self._h)), "usdm_allreduce_p2p_create")
        self.dev_ptr = None
        self.base = lib.usdm_allreduce_p2p_base(self._h)
        self.nbytes = lib.usdm_allreduce_p2p_bytes(C.c_int32(n_sites), C.c_int32(max_elems))

    # ---- wiring
    def export_handle(self):
        buf = (C.c_char * HANDLE_BYTES)()
        check(lib.usdm_allreduce_p2p_export(self._h, buf), "usdm_allreduce_p2p_export")
        return bytes(buf)

    def import_handle(self, peer, handle):
        check(lib.usdm_allreduce_p2p_import(self._h, C.c_int32(peer), C.c_char_p(handle)), "usdm_allreduce_p2p_import")

    def attach(self, peer, other):
        check(lib.usdm_allreduce_p2p_attach(self._h, C.c_int32(peer), C.c_void_p(other.base)), "usdm_allreduce_p2p_attach")

    def commit(self):
        check(lib.usdm_allreduce_p2p_commit(self._h, C.c_void_p(0)), "usdm_allreduce_p2p_commit")
        self.dev_ptr = lib.usdm_allreduce_p2p_dev(self._h)
        if not self.dev_ptr:
            raise P2PError("usdm_allreduce_p2p_commit left no device view")

    @classmethod
    def from_process_group(cls, group, n_sites, max_elems, timeout_ms=None):
        """One rank per process: exchange the hipIpc handles over `group` and map every peer.  Raises P2PError on EVERY rank
        if any rank failed (see try_from_process_group): the ranks never leave the group's collective sequence one by one."""
        c, why = cls.try_from_process_group(group, n_sites, max_elems, timeout_ms)
        if c is None:
            raise P2PError(f"peer-to-peer communicator could not be wired: {why}")
        return c

    @classmethod
    def try_from_process_group(cls, group, n_sites, max_elems, timeout_ms=None):
        """-> (comm, None) on every rank, or (None, reason) on every rank.  A local failure (hipMalloc, hipIpcGetMemHandle,
        hipIpcOpenMemHandle of one peer, ...) is RECORDED, not raised: every rank runs the same two object all-gathers, learns
        whether all ranks succeeded, and only then do they all keep or all drop the transport (a rank that leaves
        the collective sequence early leaves its peers in mismatched collectives - a hang or garbage under RCCL)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        c, handle, err = None, None, None
        try:
            c = cls(rank, world, n_sites, max_elems, timeout_ms)
            handle = c.export_handle()
        except Exception as e:  # noqa: BLE001
            err = f"rank {rank}: create/export: {e!r}"
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
        if err is None and all(h is not None for h in handles):
            try:
                for r in range(world):
                    if r != rank:
                        c.import_handle(r, handles[r])
                c.commit()
            except Exception as e:  # noqa: BLE001
                err = f"rank {rank}: import/commit: {e!r}"
        elif err is None:
            err = f"rank {rank}: peer(s) {[r for r, h in enumerate(handles) if h is None]} exported no handle"
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        bad = [e for e in errs if e is not None]
        if bad:
            if c is not None:
                c.close()
            return None, "; ".join(bad)
        dist.barrier(group=group)      # every rank has mapped every buffer before anybody writes
        return c, None

    @classmethod
    def in_process(cls, world, n_sites, max_elems, timeout_ms=None):
        """`world` logical ranks inside this process (one GPU): the single-GPU validation form."""
        comms = [cls(r, world, n_sites, max_elems, timeout_ms) for r in range(world)]
        for a in comms:
            for b in comms:
                if a is not b:
                    a.attach(b.rank, b)
            a.commit()
        return comms

    # ---- use
    def check_site(self, site, n):
        if self.dev_ptr is None:
            raise P2PError("P2PComm is not committed")
        if not (0 <= site < self.n_sites) or n > self.max_elems:
            raise ValueError(f"p2p site {site} / {n} elements outside the communicator ({self.n_sites} sites x {self.max_elems})")

    def status(self):
        """(err, epoch): synchronous 8-byte read of this rank's header."""
        err, ep = C.c_int32(0), C.c_int32(0)
        check(lib.usdm_allreduce_p2p_error(self._h, C.byref(err), C.byref(ep)), "usdm_allreduce_p2p_error")
        return err.value, ep.value

    ERR_PEER = 8          # USDM_P2P_ERR_PEER (include/usdm_hip.h): ORed into THIS rank's word, with its own code, by a peer whose wait expired

    @staticmethod
    def describe_error(err, rank, epoch):
        """Message for a non-zero error word; None for 0.  A word that carries only the PEER bit still raises: the rank that timed out
        has substituted zeros for the missing partials, so every rank's results are invalid from that exchange on."""
        if not err:
            return None
        codes, peer = err & ~P2PComm.ERR_PEER, bool(err & P2PComm.ERR_PEER)
        what = "a PEER's wait expired (it carried on with zeros; its timeout code is included)" if peer else \
               "a wait of this rank expired: a peer never delivered its partial sums within the bound"
        return (f"peer-to-peer all-reduce failed on rank {rank} (error word {err:#x}: timeout codes {codes:#x}, epoch {epoch}): {what}; "
                "results from this point on are invalid")

    def raise_if_failed(self):
        err, ep = self.status()
        msg = self.describe_error(err, self.rank, ep)
        if msg:
            raise P2PError(msg)

    def close(self, group=None):
        """Unmaps the peers' buffers and frees this rank's.  Pass the process group to close COLLECTIVELY: a rank whose wait expires
        writes the PEER bit into every peer's error word through its mapping of their buffers, so no rank may free its buffer while a
        peer can still be inside a decode kernel - the barrier orders every rank's last launch before any unmap."""
        if self._h:
            if group is not None:
                import torch
                import torch.distributed as dist
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                dist.barrier(group=group)
            lib.usdm_allreduce_p2p_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class InProcessGroup:
    """Collectives between LOGICAL ranks that live in one process on one GPU (validation of the tensor-parallel algebra and
    of the peer-to-peer exchange without a multi-GPU node).  Two flavours:
      threaded=False  ranks are driven one after another by a lockstep harness (tests/_tp_lockstep.py): a collective completes
                      when the LAST rank has called it (calls must come in rank order);
      threaded=True   every rank runs generate() in its own thread on its own stream: a collective is a thread barrier.
    The prefill all-reduces and the RCCL-path gathers go through here; the decode exchange does not (it is in the kernels)."""

    def __init__(self, world, threaded=False):
        import threading
        self.world, self.threaded = world, threaded
        self._pend = []
        self._bar = threading.Barrier(world) if threaded else None
        self._slots = [None] * world

    def _sum_into_all(self, tensors):
        import torch
        torch.cuda.synchronize()
        acc = tensors[0].clone()
        for t in tensors[1:]:
            acc += t                      # rank order
        for t in tensors:
            t.copy_(acc)
        torch.cuda.synchronize()

    def usdm_all_reduce(self, rank, t):
        if self.threaded:
            self._slots[rank] = t
            if self._bar.wait() == 0:
                self._sum_into_all(list(self._slots))
            self._bar.wait()
            return
        assert rank == len(self._pend), "lockstep collectives must be called in rank order"
        self._pend.append(t)
        if len(self._pend) == self.world:
            self._sum_into_all(self._pend)
            self._pend = []

    def usdm_all_gather(self, rank, dsts, srcs):
        """dsts[i] <- concatenation over ranks of srcs[i] (lists: one rendezvous for several tensors)."""
        import torch

        def finish(entries):
            torch.cuda.synchronize()
            for i in range(len(entries[0][0])):
                cat = torch.cat([e[1][i] for e in entries])
                for e in entries:
                    e[0][i].copy_(cat.reshape(e[0][i].shape))      # rank-major, as all_gather_into_tensor lays it out
            torch.cuda.synchronize()
        if self.threaded:
            self._slots[rank] = (dsts, srcs)
            if self._bar.wait() == 0:
                finish(list(self._slots))
            self._bar.wait()
            return
        assert rank == len(self._pend), "lockstep collectives must be called in rank order"
        self._pend.append((dsts, srcs))
        if len(self._pend) == self.world:
            finish(self._pend)
            self._pend = []


def agree(group, local_failure):
    """Every rank contributes its first local failure (a string) or None; every rank gets the same verdict back: None if all
    are None, else the failures joined.  One object all-gather, reached by every rank whatever happened before."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    got = [None] * world
    dist.all_gather_object(got, local_failure, group=group)
    bad = [f"rank {r}: {m}" for r, m in enumerate(got) if m is not None]
    return "; ".join(bad) if bad else None


def self_test(comm, group, device, rounds=4, N=4096, K=512, fused=True):
    """Start-up check of a freshly wired communicator across the REAL ranks of `group`: `rounds` fused all-reduces (the GEMV
    epilogue form, both parities) checked against the group's own all_reduce of the same f32 partials, and `rounds` cross-rank
    token picks checked against the known winner.  The default shape is the o_proj shard of the 7B at TP = 8 (4096 outputs: the
    256-workgroup, 16-wave launch the decode step uses), so that a node on which those grids cannot make progress fails HERE.
    fused=False checks the split form (put in the epilogue + usdm_allreduce_p2p_reduce).
    Returns the SAME verdict on every rank: None if everything matched everywhere, else a description of what failed where
    (callers fall back to the RCCL path).  A rank that sees a timeout, a mismatch or an exception records it and KEEPS GOING:
    every rank executes every round's group collective and the final agreement, so the collective sequence on the group can
    never get out of step because one rank's transport is broken."""
    import torch
    import torch.distributed as dist
    from . import ops
    fail = None

    def note(msg):
        nonlocal fail
        if fail is None:
            fail = msg

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    staged = dist.get_backend(group) == "gloo"
    ep0 = 0
    try:
        if comm.n_sites < 2:
            note("self-test needs at least two sites (rows + token pick)")
        _, ep0 = comm.status()
    except Exception as e:  # noqa: BLE001
        note(f"status raised {e!r}")
    g = torch.Generator().manual_seed(1000 + rank)
    gs = torch.Generator().manual_seed(999)          # shared by all ranks
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=device)
    for it in range(rounds):
        W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(device)
        x = torch.randn(K, generator=g).to(torch.bfloat16).to(device)
        h0 = torch.randn(N, generator=gs).to(torch.bfloat16).to(device)
        part = torch.zeros(N, dtype=torch.float32, device=device)
        try:
            ops.gemv(W, x, N=N, K=K, round_bf16=False, y32=part)
        except Exception as e:  # noqa: BLE001
            note(f"round {it}: gemv raised {e!r}")
        # the group's own collective: ALWAYS executed, by every rank, in every round
        if staged:
            c = part.cpu(); dist.all_reduce(c, group=group); tot = c.to(device)
        else:
            tot = part.clone(); dist.all_reduce(tot, group=group)
        if fail is not None and "needs at least two sites" in fail:
            continue
        try:
            want = (h0.float() + tot.to(torch.bfloat16).float()).to(torch.bfloat16)
            h = h0.clone()
            site = it % (comm.n_sites - 1)            # (the last site is the token pick's)
            ops.gemv(W, x, N=N, K=K, residual=h, y16=h, p2p=comm, p2p_site=site, p2p_mode=1 if fused else 2)
            if not fused:
                ops.p2p_reduce(comm, site, N, h)
            torch.cuda.synchronize()
            err, _ = comm.status()
            if err:
                note(f"round {it}: exchange timed out (error word {err:#x})")
            diff = (h.float() - want.float()).abs()
            ulp = want.float().abs().clamp_min(1e-3) * 2 ** -7
            if bool((diff > 2 * ulp).any()) or float((diff == 0).float().mean()) < 0.98:
                note(f"round {it}: all-reduce result differs from the group's all_reduce (max diff {float(diff.max()):.3e})")
            # token pick: rank (it % world) holds the winner
            pv = torch.full((4,), -1.0 - rank, device=device)
            pi = torch.arange(4, dtype=torch.int32, device=device) + 100 * rank
            if rank == it % world:
                pv[2] = 5.0
            nxt, out, step, pos = i32(1), i32(8), i32(1), i32(1)
            st = ops.decode_state(nxt, out, step, pos)
            ops.argmax_p2p(pv, pi, 4, st, comm, comm.n_sites - 1, phase=0)
            torch.cuda.synchronize()
            if int(nxt.item()) != 100 * (it % world) + 2:
                note(f"round {it}: cross-rank token pick returned {int(nxt.item())}")
        except Exception as e:  # noqa: BLE001 - recorded; the remaining rounds' collectives still run
            note(f"round {it}: raised {e!r}")
    try:
        err, ep = comm.status()
        if err or ep != ep0 + rounds:
            note(f"status after self-test: error word {err:#x}, epoch {ep} (expected {ep0 + rounds})")
    except Exception as e:  # noqa: BLE001
        note(f"status raised {e!r}")
    return agree(group, fail)
