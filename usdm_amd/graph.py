"""hipGraph capture of a recorded launch plan (torch.cuda.CUDAGraph is hipGraph on ROCm)."""
import contextlib
import gc
import os

import torch


@contextlib.contextmanager
def _no_gc():
    """Python's cyclic GC must not run while a stream is capturing: collecting an old hipGraph / tensor there issues HIP
    calls that are illegal during capture and abort the process (seen once in the full GPU test run)."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class GraphedPlan:
    """First run() launches eagerly (and validates every launch); the second run() captures the same
    launch sequence into a hipGraph on a capture stream; later runs replay the graph."""

    def __init__(self, plan, enabled=None):
        self.plan = plan
        self.enabled = (os.environ.get("USDM_NO_GRAPH", "0") != "1") if enabled is None else enabled
        self.graph = None
        self.runs = 0

    def run(self):
        self.runs += 1
        if not self.enabled or self.runs == 1:
            self.plan.run()
            return
        if self.graph is None:
            # capture outside inference mode: the RNG bookkeeping tensors torch registers at the first capture must
            # not become inference tensors (callers such as reconstruct_speech run under @torch.inference_mode()),
            # or a later capture outside inference mode fails with "Inplace update to inference tensor"
            with torch.inference_mode(False), _no_gc():
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.plan.run()
            self.graph = g
        self.graph.replay()


class GraphedSegments:
    """Same as GraphedPlan for a list of segments (Plans interleaved with callables such as RCCL collectives).
    Capturing collectives into a hipGraph is attempted once; if the runtime refuses, the segments keep running
    eagerly (still the HIP path, just launch-bound)."""

    def __init__(self, segs, run_segs, enabled=None):
        self.segs, self.run_segs = segs, run_segs
        # Collectives are captured with the kernels (torch's ProcessGroupNCCL supports stream capture; verified here on a
        # 1-rank RCCL group, the only kind a one-GPU box can form).  Eager TP is host-bound (~65 collectives + ~200
        # launches per token), so the graph is the default; USDM_TP_GRAPH=0 forces eager, and any capture error falls
        # back to eager launches.
        self.enabled = (os.environ.get("USDM_NO_GRAPH", "0") != "1" and os.environ.get("USDM_TP_GRAPH", "1") == "1") if enabled is None else enabled
        self.graph, self.runs, self.failed = None, 0, None

    def run(self):
        self.runs += 1
        if not self.enabled or self.runs == 1 or self.failed:
            self.run_segs(self.segs)
            return
        if self.graph is None:
            try:
                with torch.inference_mode(False), _no_gc():
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.run_segs(self.segs)
                self.graph = g
            except Exception as e:  # noqa: BLE001 - any capture failure falls back to eager launches
                self.failed = repr(e)
                torch.cuda.synchronize()
                self.run_segs(self.segs)
                return
        self.graph.replay()
