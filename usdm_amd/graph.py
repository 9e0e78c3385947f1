"""hipGraph capture of a recorded launch plan (torch.cuda.CUDAGraph is hipGraph on ROCm)."""
import os

import torch


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
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.plan.run()
            self.graph = g
        self.graph.replay()
