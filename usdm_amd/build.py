"""Builds libusdm_hip.so (all HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

Used by __graft_entry__.build(); the .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libusdm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-ffp-contract=off"]
FLAGS += os.environ.get("USDM_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DUSDM_GEMM_TRACE for tools/gemm_trace.py


# per-file extra flags.  attn.hip: MFMA results stay in VGPRs (the softmax consumes them with VALU ops; with the default AGPR
# form 22 % of the loop's VALU instructions were v_accvgpr_read/write moves)
EXTRA = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


# Kernels that measured SLOWER than the default path (profiles/r02_decode_ablation.txt 3-4, profiles/r03_tp_ablation.txt) are kept for
# the record and their tests, but not in the product library: they build into libusdm_hip_experimental.so, which nothing loads
# unless usdm_amd.ops.gemv_chain / gemv_engine are called (include/usdm_hip_experimental.h).
EXPERIMENTAL = {"llm_chain_k.hip", "llm_engine_k.hip"}
OUT_EXP = os.path.join(HERE, "libusdm_hip_experimental.so")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "..", "include", h) for h in ("usdm_hip.h", "usdm_hip_experimental.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objs, jobs = [], []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    objs_exp = []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s.replace(".hip", ".o"))
        (objs_exp if s in EXPERIMENTAL else objs).append(obj)
        if force or _stale(obj, src):
            jobs.append([HIPCC, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT) or not os.path.exists(OUT_EXP):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
        # the experimental kernels resolve usdm_set_error (csrc/core.hip) in the product library, next to which they are installed
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT_EXP, *objs_exp, "-L" + HERE, "-lusdm_hip", "-Wl,-rpath,$ORIGIN"])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
