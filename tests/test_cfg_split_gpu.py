"""GPU: the two classifier-free-guidance halves of the Voicebox estimator on a PAIR of ranks (optional row;
reference: model/voicebox.py:60-71 builds the doubled batch [null tokens ; tokens], [0 ; cond] and combines v_c + gs (v_c - v_u)).

Two processes share this GPU (gloo group, velocities all-gathered through host memory - the validation form; real ranks use RCCL).
Each runs Voicebox.generate(cfg_group=pair) at batch 1; the result must equal the single-process batch-2 generate() of the same
model / inputs / noise.  Every per-row operation of the estimator is independent of the batch it sits in and the GEMM tiles
accumulate in the same order whatever tile is selected (tests/test_gemm_gpu.py), so the expected difference is ZERO; the test asserts
a tolerance far below the bf16 plan's own error and prints what it measured."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ld(name):
    return {k: (torch.from_numpy(v) if v.ndim else v.item()) for k, v in np.load(os.path.join(G, name)).items()}


def _model(dev):
    from oracle import voicebox_oracle as VO
    from tests.golden.configs import SMALL_VB
    from usdm_amd.voicebox.model import Voicebox
    d = _ld("voicebox_small.npz")
    kw = {k: SMALL_VB[k] for k in SMALL_VB if k != "sigma_min"}
    m = Voicebox(**kw, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=SMALL_VB["sigma_min"])
    m.load_state_dict(VO.random_state_dict(SMALL_VB, int(d["seed"])), strict=True)
    return m.to(dev).eval(), d


def _gen(m, d, dev, **kw):
    to = lambda t: t.to(dev)
    tr = []
    heun = m.generate(to(d["x"]), to(d["cond"]), to(d["lengths"]), n_timesteps=int(d["nt_h"]), solver="heun", gradient_scale=1.0,
                      speech_prompt=True, prompt_lengths=torch.tensor([int(d["P"])]).to(dev), noise=d["noise_h"], trace=tr, **kw)
    euler = m.generate(to(d["x"]), torch.zeros_like(to(d["cond"])), to(d["lengths"]), n_timesteps=int(d["nt_e"]), solver="euler",
                       gradient_scale=float(d["gs_e"]), speech_prompt=False, noise=d["noise_e"], **kw)
    return heun.cpu(), euler.cpu(), [t.cpu() for t in tr]


def _proc(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        m, d = _model(dev)
        pair = dist.new_group([0, 1])
        heun, euler, tr = _gen(m, d, dev, cfg_group=pair)
        heun2, _, _ = _gen(m, d, dev, cfg_group=pair)          # second call: the captured graph of the batch-1 plan
        dist.barrier()
        q.put((rank, heun.numpy(), euler.numpy(), [t.numpy() for t in tr], bool(torch.equal(heun, heun2)), None))
    except Exception as e:  # noqa: BLE001
        q.put((rank, None, None, None, None, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_cfg_halves_on_two_ranks_equal_the_batch2_plan(dev):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_proc, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(r[5] is None for r in res), [r[5] for r in res]
    (_, h0, e0, tr0, same0, _), (_, h1, e1, tr1, same1, _) = res
    assert same0 and same1
    assert np.array_equal(h0, h1) and np.array_equal(e0, e1)                 # both ranks of the pair return the same mel
    m, d = _model(dev)
    heun, euler, tr = _gen(m, d, dev)                                        # the batch-2 plan in this process
    dh = np.abs(h0 - heun.numpy()).max() / np.abs(heun.numpy()).max()
    de = np.abs(e0 - euler.numpy()).max() / np.abs(euler.numpy()).max()
    dv = max(np.abs(a - b.numpy()).max() for a, b in zip(tr0, tr))
    rel_ref = ((heun.double() - d["gen_h"].double()).norm() / d["gen_h"].double().norm()).item()
    print(f"CFG halves on two ranks vs the batch-2 plan: max rel diff Heun+prompt {dh:.2e}, Euler gs=0.7 {de:.2e}, per-NFE velocities "
          f"{dv:.2e} (the bf16 plan itself is {rel_ref:.2e} from the reference golden vector)")
    assert len(tr0) == len(tr) and tr0[0].shape == tr[0].shape == (2,) + tuple(heun.shape[1:])
    assert dh <= 1e-5 and de <= 1e-5 and dv <= 1e-5
