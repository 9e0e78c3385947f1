"""Single-GPU harness: N LOGICAL tensor-parallel ranks of USDMForCausalLM inside one process, driven in lockstep.

Every rank's launch plan is cut into segments at its exchange points; the harness runs segment k of rank 0, 1, .. N-1 before
segment k+1 of anybody, so that a rank's peer-to-peer puts (split form) are complete, by stream order, before any rank's
reduce launches.  Addressing, slot layout, summation order, epoch/parity handling and the token pick are thereby exercised
exactly as on N GPUs; what this cannot show is xGMI visibility and timing."""
import torch

from usdm_amd import ops


def _run_lockstep(seg_lists):
    n = len(seg_lists[0])
    assert all(len(s) == n for s in seg_lists)
    for k in range(n):
        for segs in seg_lists:
            s = segs[k]
            if isinstance(s, ops.Plan):
                s.run()
            else:
                s()


@torch.no_grad()
def lockstep_generate(models, input_ids, max_new_tokens, bad_words_ids=None):
    """Greedy generation with every logical rank stepping together.  Returns the id list and checks that all ranks agree."""
    prefill = []
    for m in models:
        segs, _ = m._setup_call(input_ids, 0, False, bad_words_ids, None, 0)
        prefill.append(segs)
    _run_lockstep(prefill)                       # prefill + first token
    decode = [m._build_decode(False) for m in models]
    for _ in range(max_new_tokens - 1):
        _run_lockstep(decode)
    torch.cuda.synchronize()
    outs = [m.st_out[:max_new_tokens].tolist() for m in models]
    for m in models:
        assert int(m.st_step.item()) == max_new_tokens
        if m.p2p is not None:
            m.p2p.raise_if_failed()
    assert all(o == outs[0] for o in outs), "logical ranks disagree on the generated tokens"
    return input_ids[0].tolist() + outs[0]
