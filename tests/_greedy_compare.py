"""Shared checker: greedy ids of the HIP LLM vs the CPU oracle, continuing past legitimate divergences.

A divergence is legitimate only at a near-tie of the ORACLE's own bf16 logits (top-2 gap within bf16 noise of the logit
magnitude).  After one, the HIP model is re-prompted with the oracle's tokens up to and including the diverging position
(teacher forcing through the prefill path) so that the rest of the sequence is still compared token by token."""
import torch

NEAR_TIE_REL = 2 ** -6      # two bf16 ulps of the top logit
NEAR_TIE_ABS = 1e-3


def compare_greedy(model, dev, ids, ref, ref_logits, new, max_restarts=8, **gen_kw):
    """ids: int64 [L0] prompt; ref: oracle id list (prompt + new); ref_logits: [new, V] masked oracle logits.
    Returns (list of diverging generated-token indices, number of tokens compared)."""
    L0 = ids.numel()
    ref_gen = ref[L0:]
    assert len(ref_gen) == new
    done, div = 0, []
    while done < new:
        prompt = torch.tensor(ref[:L0 + done], dtype=torch.long)[None].to(dev)
        out = model.generate(input_ids=prompt, max_new_tokens=new - done, **gen_kw)[0].tolist()
        assert out[:L0 + done] == ref[:L0 + done]
        gen = out[L0 + done:]
        assert len(gen) == new - done
        first = next((i for i in range(len(gen)) if gen[i] != ref_gen[done + i]), None)
        if first is None:
            done = new
            break
        j = done + first
        top2 = torch.topk(ref_logits[j], 2).values
        gap = (top2[0] - top2[1]).item()
        assert gap <= NEAR_TIE_REL * top2[0].abs().item() + NEAR_TIE_ABS, \
            f"generated token {j}: HIP {gen[first]} vs oracle {ref_gen[j]} with oracle top-2 gap {gap} (not a near-tie)"
        # the HIP choice must itself lie inside the oracle's near-tie band (ties can be more than two-way)
        mine = ref_logits[j][gen[first]].item()
        assert top2[0].item() - mine <= NEAR_TIE_REL * top2[0].abs().item() + NEAR_TIE_ABS, \
            f"token {j}: HIP picked id {gen[first]} whose oracle logit {mine} is outside the near-tie band below {top2[0].item()}"
        div.append(j)
        assert len(div) <= max_restarts, f"too many near-tie divergences: {div}"
        done = j + 1
    return div, new


def check_against_oracle(out, ref, ref_logits, L0):
    """One already-generated id list vs the oracle's: equal, or first difference at an oracle near-tie with the HIP pick inside
    the tie band (no continuation past it).  Returns the index of the first difference among the generated tokens, or None."""
    assert out[:L0] == ref[:L0]
    n = min(len(out), len(ref))
    first = next((i for i in range(L0, n) if out[i] != ref[i]), None)
    if first is None:
        assert len(out) == len(ref), (len(out), len(ref))
        return None
    lg = ref_logits[first - L0]
    top = lg.max().item()
    tol = NEAR_TIE_REL * abs(top) + NEAR_TIE_ABS
    top2 = torch.topk(lg, 2).values
    assert (top2[0] - top2[1]).item() <= tol, f"generated token {first - L0}: differs from the oracle away from a near-tie (gap {(top2[0] - top2[1]).item()})"
    assert top - lg[out[first]].item() <= tol, f"generated token {first - L0}: HIP pick outside the oracle's near-tie band"
    return first - L0
