"""CPU: host-side pieces of the vLLM-style serving surface (usdm_amd/serving.py): SamplingParams validation and the recognition of
logits processors that are fixed masks (the reference's three, src/inference_vllm.py:70-83) vs history-dependent ones."""
import pytest
import torch

from usdm_amd.serving import SamplingParams, static_mask_of


def u2t(token_ids, logits):
    logits[32000:42003] = float("-inf")
    return logits


def t2u(token_ids, logits):
    logits[0:28705] = float("-inf")
    logits[28706:32002] = float("-inf")
    return logits


def test_sampling_params_validation_and_greedy_detection():
    sp = SamplingParams(max_tokens=8192, top_p=1.0, top_k=1, temperature=1.0, stop_token_ids=[28705], logits_processors=[t2u])
    assert sp.greedy and sp.max_tokens == 8192 and sp.stop_token_ids == [28705]
    assert SamplingParams(temperature=0.0).greedy and not SamplingParams(temperature=0.8, top_k=50).greedy
    for bad in (dict(temperature=-1), dict(top_p=0.0), dict(top_p=1.5), dict(top_k=0), dict(top_k=-2)):
        with pytest.raises(ValueError):
            SamplingParams(**bad)
    assert SamplingParams(n=2, temperature=0.7).n == 2
    for bad in (dict(n=2), dict(n=2, temperature=0.0), dict(n=0), dict(n=1.5)):      # greedy copies would be identical (vllm refuses too)
        with pytest.raises(ValueError):
            SamplingParams(**bad) if "temperature" in bad or bad["n"] != 2 else SamplingParams(n=2, top_k=1)


def test_reference_processors_are_recognised_as_static_masks():
    V = 42003
    m = static_mask_of([u2t], V, "cpu")
    assert m is not None and int(m.sum()) == 10003 and m[31999] == 0 and m[32000] == 1
    m = static_mask_of([t2u], V, "cpu")
    assert m is not None and m[28705] == 0 and m[28704] == 1 and m[32001] == 1 and m[32002] == 0 and int((m == 0).sum()) == 10002
    assert int(static_mask_of([], V, "cpu").sum()) == 0
    both = static_mask_of([u2t, t2u], V, "cpu")            # processors compose
    assert both is not None and int((both == 0).sum()) == 1 and both[28705] == 0


def test_history_or_value_dependent_processors_are_not_static():
    V = 1000

    def no_repeat(token_ids, logits):
        if token_ids:
            logits[token_ids[-1]] = float("-inf")
        return logits

    def temperature_like(token_ids, logits):
        return logits * 0.5

    def length_penalty(token_ids, logits):
        logits[7] = logits[7] - 0.1 * len(token_ids)
        return logits
    for f in (no_repeat, temperature_like, length_penalty):
        assert static_mask_of([f], V, "cpu") is None
