"""GPU: the vLLM-style serving surface (usdm_amd/serving.py; reference src/inference_vllm.py:42-83,109-125) against the CPU oracle:
the reference's three rounds with its own logits-processor functions and SamplingParams, continuous batching over the 4 decode
slots, history-dependent processors, sampling knobs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(vocab_size=42003, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
           num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
SMALL = dict(CFG, vocab_size=1000)


# the reference's processors (src/inference_vllm.py:70-83), restated
def bad_word_processor_unit2text(token_ids, logits):
    logits[32000:42003] = float("-inf")
    return logits


def bad_word_processor_text2text(token_ids, logits):
    logits[32002:42003] = float("-inf")
    return logits


def bad_word_processor_text2unit(token_ids, logits):
    logits[0:28705] = float("-inf")
    logits[28706:32002] = float("-inf")
    return logits


def _ban_list(proc, V):
    lg = proc([], torch.zeros(V))
    return [[i] for i in torch.nonzero(torch.isneginf(lg)).flatten().tolist()]


def test_reference_sampling_params_and_processors_vs_oracle(dev):
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams, static_mask_of
    sd = MO.random_state_dict(CFG, seed=41)
    m = USDMForCausalLM.from_state_dict(sd, CFG, dev, ctx_max=256)
    eng = LLM(model=m)
    g = torch.Generator().manual_seed(1)
    for proc, stop, lo, hi in ((bad_word_processor_unit2text, 98, 0, 32000), (bad_word_processor_text2text, 32001, 0, 32002),
                               (bad_word_processor_text2unit, 28705, 32002, 42003)):
        mask = static_mask_of([proc], 42003, dev)
        assert mask is not None and int(mask.sum()) == len(_ban_list(proc, 42003))          # recognised as a fixed mask
        sp = SamplingParams(max_tokens=20, top_p=1.0, top_k=1, temperature=1.0, stop_token_ids=[stop], logits_processors=[proc])
        ids = torch.randint(3, 32000, (45,), generator=g)
        out = eng.generate(prompt_token_ids=[ids.tolist()], sampling_params=sp)
        assert len(out) == 1 and out[0].prompt_token_ids == ids.tolist()
        toks = out[0].outputs[0].token_ids
        ref, ref_logits = MO.greedy_generate(sd, CFG, ids, 20, bad_words_ids=_ban_list(proc, 42003), eos_token_id=stop, return_logits=True)
        check_against_oracle(ids.tolist() + toks, ref, ref_logits, 45)
        assert all(lo <= t < hi or t == stop for t in toks)
        assert out[0].outputs[0].finish_reason == ("stop" if toks[-1] == stop else "length")
    assert eng.stats["hook_requests"] == 0                                                       # the fast path served all three


def test_continuous_batching_vs_oracle(dev):
    """Nine greedy requests with one mask, ragged prompts, different max_tokens and a stop id: served through the 4 decode slots with
    slot turnover; every sequence equals oracle greedy generation of its prompt (near-tie rule)."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams
    sd = MO.random_state_dict(SMALL, seed=43)
    eng = LLM(model=USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=256), max_num_seqs=4)
    g = torch.Generator().manual_seed(2)

    def ban(token_ids, logits):
        logits[0:250] = float("-inf")
        return logits
    bad = [[i] for i in range(250)]
    lens = (40, 17, 65, 33, 50, 21, 58, 29, 44)
    prompts = [torch.randint(0, 1000, (L,), generator=g) for L in lens]
    probe = MO.greedy_generate(sd, SMALL, prompts[1], 12, bad_words_ids=bad)
    stop = probe[17 + 5]
    max_new = (30, 12, 9, 26, 18, 35, 5, 22, 16)
    sps = [SamplingParams(max_tokens=mn, top_k=1, stop_token_ids=[stop], logits_processors=[ban]) for mn in max_new]
    outs = eng.generate(prompt_token_ids=[p.tolist() for p in prompts], sampling_params=sps)
    assert [o.request_id for o in outs] == [str(i) for i in range(9)]
    firsts = []
    for p, mn, o in zip(prompts, max_new, outs):
        ref, ref_logits = MO.greedy_generate(sd, SMALL, p, mn, bad_words_ids=bad, eos_token_id=stop, return_logits=True)
        toks = o.outputs[0].token_ids
        firsts.append(check_against_oracle(p.tolist() + toks, ref, ref_logits, p.numel()))
        assert len(toks) <= mn and all(t >= 250 for t in toks)
        assert o.outputs[0].finish_reason == ("stop" if toks[-1] == stop else "length")
    st = eng.stats
    print("continuous batching:", st, "first differences:", firsts)
    assert st["batched_requests"] == 9 and st["admissions"] == 9 and st["max_active"] == 4        # 4 slots, 5 refills
    assert st["batched_steps"] < sum(max_new)                                                      # steps were shared between sequences


def test_sampled_requests_inside_a_continuous_batch_equal_their_single_runs(dev):
    """Per-slot sampling state in the batched decode (reference: src/inference_vllm.py:109-123 hands every request its own
    SamplingParams; the demo's knobs: streamlit_demo.py:201-211): two SAMPLED requests (different temperature / top-k / top-p / seed)
    and four greedy ones share one mask and are served together over the 4 slots.  A sampled request must return exactly the tokens
    it returns when served alone on the single-sequence graph (same seed -> same Philox stream, same logits bit for bit); the greedy
    ones must still equal the oracle."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams
    sd = MO.random_state_dict(SMALL, seed=49)
    eng = LLM(model=USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=256), max_num_seqs=4)      # VALU form: bit-identical per slot
    g = torch.Generator().manual_seed(6)

    def ban(token_ids, logits):
        logits[0:300] = float("-inf")
        return logits
    bad = [[i] for i in range(300)]
    lens = (38, 22, 51, 30, 45, 27)
    prompts = [torch.randint(0, 1000, (L,), generator=g).tolist() for L in lens]
    sps = [SamplingParams(max_tokens=18, top_k=1, logits_processors=[ban]),
           SamplingParams(max_tokens=24, temperature=1.3, top_p=0.9, top_k=50, seed=11, logits_processors=[ban]),
           SamplingParams(max_tokens=9, top_k=1, logits_processors=[ban]),
           SamplingParams(max_tokens=30, temperature=0.8, top_p=1.0, top_k=-1, seed=12345, logits_processors=[ban]),
           SamplingParams(max_tokens=21, temperature=0.0, logits_processors=[ban]),
           SamplingParams(max_tokens=14, top_k=1, logits_processors=[ban])]
    outs = eng.generate(prompt_token_ids=prompts, sampling_params=sps)
    st = dict(eng.stats)
    assert st["batched_requests"] == 6 and st["sampled_in_batch"] == 2 and st["max_active"] == 4
    for i in (1, 3):                                   # the sampled ones, alone
        alone = eng.generate(prompt_token_ids=[prompts[i]], sampling_params=sps[i])[0].outputs[0].token_ids
        got = outs[i].outputs[0].token_ids
        assert got == alone, f"request {i}: sampled inside the batch {got} != alone {alone}"
        assert len(got) == sps[i].max_tokens and all(t >= 300 for t in got)
    a = eng.generate(prompt_token_ids=[prompts[1]], sampling_params=SamplingParams(max_tokens=24, temperature=1.3, top_p=0.9, top_k=50, seed=12,
                                                                                  logits_processors=[ban]))[0].outputs[0].token_ids
    assert a != outs[1].outputs[0].token_ids          # another seed, another stream
    for i in (0, 2, 4, 5):                             # the greedy ones still equal the oracle (they rode the SAMPLING graph with top_k = 1)
        p = torch.tensor(prompts[i])
        ref, ref_logits = MO.greedy_generate(sd, SMALL, p, sps[i].max_tokens, bad_words_ids=bad, return_logits=True)
        check_against_oracle(prompts[i] + outs[i].outputs[0].token_ids, ref, ref_logits, len(prompts[i]))
    print("per-slot sampling in a continuous batch:", st)


def test_n_completions_per_prompt_are_the_seeded_single_requests(dev):
    """SamplingParams(n = 3) (vllm surface, src/inference_vllm.py:109-123 passes SamplingParams through unchanged): three
    completions per prompt, completion j equal to a single request with seed + j, served through the continuous batch together with
    a greedy request; greedy with n > 1 is refused as vllm does."""
    from oracle import mistral_oracle as MO
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams
    sd = MO.random_state_dict(SMALL, seed=50)
    eng = LLM(model=USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=256))
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randint(0, 1000, (L,), generator=g).tolist() for L in (33, 20)]
    kw = dict(max_tokens=16, temperature=1.1, top_p=0.95, top_k=40)
    outs = eng.generate(prompt_token_ids=prompts, sampling_params=[SamplingParams(n=3, seed=100, **kw), SamplingParams(max_tokens=12, top_k=1)])
    assert len(outs) == 2 and [len(o.outputs) for o in outs] == [3, 1] and [c.index for c in outs[0].outputs] == [0, 1, 2]
    assert eng.stats["batched_requests"] == 4 and eng.stats["sampled_in_batch"] == 3
    seqs = [c.token_ids for c in outs[0].outputs]
    assert len({tuple(t) for t in seqs}) == 3                                          # three different streams
    for j in range(3):
        alone = eng.generate(prompt_token_ids=[prompts[0]], sampling_params=SamplingParams(seed=100 + j, **kw))[0].outputs[0].token_ids
        assert seqs[j] == alone, f"completion {j}: {seqs[j]} != seeded single request {alone}"
    with pytest.raises(ValueError):
        SamplingParams(n=2, temperature=0.0)
    with pytest.raises(ValueError):
        SamplingParams(n=0)


def test_batched_request_running_into_the_context_limit(dev):
    """the reference passes max_tokens = tokenizer.model_max_length, so a greedy sequence that never emits its stop id
    decodes up to the context limit.  In a continuous batch it must stop AT the limit (never append cache rows >= ctx_max, which
    would be another head's / slot's rows) while its neighbours keep decoding and a queued request takes over its slot."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams
    sd = MO.random_state_dict(SMALL, seed=47)
    m = USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=128)
    eng = LLM(model=m, max_num_seqs=4)
    g = torch.Generator().manual_seed(5)
    lens = (101, 20, 99, 33, 25)                      # slots 0 and 2 hit the limit (27 / 29 tokens of room) at different steps
    want = (100000, 70, 100000, 60, 50)               # what the caller asks for (model_max_length-style for the long prompts)
    prompts = [torch.randint(0, 1000, (L,), generator=g) for L in lens]
    sps = [SamplingParams(max_tokens=w, top_k=1) for w in want]
    outs = eng.generate(prompt_token_ids=[p.tolist() for p in prompts], sampling_params=sps)
    for p, w, o in zip(prompts, want, outs):
        n = min(w, 128 - p.numel())
        ref, ref_logits = MO.greedy_generate(sd, SMALL, p, n, return_logits=True)
        toks = o.outputs[0].token_ids
        check_against_oracle(p.tolist() + toks, ref, ref_logits, p.numel())
        assert len(toks) == n and o.outputs[0].finish_reason == "length"
    assert eng.stats["batched_requests"] == 5 and eng.stats["max_active"] == 4
    # the guard inside usdm_attn_decode: a position at / past the end of the cache appends nothing and reads nothing out of range
    bb = m._batch_buffers(4)
    before = bb["kc"].clone()
    bb["pos"].fill_(128); bb["step"].zero_()
    bb["decode"].run()
    torch.cuda.synchronize()
    assert torch.equal(before, bb["kc"])


def test_history_dependent_processor_and_sampling(dev):
    from oracle import mistral_oracle as MO
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, SamplingParams, static_mask_of
    sd = MO.random_state_dict(SMALL, seed=45)
    eng = LLM(model=USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=256))

    def no_repeat(token_ids, logits):                       # depends on the history: not a static mask
        if token_ids:
            logits[token_ids[-1]] = float("-inf")
        logits[0:100] = float("-inf")
        return logits
    assert static_mask_of([no_repeat], 1000, dev) is None
    ids = torch.randint(0, 1000, (30,), generator=torch.Generator().manual_seed(3))
    out = eng.generate(prompt_token_ids=[ids.tolist()], sampling_params=SamplingParams(max_tokens=16, top_k=1, logits_processors=[no_repeat]))
    toks = out[0].outputs[0].token_ids
    assert eng.stats["hook_requests"] == 1 and len(toks) == 16
    # the same rule applied by hand on top of the CPU oracle
    hist = ids.tolist()
    logits, cache = MO.forward(sd, SMALL, ids)
    ref, gaps = [], []
    for _ in range(16):
        lg = no_repeat(hist, logits[-1].clone())
        top2 = torch.topk(lg, 2).values
        gaps.append((top2[0] - top2[1]).item())
        t = int(lg.argmax())
        ref.append(t); hist.append(t)
        logits, cache = MO.forward(sd, SMALL, torch.tensor([t]), cache)
    first = next((i for i in range(16) if toks[i] != ref[i]), None)
    print("history-dependent processor vs oracle: first difference", first)
    assert first is None or gaps[first] <= 2 ** -6 * 8 + 1e-3
    assert all(toks[i] != toks[i - 1] for i in range(1, 16)) and toks[0] != ids[-1].item() and all(t >= 100 for t in toks)
    # sampling knobs: reproducible per seed, different across seeds, never a banned id
    def ban(token_ids, logits):
        logits[0:500] = float("-inf")
        return logits
    sp = lambda seed: SamplingParams(max_tokens=20, temperature=1.3, top_p=0.95, top_k=40, seed=seed, logits_processors=[ban])
    a = eng.generate(prompt_token_ids=[ids.tolist()], sampling_params=sp(1))[0].outputs[0].token_ids
    b = eng.generate(prompt_token_ids=[ids.tolist()], sampling_params=sp(1))[0].outputs[0].token_ids
    c = eng.generate(prompt_token_ids=[ids.tolist()], sampling_params=sp(2))[0].outputs[0].token_ids
    assert a == b and a != c and all(t >= 500 for t in a + c)


def test_continuous_batching_16_slots_matrix_cores_vs_oracle(dev):
    """Round 4: 22 greedy requests with one mask through the SIXTEEN decode slots of the matrix-core batch kernel (usdm_gemv_batch
    form 1; the request lists of src/inference_vllm.py:109-125): ragged prompts, different max_tokens, a stop id, slot turnover
    (22 requests on 16 slots); every sequence equals oracle greedy generation of its prompt under the near-tie rule."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.serving import LLM, MAX_SLOTS, SamplingParams
    assert MAX_SLOTS == 16
    sd = MO.random_state_dict(SMALL, seed=53)
    eng = LLM(model=USDMForCausalLM.from_state_dict(sd, SMALL, dev, ctx_max=256))
    g = torch.Generator().manual_seed(8)

    def ban(token_ids, logits):
        logits[0:250] = float("-inf")
        return logits
    bad = [[i] for i in range(250)]
    n = 22
    lens = [int(v) for v in torch.randint(12, 70, (n,), generator=g)]
    prompts = [torch.randint(0, 1000, (L,), generator=g) for L in lens]
    max_new = [int(v) for v in torch.randint(5, 36, (n,), generator=g)]
    stop = MO.greedy_generate(sd, SMALL, prompts[1], 12, bad_words_ids=bad)[lens[1] + 5]
    sps = [SamplingParams(max_tokens=mn, top_k=1, stop_token_ids=[stop], logits_processors=[ban]) for mn in max_new]
    outs = eng.generate(prompt_token_ids=[p.tolist() for p in prompts], sampling_params=sps)
    firsts = []
    for p, mn, o in zip(prompts, max_new, outs):
        ref, ref_logits = MO.greedy_generate(sd, SMALL, p, mn, bad_words_ids=bad, eos_token_id=stop, return_logits=True)
        toks = o.outputs[0].token_ids
        firsts.append(check_against_oracle(p.tolist() + toks, ref, ref_logits, p.numel()))
        assert len(toks) <= mn and all(t >= 250 for t in toks)
    st = eng.stats
    print("16-slot continuous batching:", st, "first differences:", firsts)
    assert st["batched_requests"] == n and st["admissions"] == n and st["max_active"] == 16
    assert st["batched_steps"] < sum(max_new) // 4
