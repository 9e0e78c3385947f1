"""vLLM-style serving surface of the speech-text LLM (reference: src/inference_vllm.py:42-66,70-83,109-125 and the demo's
knobs, src/streamlit_demo.py:201-211,258-287): `LLM(model=..., download_dir=...)`, `SamplingParams(max_tokens, top_p, top_k,
temperature, stop_token_ids, logits_processors)`, `llm.generate(prompts, sampling_params) -> [RequestOutput]` with
`.outputs[0].text / .token_ids`, on top of usdm_amd.llm.USDMForCausalLM.

How requests are executed
  * logits processors that are STATIC masks (the reference's three `bad_word_processor_*` only write -inf into fixed id ranges)
    are recognised by probing and turned into the device-side ban mask: banned lm_head rows are not even streamed, and the
    decode step stays one hipGraph;
  * several requests that share a mask - greedy (top_k = 1 / temperature = 0, what the reference passes) AND sampled - are served by
    CONTINUOUS BATCHING over the 16 sequence slots of the batched decode (usdm_gemv_batch: the weights are streamed once per step
    for all slots; 4 slots for groups of <= 4 requests): a finished sequence's slot is refilled from the queue at the next scheduling point (every 8 steps) while
    the other slots keep decoding.  Sampling state is PER SLOT (usdm_sample_final's batched form: every slot has its own
    temperature / top-k / top-p / seed block on the device and its own Philox counter), so a request sampled inside a batch
    returns exactly the tokens it returns alone; a greedy request in a sampled batch carries top_k = 1;
  * a single request uses the single-sequence graph (usdm_sample_final: temperature / top-k / top-p on the device, `seed` per
    request);
  * processors that really depend on the token history run as Python between the lm_head launch and the pick of every step
    (eager launches: correct, not fast).
"""
import os
from collections import deque

import torch

from .graph import GraphedPlan, GraphedSegments

MAX_SLOTS = 16      # usdm_gemv_batch streams the weights once per step for up to 16 sequences (matrix-core form above 4)
SMALL_SLOTS = 4     # groups of <= 4 requests use the 4-slot plan (VALU form: per slot bit-identical with the single-request path)
CHUNK = 8           # decode steps between two scheduling points (host sync: stop checks, slot turnover)


class SamplingParams:
    """The subset of vllm.SamplingParams the reference and its demo use (inference_vllm.py:109-123)."""

    def __init__(self, n=1, temperature=1.0, top_p=1.0, top_k=-1, max_tokens=16, min_tokens=0, stop_token_ids=None,
                 logits_processors=None, seed=None, skip_special_tokens=True, ignore_eos=False, static_logits_mask=None, **unused):
        if not isinstance(n, int) or n < 1:
            raise ValueError("n must be a positive integer")
        if temperature < 0 or not (0 < top_p <= 1) or (top_k < -1 or top_k == 0):
            raise ValueError("temperature >= 0, 0 < top_p <= 1, top_k = -1 (off) or >= 1")
        self.n, self.temperature, self.top_p, self.top_k = n, float(temperature), float(top_p), int(top_k)
        self.max_tokens, self.min_tokens = int(max_tokens), int(min_tokens)
        self.stop_token_ids = list(stop_token_ids or [])
        self.logits_processors = list(logits_processors or [])
        self.seed, self.skip_special_tokens, self.ignore_eos = seed, skip_special_tokens, ignore_eos
        self.static_logits_mask = static_logits_mask      # None: probe the processors; True / False: caller's word
        if self.n > 1 and self.greedy:
            raise ValueError("n must be 1 when using greedy sampling (the n completions would be identical)")   # as vllm

    @property
    def greedy(self):
        return self.temperature == 0.0 or self.top_k == 1


class CompletionOutput:
    def __init__(self, index, text, token_ids, finish_reason, stop_reason=None):
        self.index, self.text, self.token_ids, self.finish_reason, self.stop_reason = index, text, token_ids, finish_reason, stop_reason
        self.cumulative_logprob, self.logprobs = None, None

    def __repr__(self):
        return f"CompletionOutput(index={self.index}, text={self.text!r}, token_ids={self.token_ids}, finish_reason={self.finish_reason})"


class RequestOutput:
    def __init__(self, request_id, prompt, prompt_token_ids, outputs):
        self.request_id, self.prompt, self.prompt_token_ids, self.outputs, self.finished = request_id, prompt, prompt_token_ids, outputs, True

    def __repr__(self):
        return f"RequestOutput(request_id={self.request_id}, outputs={self.outputs})"


def static_mask_of(processors, vocab, device):
    """If the processors only ever write -inf into a FIXED set of ids (and leave every other logit alone), return that set as a
    0/1 uint8 mask [vocab]; else None.  Decided by probing with several histories / logit vectors (a heuristic: a processor that
    reacts only to some particular history can pass; SamplingParams(static_logits_mask=False) forces the general path)."""
    if not processors:
        return torch.zeros(vocab, dtype=torch.uint8, device=device)
    g = torch.Generator().manual_seed(0)
    hists = [[], [vocab - 1], [5, vocab // 2, 3, vocab // 2], [vocab // 3] * 9 + [vocab - 2], list(range(7, min(vocab, 39)))]
    probes = [(h, torch.zeros(vocab) if i == 0 else torch.randn(vocab, generator=g)) for i, h in enumerate(hists)]
    masks = []
    for hist, lg in probes:
        out = lg.clone().to(device)
        for p in processors:
            out = p(list(hist), out)
        if not torch.is_tensor(out) or out.shape != (vocab,):
            return None
        out = out.float().cpu()
        m = torch.isneginf(out)
        if not torch.equal(out[~m], lg[~m]):
            return None
        masks.append(m)
    if any(not torch.equal(masks[0], m) for m in masks[1:]):
        return None
    return masks[0].to(torch.uint8).to(device)


class LLM:
    """`LLM(model='naver-ai/USDM-DailyTalk', download_dir=cache)` as in inference_vllm.py:105 (hub names resolve inside
    download_dir), or `LLM(model=<USDMForCausalLM>, tokenizer=<tokenizer>)` around objects that are already loaded."""

    def __init__(self, model, tokenizer=None, download_dir=None, gpu_memory_utilization=0.9, max_model_len=None, device="cuda",
                 dtype="bfloat16", max_num_seqs=MAX_SLOTS, **unused):
        # max_num_seqs (vllm's name): sequences decoded per step, 1..16.  <= 4: the VALU batch kernel (per slot bit-identical with the
        # single-request path); above: the matrix-core form (usdm_gemv_batch form 1)
        self.max_slots = max(1, min(int(max_num_seqs), MAX_SLOTS))
        from .llm import USDMForCausalLM
        if isinstance(model, USDMForCausalLM):
            self.llm = model
        else:
            from .checkpoints import resolve_local
            path = model if os.path.isdir(str(model)) else resolve_local(download_dir or ".", model, must_contain=("config.json",))
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(path, local_files_only=True)
            ctx = min(int(max_model_len or getattr(tokenizer, "model_max_length", 4096) or 4096), 8192)
            self.llm = USDMForCausalLM.from_pretrained(path, device=device, ctx_max=ctx)
        if isinstance(tokenizer, str):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer, local_files_only=True)
        self.tokenizer = tokenizer
        self.stats = dict(requests=0, batched_requests=0, admissions=0, max_active=0, batched_steps=0, hook_requests=0)
        self._next_id = 0

    def get_tokenizer(self):
        return self.tokenizer

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def generate(self, prompts=None, sampling_params=None, prompt_token_ids=None, use_tqdm=False):
        if prompts is None and prompt_token_ids is None:
            raise ValueError("prompts or prompt_token_ids is required")
        if isinstance(prompts, str):
            prompts = [prompts]
        if prompt_token_ids is not None and prompt_token_ids and isinstance(prompt_token_ids[0], int):
            prompt_token_ids = [prompt_token_ids]
        n = len(prompts) if prompts is not None else len(prompt_token_ids)
        sps = sampling_params if isinstance(sampling_params, (list, tuple)) else [sampling_params or SamplingParams()] * n
        if len(sps) != n:
            raise ValueError("one SamplingParams per prompt (or a single one for all)")
        dev, V = self.llm.device, self.llm.cfg["vocab_size"]
        reqs = []
        for i in range(n):
            if prompt_token_ids is not None:
                ids = list(prompt_token_ids[i])
                text = prompts[i] if prompts is not None else None
            else:
                if self.tokenizer is None:
                    raise ValueError("text prompts need a tokenizer")
                text = prompts[i]
                ids = list(self.tokenizer(text).input_ids)
            sp = sps[i]
            mask = sp.static_logits_mask
            if mask is None or mask is True:
                m = static_mask_of(sp.logits_processors, V, dev)
                if m is None and mask is True:
                    raise ValueError("static_logits_mask=True but the processors are not a fixed -inf mask")
                mask = m
            else:
                mask = None if sp.logits_processors else torch.zeros(V, dtype=torch.uint8, device=dev)
            stops = set(sp.stop_token_ids)
            eos = getattr(self.tokenizer, "eos_token_id", None)
            if eos is not None and not sp.ignore_eos:
                stops.add(int(eos))
            room = self.llm.ctx_max - len(ids)
            seed = sp.seed
            if not sp.greedy and seed is None:      # as generate(): a fresh stream per call, reproducible under torch.manual_seed
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            # n > 1 (vllm.SamplingParams.n): the request fans out into n sequences that share prompt, knobs and mask and draw from
            # the Philox streams seed, seed + 1, ...: completion j is exactly what a single request with seed + j returns, and the
            # copies ride the continuous batch like any other requests (the prompt is prefilled once per copy)
            for j in range(sp.n):
                reqs.append(dict(i=(i, j), rid=str(self._next_id + i), text=text, ids=ids, sp=sp, mask=mask, stops=stops,
                                 seed=(seed + j) if seed is not None else None,
                                 max_new=max(0, min(sp.max_tokens, room, self.llm.max_out))))
        self._next_id += n
        self.stats["requests"] += n
        done = {}
        # continuous batching: requests with the same static mask (greedy and sampled alike), at least two of them
        groups = {}
        for r in reqs:
            # (tensor parallel: greedy requests only - the sampling kernel needs the full logit row on one GPU; every rank serves the
            # same request list and sees the same tokens, so all ranks take the same scheduling decisions)
            if r["mask"] is not None and r["max_new"] > 0 and (not self.llm.tp_path or r["sp"].greedy):
                groups.setdefault(bytes(r["mask"].cpu().numpy().tobytes()), []).append(r)
        for grp in groups.values():
            if len(grp) >= 2:
                for r, toks, why in self._run_batched(grp):
                    done[r["i"]] = (toks, why)
        for r in reqs:
            if r["i"] not in done:
                done[r["i"]] = self._run_single(r)
        outs = []
        for r in reqs:
            if r["i"][1] == 0:
                outs.append(RequestOutput(r["rid"], r["text"], r["ids"], []))
            outs[-1].outputs.append(self._finish(r, *done[r["i"]]))
        return outs

    # ------------------------------------------------------------------ one request on the single-sequence graph
    def _run_single(self, r):
        sp, llm = r["sp"], self.llm
        if r["max_new"] <= 0:
            return [], "length"
        ids = torch.tensor([r["ids"]], dtype=torch.long, device=llm.device)
        kw = dict(input_ids=ids, max_new_tokens=r["max_new"], eos_token_id=sorted(r["stops"]) or None, min_new_tokens=sp.min_tokens)
        sampled = not sp.greedy
        if sampled:
            kw.update(do_sample=True, temperature=sp.temperature, top_p=sp.top_p, top_k=(sp.top_k if sp.top_k > 0 else None), seed=r["seed"])
        if r["mask"] is not None:
            out = llm.generate(ban_mask=r["mask"], **kw)
        else:       # history-dependent processors: Python between the lm_head launch and the pick of every step
            self.stats["hook_requests"] += 1
            prompt = list(r["ids"])

            def hook():
                step = int(llm.st_step.item())
                hist = prompt + llm.st_out[:step].tolist()
                lg = llm.last_logits
                for p in sp.logits_processors:
                    lg = p(hist, lg)
                if lg is not llm.last_logits:
                    llm.last_logits.copy_(lg)
            if not sampled:
                kw.update(do_sample=False)
            out = llm.generate(_logits_hook=hook, **kw)
        toks = out[0, len(r["ids"]):].tolist()
        why = "stop" if (toks and toks[-1] in r["stops"] and len(toks) >= sp.min_tokens) else "length"
        return toks, why

    # ------------------------------------------------------------------ continuous batching over the decode slots
    def _run_batched(self, grp):
        llm = self.llm
        from . import ops
        nslots = min(self.max_slots, llm.max_batch(), SMALL_SLOTS if len(grp) <= SMALL_SLOTS else MAX_SLOTS)
        bb = llm._batch_buffers(nslots)
        sampled = any(not r["sp"].greedy for r in grp)       # one sampled request -> the whole group runs on the sampling graph
        key = "decode_sampled" if sampled else "decode"
        if bb[key] is None:
            built = llm._build_decode_batch(nslots, sampling=sampled)
            bb[key] = GraphedSegments(built, llm._run_segs) if isinstance(built, list) else GraphedPlan(built)
        decode = bb[key]
        self.stats["sampled_in_batch"] = self.stats.get("sampled_in_batch", 0) + sum(not r["sp"].greedy for r in grp)
        for b in range(nslots):                               # idle slots: harmless greedy knobs
            ops.set_sample_params(bb["sp"][b], 1.0, 1, 1.0, 0)
        llm.ban.copy_(grp[0]["mask"][llm.v0:llm.v1])
        queue, slots, results = deque(grp), [None] * nslots, []
        self.stats["batched_requests"] += len(grp)
        while queue or any(s is not None for s in slots):
            for b in range(nslots):                               # admit: prefill the prompt into the free slot's cache
                if slots[b] is None and queue:
                    r = queue.popleft()
                    L = len(r["ids"])
                    bb["step"][b] = 0
                    bb["pos"][b] = L
                    sp = r["sp"]
                    if sp.greedy:
                        ops.set_sample_params(bb["sp"][b], 1.0, 1, 1.0, 0)
                    else:
                        ops.set_sample_params(bb["sp"][b], sp.temperature, max(sp.top_k, 0), sp.top_p, r["seed"])
                    # the first token is picked by the prefill: sampled too when the group runs on the sampling graph
                    segs, io = bb["prefill"].get_or_build((L, b, sampled), lambda: llm._build_prefill(L, True if sampled else None, slot=bb["slots"][b]))
                    io["ids"].copy_(torch.tensor(r["ids"], dtype=torch.long))
                    llm._run_segs(segs)                               # (+ first token)
                    slots[b] = dict(r=r, produced=1)
                    self.stats["admissions"] += 1
            active = [b for b in range(nslots) if slots[b] is not None]
            self.stats["max_active"] = max(self.stats["max_active"], len(active))
            for b in range(nslots):                               # idle slots decode garbage into row 0 of their own cache
                if slots[b] is None:
                    bb["step"][b] = 0
                    bb["pos"][b] = 0
            # tokens already known (the prefill's first token) are checked before any further step is spent
            toks = bb["out"].tolist()                                # host sync = scheduling point
            need, freed = 0, False
            for b in active:
                s, r = slots[b], slots[b]["r"]
                seq = toks[b][:s["produced"]]
                end = next((i + 1 for i, t in enumerate(seq) if t in r["stops"] and i + 1 >= r["sp"].min_tokens), None)
                if end is not None or s["produced"] >= r["max_new"]:
                    n = end if end is not None else r["max_new"]
                    results.append((r, seq[:n], "stop" if end is not None and end <= n else "length"))
                    slots[b] = None
                    freed = True
                else:
                    need = max(need, 1)
            if not need or (freed and queue):
                continue                                             # refill the freed slot(s) before spending more steps
            live = [b for b in range(nslots) if slots[b] is not None]
            # never more steps than the slot with the FEWEST tokens left: batch slots have no device-side `done` word, so a
            # slot would otherwise keep decoding to the end of the chunk - past its max_new, and for a request that runs to the
            # context limit (the reference passes max_tokens = model_max_length) past ctx_max, i.e. into the next head's / the
            # next slot's cache rows.  usdm_attn_decode additionally refuses positions >= ctx_max.
            n = min(CHUNK, min(slots[b]["r"]["max_new"] - slots[b]["produced"] for b in live))
            for _ in range(n):
                decode.run()
            self.stats["batched_steps"] += n
            for b in live:
                slots[b]["produced"] = min(slots[b]["produced"] + n, slots[b]["r"]["max_new"])
        return results

    # ------------------------------------------------------------------ detokenise
    def _finish(self, r, toks, why):
        text = ""
        if self.tokenizer is not None:
            text = self.tokenizer.decode(toks, skip_special_tokens=r["sp"].skip_special_tokens)
        stop_reason = toks[-1] if (why == "stop" and toks) else None
        return CompletionOutput(r["i"][1], text, toks, why, stop_reason)
