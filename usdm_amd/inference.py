"""Drop-in for the reference CLI module src/inference.py: same helper names, same `sample(...)`
signature and control flow (ASR round -> text round -> unit round -> reconstruct_speech -> wav).

Differences forced by the environment: models are passed in / loaded from local paths (no hub), and
librosa is absent, so audio loading uses scipy (polyphase resampling to 16 kHz; host I/O outside the hot path).
"""
import argparse
import re

import numpy as np
import torch
from scipy.io.wavfile import read as wav_read
from scipy.io.wavfile import write
from scipy.signal import resample_poly

from .voicebox.util.model_util import initialize_decoder, reconstruct_speech  # noqa: F401

device = torch.device("cuda")


def default_template(user_unit, user_text=None, agent_text=None):
    """Unified template for the ASR, T2T and TTS stages (src/inference.py:16-27)."""
    template = (
        "Below is a conversation between the user and the agent. Each turn includes the user's speech and its corresponding transcript, "
        "along with the agent's response text and the corresponding speech.\n"
        "\n### User\n"
        f"{user_unit}<|correspond|>"
    )
    if user_text:
        template += f"{user_text}\n### Agent\n"
    if agent_text:
        template += f"{agent_text}<|correspond|>"
    return template


def strip_exact_multiple(text, patterns):
    """src/inference.py:31-37."""
    for pattern in patterns:
        if text.startswith(pattern):
            text = text[len(pattern):]
        if text.endswith(pattern):
            text = text[:-len(pattern)]
    return text


def generate_bad_words_ids(start, end, exclude=[]):
    """src/inference.py:41-45 (note: `pop` removes by INDEX; correct only because start == 0 whenever exclude is used)."""
    bad_words_ids = [[token_id] for token_id in range(start, end)]
    for e in exclude:
        bad_words_ids.pop(e)
    return bad_words_ids


def load_audio_16k(path):
    sr, data = wav_read(path)
    x = data.astype(np.float32)
    if data.dtype == np.int16:
        x /= 32768.0
    elif data.dtype == np.int32:
        x /= 2147483648.0
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != 16000:
        g = np.gcd(16000, sr)
        x = resample_poly(x, 16000 // g, sr // g).astype(np.float32)
    return x


_BAD = {}


def _bad(name, *a, **k):
    if name not in _BAD:
        _BAD[name] = generate_bad_words_ids(*a, **k)
    return _BAD[name]


@torch.inference_mode()
def sample(user_path, reference_path, model, unit_extractor, voicebox, vocoder, tokenizer, output_path,
           reference_mel=None, n_timesteps=50):
    """src/inference.py:48-89."""
    bad_words_ids_unit2text = _bad("u2t", 32000, 42003)
    bad_words_ids_text2text = _bad("t2t", 32002, 42003)
    bad_words_ids_text2unit = _bad("t2u", 0, 32002, exclude=[28705])
    pattern = re.compile(r"<\|unit(\d+)\|>")

    user_wav = load_audio_16k(user_path) if isinstance(user_path, str) else user_path
    user_unit = ''.join([f'<|unit{i}|>' for i in unit_extractor.predict(torch.as_tensor(user_wav, dtype=torch.float32).to(device), 35 - 1).cpu().tolist()])

    def run(model_input, bad, eos):
        ids = torch.LongTensor(tokenizer(model_input).input_ids).to(device).unsqueeze(0)
        return model.generate(input_ids=ids, max_length=tokenizer.model_max_length, do_sample=True, bad_words_ids=bad,
                              top_p=1.0, top_k=1, temperature=1.0, eos_token_id=eos)

    outputs = run(default_template(user_unit=user_unit), bad_words_ids_unit2text, tokenizer("\n").input_ids[-1])
    user_text = strip_exact_multiple(tokenizer.decode(outputs[0]).split("<|correspond|>")[-1], ["\n", " "])
    outputs = run(default_template(user_unit=user_unit, user_text=user_text), bad_words_ids_text2text,
                  tokenizer("<|correspond|>").input_ids[-1])
    agent_text = strip_exact_multiple(tokenizer.decode(outputs[0]).split("\n")[-1], ["\n", " ", "<|correspond|>"])
    outputs = run(default_template(user_unit=user_unit, user_text=user_text, agent_text=agent_text), bad_words_ids_text2unit, 28705)
    agent_unit = tokenizer.decode(outputs[0]).split("<|correspond|>")[-1]

    matches = [int(x) for x in pattern.findall(agent_unit)]
    agent_unit = torch.LongTensor(matches).to(device)
    reference_unit = None
    if reference_mel is not None:
        if reference_path is None:
            raise ValueError("reference_mel needs reference_path too (the prompt's unit ids are extracted from that audio)")
        ref_wav = load_audio_16k(reference_path)
        reference_unit = unit_extractor.predict(torch.as_tensor(ref_wav, dtype=torch.float32).to(device), 35 - 1)
        reference_path = None
    audio = reconstruct_speech(agent_unit, device, reference_path, unit_extractor, voicebox, vocoder, n_timesteps=n_timesteps,
                               reference_mel=reference_mel, reference_unit=reference_unit)
    write(output_path, vocoder.h.sampling_rate, audio)
    return audio


def load_models(model_cache_dir, dev=None, ctx_max=None):
    """What the reference's __main__ does between argument parsing and sample() (src/inference.py:105-129), from LOCAL copies:
    hub names resolve inside model_cache_dir (huggingface_hub cache layout or plain <name>/ directories, checkpoints.py)."""
    import os
    from .checkpoints import resolve_local
    from .llm import USDMForCausalLM
    from .unit_extractor import UnitExtractor
    dev = torch.device(dev or device)
    # Load voicebox, vocoder configuration and checkpoint
    voicebox, vocoder = initialize_decoder(model_cache_dir, dev)
    # Load unit extractor (speech tokenizer): the reference's own call, resolved inside the cache directory
    os.environ.setdefault("USDM_MODEL_CACHE_DIR", model_cache_dir)
    unit_extractor = UnitExtractor("xlsr2_1b_v2", "https://dl.fbaipublicfiles.com/seamlessM4T/models/unit_extraction/kmeans_10k.npy",
                                   device=dev, cache_dir=model_cache_dir)
    # Load USDM model and tokenizer
    llm_dir = resolve_local(model_cache_dir, "naver-ai/USDM-DailyTalk", must_contain=("config.json",))
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(llm_dir, local_files_only=True)
    # the reference generates up to tokenizer.model_max_length (inference.py:64); beyond the 4096-token sliding window the attention
    # kernels bound their key range (llm.py), so the cache may be longer than the window.  8192 = the USDM tokenizer's limit.
    want = ctx_max or min(int(getattr(tokenizer, "model_max_length", 4096) or 4096), 8192)
    model = USDMForCausalLM.from_pretrained(llm_dir, device=dev, torch_dtype=torch.bfloat16, ctx_max=want).to(dev).eval()
    return model, unit_extractor, voicebox, vocoder, tokenizer


def main(argv=None):
    global device
    parser = argparse.ArgumentParser()
    parser.add_argument('--input_path', type=str, required=True,
                        help="Path to the input file containing the speech data to process.")
    parser.add_argument('--reference_path', type=str, default=None,
                        help="Path to the reference audio file for speaker adaptation (optional). If not provided, the model will perform speaker unconditional generation.")
    parser.add_argument('--model_cache_dir', type=str, required=True,
                        help="Directory holding the model checkpoints (the reference's download cache, or local <name>/ directories).")
    parser.add_argument('--output_path', type=str, required=True,
                        help="Path to save the spoken response.")
    args = parser.parse_args(argv)

    device = torch.device("cuda")
    model, unit_extractor, voicebox, vocoder, tokenizer = load_models(args.model_cache_dir, device)
    try:
        sample(args.input_path, args.reference_path, model, unit_extractor, voicebox, vocoder, tokenizer, args.output_path)
    except Exception as e:       # the reference swallows sampling errors the same way (src/inference.py:131-134)
        print(f"Error while sampling: {e}")
        return 1
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
