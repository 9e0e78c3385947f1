"""Oracle: integer/host pieces of the path, restated with plain Python loops (TEST INFRASTRUCTURE).

  process_unit            src/decoder/voicebox/util/model_util.py:50-54
  generate_bad_words_ids  src/inference.py:41-45 (+ the three calls at :51-53)
Pinned by tests/golden/process_unit.npz (reference function run in the build container).
"""


def process_unit(units, sampling_rate=22050, hop_size=256, token_sr=50):
    """repeat_interleave(sr//50) -> truncate to a multiple of hop -> per-frame mode (ties -> smallest id)."""
    rep = sampling_rate // token_sr
    n = len(units) * rep
    new_length = n // hop_size * hop_size
    out = []
    for f in range(new_length // hop_size):
        counts = {}
        for i in range(f * hop_size, (f + 1) * hop_size):
            u = int(units[i // rep])
            counts[u] = counts.get(u, 0) + 1
        best = max(counts.values())
        out.append(min(u for u, c in counts.items() if c == best))
    return out, new_length


def banned_ranges(stage):
    """Token-id ranges masked in each of the three generate() rounds (inference.py:51-53)."""
    if stage == "unit2text":
        return [(32000, 42003)]
    if stage == "text2text":
        return [(32002, 42003)]
    if stage == "text2unit":
        return [(0, 28705), (28706, 32002)]
    raise ValueError(stage)
