"""Bounded caches for launch plans.

Every stage of the path pre-builds a launch plan (argument structs + workspace + hipGraph) per input shape.  Real utterances
nearly always differ in length, so (ADVICE r01 / VERDICT r01 #10):
  * LRU        - a plan cache never holds more than `maxsize` plans; the least recently used plan (its workspace and its
                 hipGraph with it) is dropped first;
  * bucket()   - lengths are rounded up to a multiple so that near-equal lengths share a plan where the kernels take the
                 true length from device memory (Voicebox: kv_len / valid_len masks), or share a WORKSPACE where the true
                 length is baked into the argument structs (BigVGAN, XLS-R);
  * Arena      - a replayable bump allocator: plans built for different exact lengths inside one bucket draw the same buffers
                 in the same order, so a bucket owns ONE workspace however many exact-length plans it has seen.  Plans assume
                 a zero-initialised workspace (padding rows / columns), so the arena is re-zeroed whenever a different plan
                 takes it over (one memset, tens of microseconds, only on a switch).
"""
from collections import OrderedDict

import torch


def bucket(n, q):
    """n rounded up to a multiple of q."""
    return (int(n) + q - 1) // q * q


class LRU:
    def __init__(self, maxsize):
        self.maxsize, self.d = int(maxsize), OrderedDict()
        self.evictions = 0

    def get(self, key):
        v = self.d.get(key)
        if v is not None:
            self.d.move_to_end(key)
        return v

    def put(self, key, value):
        self.d[key] = value
        self.d.move_to_end(key)
        while len(self.d) > self.maxsize:
            self.d.popitem(last=False)
            self.evictions += 1
        return value

    def get_or_build(self, key, build):
        v = self.get(key)
        if v is None:
            v = self.put(key, build())
        return v

    def clear(self):
        self.d.clear()

    def __len__(self):
        return len(self.d)

    def __contains__(self, key):
        return key in self.d

    def __iter__(self):
        return iter(self.d)

    def keys(self):
        return self.d.keys()

    def values(self):
        return self.d.values()


class Arena:
    """Replayable bump allocator over device tensors (see the module docstring)."""

    def __init__(self, device):
        self.device, self.bufs, self.i, self.owner = device, [], 0, None
        self.retired = []     # buffers a later build replaced: earlier plans hold raw POINTERS into them, so they stay alive here

    def begin(self):
        self.i = 0
        return self

    def zeros(self, *shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= int(s)
        if self.i < len(self.bufs) and self.bufs[self.i].dtype == dtype and self.bufs[self.i].numel() >= n:
            b = self.bufs[self.i]
        else:
            b = torch.zeros(max(n, 1), dtype=dtype, device=self.device)
            if self.i < len(self.bufs):
                # a build that asks for more than the reservation (or another dtype): plans built before keep replaying on the old
                # buffer through the raw pointers in their argument structs, so the arena keeps it alive (and zeroes it in take())
                self.retired.append(self.bufs[self.i])
                self.bufs[self.i] = b
            else:
                self.bufs.append(b)
        self.i += 1
        return b[:n].view(*shape)

    def take(self, owner):
        """Called before a plan runs: a different plan used the workspace last -> restore the all-zero initial condition."""
        if self.owner is not owner:
            if self.owner is not None:
                for b in self.bufs + self.retired:
                    b.zero_()
            self.owner = owner

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self.bufs + self.retired)
