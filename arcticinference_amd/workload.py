"""Seeded synthetic token streams for the spec-decode hot path.

No dataset or tokenizer travels to the GPU box, so every test / bench draws its
prompts and ground-truth responses from this source: an order-2 Markov chain
over a Zipf(1.1) vocabulary with planted motifs that re-occur, so that suffix
matching finds non-trivial match lengths (uniform-random ids would never match).
Responses come from the same source as prompts, which lets acceptance be scored
model-free exactly the way the reference's simulator does
(/root/reference/arctic_inference/common/suffix_cache/simulator.py:70-90).

Pure numpy; deterministic for a given (seed, request index) on every machine.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

LLAMA3_VOCAB = 128256


@dataclass
class TokenSource:
    vocab_size: int = LLAMA3_VOCAB
    seed: int = 0
    zipf_a: float = 1.1
    n_motifs: int = 64
    motif_min: int = 16
    motif_max: int = 64
    p_motif: float = 0.3
    p_markov: float = 0.6

    def __post_init__(self):
        rng = np.random.default_rng(self.seed)
        ranks = np.arange(1, self.vocab_size + 1, dtype=np.float64)
        w = ranks ** (-self.zipf_a)
        self._cdf = np.cumsum(w / w.sum())
        # a fixed permutation so that frequent tokens are not the small ids
        self._perm = rng.permutation(self.vocab_size).astype(np.int64)
        lens = rng.integers(self.motif_min, self.motif_max + 1, size=self.n_motifs)
        self._motifs = [self._zipf(rng, int(n)) for n in lens]

    def _zipf(self, rng: np.random.Generator, n: int) -> np.ndarray:
        u = rng.random(n)
        return self._perm[np.searchsorted(self._cdf, u, side="left").clip(0, self.vocab_size - 1)]

    def _markov_next(self, a: int, b: int) -> int:
        # deterministic successor of the bigram (a, b): an order-2 transition
        h = (a * 1000003 + b * 999983 + 12345) % 2147483647
        u = (h % 1000003) / 1000003.0
        return int(self._perm[min(int(np.searchsorted(self._cdf, u, side="left")), self.vocab_size - 1)])

    def stream(self, n: int, request: int, salt: int = 0) -> np.ndarray:
        """`n` tokens for request index `request` (independent RNG per request)."""
        rng = np.random.default_rng([self.seed, request, salt])
        out: List[int] = []
        while len(out) < n:
            if rng.random() < self.p_motif:
                m = self._motifs[int(rng.integers(self.n_motifs))]
                cut = int(rng.integers(len(m) // 2, len(m) + 1))
                out.extend(int(t) for t in m[:cut])
            else:
                run = int(rng.integers(8, 33))
                fresh = self._zipf(rng, run)
                coin = rng.random(run)
                for j in range(run):
                    if len(out) >= 2 and coin[j] < self.p_markov:
                        out.append(self._markov_next(out[-2], out[-1]))
                    else:
                        out.append(int(fresh[j]))
        return np.asarray(out[:n], dtype=np.int32)

    def request(self, request: int, prompt_len: int, gen_len: int) -> Tuple[np.ndarray, np.ndarray]:
        s = self.stream(prompt_len + gen_len, request)
        return s[:prompt_len].copy(), s[prompt_len:].copy()
