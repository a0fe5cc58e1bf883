"""Deterministic, key-hashed parameter fill (test infrastructure).

Both the golden generator (which fills the *reference* model) and the tests
(which fill the oracle's / the product's state dict) call :func:`fill_like`
with the same ``seed`` so no weight blobs have to be committed: every tensor
is regenerated from ``(seed, key name, shape)``.

Design notes
------------
* biases and norm offsets are small but non-zero so the bias paths are
  exercised; norm scales are ``1 + 0.1 n``;
* ``token_predictor.linear.weight`` is scaled up so that the arg-max of the
  token logits has a margin far above fp32 re-association noise
  (SURVEY.md D5);
* 1-D schedule buffers of the diffusion decoder are *not* touched (they are
  deterministic functions of the config).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_SKIP_PREFIXES = ("decoder.",)
_SKIP_EXCEPT = ("decoder.model.",)


def _is_buffer(key: str) -> bool:
    """Diffusion schedule buffers live directly under ``decoder.``."""
    return key.startswith(_SKIP_PREFIXES) and not key.startswith(_SKIP_EXCEPT)


def _rng_for(seed: int, key: str) -> np.random.Generator:
    h = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, h]))


def make_tensor(seed: int, key: str, shape: Tuple[int, ...]) -> np.ndarray:
    rng = _rng_for(seed, key)
    n = rng.standard_normal(size=shape, dtype=np.float64)
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if key.startswith("token_predictor."):
            gain = 6.0
        if key.startswith("token_embedding."):
            return (0.5 * n).astype(np.float32)
        out = n * (gain / np.sqrt(max(fan_in, 1)))
    elif leaf in ("scale", "weight"):
        out = 1.0 + 0.1 * n
    else:  # biases
        out = 0.05 * n
    return out.astype(np.float32)


def fill_like(shapes: Iterable[Tuple[str, Tuple[int, ...]]],
              seed: int = 20250620) -> Dict[str, np.ndarray]:
    """Return ``{key: float32 array}`` for every non-buffer key."""
    out = {}
    for key, shape in shapes:
        if _is_buffer(key):
            continue
        out[key] = make_tensor(seed, key, tuple(int(s) for s in shape))
    return out
