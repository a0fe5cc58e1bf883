"""Pair carried through the coupling stack: the running tensor and the log-determinant accumulated so far
(the reference's ``TensorLogdet``; ``logdet`` starts as the float 0.0 and becomes a (B, T, k) tensor)."""
import collections

TensorLogdet = collections.namedtuple("TensorLogdet", ["tensor", "logdet"])


def add_logdet(pair: "TensorLogdet", tensor, term) -> "TensorLogdet":
    """New pair with ``tensor`` and ``pair.logdet + term``."""
    return TensorLogdet(tensor, pair.logdet + term)
