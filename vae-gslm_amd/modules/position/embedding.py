"""Positional-encoding factory (reference modules/position/embedding.py:9-40).

Only ALiBi is live in vae-gslm.yaml and is the one with a HIP kernel (folded
into attention).  The reference matches the misspelt name "Rotery" for its
rotary class, which no config uses and whose forward treats activations as
positions; this build keeps the name lookup but has no rotary
kernel, so asking for it raises ``NotImplementedError`` instead of silently
running a dead path.  T5 bucketed bias needs a dense (H,T,T) table, which the
in-kernel bias path does not take.
"""
from typing import Optional

from hparams.hp import Hparams

from .absolute import SinCos
from .alibi import ALiBi
from .rotary import Rotary  # noqa: F401  (importable; no attention kernel consumes it)
from .t5 import T5RPE  # noqa: F401


def get_positional_encoding(name: str, hp: Hparams, ndim: Optional[int] = None,
                            nheads: Optional[int] = None):
    if name == "ALiBi":
        assert nheads is not None
        return ALiBi(nheads, hp.get("maxpos", 10000))
    if name == "SinCos":
        assert ndim is not None
        return SinCos(ndim, hp.get("maxpos", 10000), hp.get("fixed_pos", False), hp.get("scaled", False))
    if name in ("Rotery", "Rotary", "T5RPE"):
        raise NotImplementedError(
            f"{name}: no MI355X kernel -- the reference ships no config that uses it "
            "(rotary is dead code there, SURVEY.md D3)")
    raise ValueError(f"{name} is not a valid PE type.")
