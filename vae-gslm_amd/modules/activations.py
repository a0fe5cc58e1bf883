"""Activation factory (reference modules/activations.py:5-18).

``hip_act_id`` tells the GEMM epilogue which activations it can fuse
(ReLU, exact-erf GELU, SiLU); anything else runs as a separate torch op."""
import torch.nn as nn

from hparams.hp import Hparams

_FACTORY = {
    "ReLU": lambda hp: nn.ReLU(),
    "SELU": lambda hp: nn.SELU(),
    "GELU": lambda hp: nn.GELU(),
    "LeakyRELU": lambda hp: nn.LeakyReLU(negative_slope=hp.slope),
    "SiLU": lambda hp: nn.SiLU(),
}


def get_activation(hp: Hparams) -> nn.Module:
    try:
        make = _FACTORY[hp.identifier]
    except KeyError:
        raise ValueError(f"{hp.identifier} not in the usable activation function lists.")
    return make(hp)


def hip_act_id(module) -> "str | None":
    """'relu' / 'gelu' / 'none' if the GEMM epilogue can fuse it, else None."""
    if module is None or isinstance(module, nn.Identity):
        return "none"
    if isinstance(module, nn.ReLU):
        return "relu"
    if isinstance(module, nn.GELU) and getattr(module, "approximate", "none") == "none":
        return "gelu"
    if isinstance(module, nn.SiLU):
        return "silu"
    return None
