"""Small helpers the training step uses (reference utils/helpers.py:15-32,
80-161).  Audio cropping / MFCC / spec-augment utilities of the reference are
out of scope for the hot path and are not provided."""
from __future__ import annotations

import re
from pathlib import Path
from typing import Any, Mapping

import torch

from .tensormask import TensorMask


def move_data_to_device(batch: Any, device) -> Any:
    """Recursively move tensors / TensorMasks (non-blocking to accelerators)."""
    device = torch.device(device) if isinstance(device, str) else device
    nb = device.type != "cpu"
    if isinstance(batch, (torch.Tensor, TensorMask)):
        return batch.to(device, non_blocking=nb)
    if isinstance(batch, Mapping):
        return type(batch)((k, move_data_to_device(v, device)) for k, v in batch.items())
    if isinstance(batch, (list, tuple)):
        return type(batch)(move_data_to_device(v, device) for v in batch)
    return batch


def get_padding(kernel_size, dilation=1, stride=1, causal=False, future=False):
    """'same' padding for odd kernels; all on one side for causal / look-ahead convs."""
    p = int(((kernel_size - 1) * dilation + 1 - stride) / 2)
    if causal:
        return (2 * p, 0)
    if future:
        return (0, 2 * p)
    return p


def make_padding_mask(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Key-side padding mask [B, Tq, Tk]; ``a`` only supplies Tq."""
    return b.unsqueeze(-2).expand(-1, a.size(1), -1)


def repeat_batch(x: TensorMask, n: int) -> TensorMask:
    return TensorMask(x.value.repeat_interleave(n, 0), x.mask.repeat_interleave(n, 0))


def pad_to_max_length(items):
    """Collate a list of ``{name: tensor}`` samples into right-padded
    ``TensorMask`` batches (time on axis 0 of every sample tensor)."""
    out = {}
    for key in items[0].keys():
        vals = [it[key] for it in items]
        if not isinstance(vals[0], torch.Tensor) or vals[0].dim() == 0:
            out[key] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else vals
            continue
        tmax = max(v.shape[0] for v in vals)
        lens = torch.tensor([v.shape[0] for v in vals])
        padded = torch.stack([torch.nn.functional.pad(v, [0, 0] * (v.dim() - 1) + [0, tmax - v.shape[0]])
                              for v in vals])
        out[key] = TensorMask.fromlength(padded, lens)
    return out


def get_last_ckpt(directory: str) -> str:
    def step_of(p: Path) -> int:
        m = re.findall(r"step=(\d+)", p.stem)
        if not m:
            raise ValueError(f"Checkpoint {p} is does not contain steps...")
        return int(m[0])
    return sorted(Path(directory).glob("*-cpt.ckpt"), key=step_of)[-1]
