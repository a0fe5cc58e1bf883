"""``TensorMask`` -- the boundary type of every module on the hot path.

Same public surface as the reference container (utils/tensormask.py:7-228 of
b04901014/vae-gslm): a tensor whose axis 1 (or 2, for B,C,T) is time, bound to
a boolean ``mask[B, T]`` with True = valid frame.  Two MI355X-specific
additions:

* ``lengths32`` -- the int32 ``[B]`` device tensor the HIP kernels take as
  their row predicate (``t < lengths[b]``).  It is computed once per mask
  tensor and cached on it, so the 16 layers of a stack share one reduction.
* masks are assumed to be right-padded prefix masks (the only kind the
  reference's collate produces: utils/helpers.py:119-123); the HIP kernels
  rely on that.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import torch

Number = Union[int, float]


def _broadcastable(mask: torch.Tensor, ndim: int) -> torch.Tensor:
    return mask.reshape(mask.shape + (1,) * (ndim - mask.dim()))


class TensorMask(object):
    __slots__ = ("value", "mask", "axis")

    def __init__(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None, axis: int = 1) -> None:
        if mask is None:
            t = x.shape[1] if axis == 1 else x.shape[2]
            mask = torch.ones((x.shape[0], t), dtype=torch.bool, device=x.device)
            mask._vg_full = True          # remembered: no padding at all
        assert mask.dim() == 2
        assert axis in (1, 2), "Only Support B T ..., B C T"
        if axis == 1:
            assert tuple(x.shape[:2]) == tuple(mask.shape)
        else:
            assert (x.shape[0], x.shape[2]) == tuple(mask.shape)
        self.value, self.mask, self.axis = x, mask, axis

    # ------------------------------------------------------------ construction
    @classmethod
    def fromlength(cls, x: torch.Tensor, length: torch.Tensor, axis: int = 1) -> "TensorMask":
        steps = torch.arange(x.shape[axis], device=x.device)
        return cls(x, steps.unsqueeze(0) < length.unsqueeze(1), axis)

    @classmethod
    def use_mask(cls, x: torch.Tensor, mask: torch.Tensor, mask_value: Number = 0) -> torch.Tensor:
        return cls(x, mask).apply_mask(mask_value).value

    @classmethod
    def resize_length(cls, length: torch.Tensor, ratio) -> torch.Tensor:
        return torch.ceil(length.float() * ratio).long()

    # ------------------------------------------------------------ HIP row predicate
    @property
    def lengths32(self) -> Optional[torch.Tensor]:
        """int32 [B] valid-frame counts (cached on the mask tensor); ``None``
        when the mask is known to be all-True."""
        m = self.mask
        if getattr(m, "_vg_full", False):
            return None
        cached = getattr(m, "_vg_len32", None)
        if cached is None:
            cached = m.sum(-1, dtype=torch.int32)
            m._vg_len32 = cached
        return cached

    @property
    def length(self) -> torch.Tensor:
        """int64 [B] valid-frame counts, cached on the mask tensor like ``lengths32`` (a step asks for them about
        ten times: two tiny launches each)."""
        m = self.mask
        cached = getattr(m, "_vg_len64", None)
        if cached is None:
            cached = m.long().sum(-1)
            try:
                m._vg_len64 = cached
            except AttributeError:
                pass
        return cached

    @property
    def device(self):
        return self.value.device

    def __len__(self) -> int:
        return len(self.value)

    def __repr__(self) -> str:
        return repr({"value": self.value, "mask": self.mask, "axis": self.axis})

    def size(self, i: Optional[int] = None):
        return self.value.size() if i is None else self.value.size(i)

    # ------------------------------------------------------------ masking / reshaping
    def apply_mask(self, mask_value: Number = 0) -> "TensorMask":
        assert self.axis == 1
        if getattr(self.mask, "_vg_full", False):         # no padding at all: nothing to mask (saves a fill + a select,
            return TensorMask(self.value, self.mask)      # forward and backward, per call)
        keep = _broadcastable(self.mask, self.value.dim())
        return TensorMask(torch.where(keep, self.value, mask_value), self.mask)

    def flatten(self) -> "TensorMask":
        assert self.axis == 1
        b, t = self.value.shape[:2]
        return TensorMask(self.value.reshape(b, t, -1), self.mask)

    def transpose(self, a: int = -1, b: int = -2) -> "TensorMask":
        return TensorMask(self.value.transpose(a, b), self.mask, axis=3 - self.axis)

    def squeeze(self, dim: Optional[int] = None) -> "TensorMask":
        v = self.value.squeeze() if dim is None else self.value.squeeze(dim)
        return TensorMask(v, self.mask)

    def expand(self) -> "TensorMask":
        return TensorMask(self.value.unsqueeze(-1), self.mask)

    def long(self) -> "TensorMask":
        return TensorMask(self.value.long(), self.mask)

    def abs(self) -> "TensorMask":
        return TensorMask(self.value.abs(), self.mask)

    def split(self, n: int) -> Tuple["TensorMask", "TensorMask"]:
        return (TensorMask(self.value[..., :n], self.mask),
                TensorMask(self.value[..., n:], self.mask))

    def cat(self, other: Union[torch.Tensor, "TensorMask"]) -> "TensorMask":
        o = other.value if isinstance(other, TensorMask) else other
        return TensorMask(torch.cat([self.value, o], 3 - self.axis), self.mask, axis=self.axis)

    def tolist(self, detach: bool = True) -> List[torch.Tensor]:
        assert self.axis == 1
        rows = [v[m] for v, m in zip(self.value, self.mask)]
        return [r.detach() for r in rows] if detach else rows

    # ------------------------------------------------------------ time-axis edits
    def _joined(self, tm, front: bool) -> "TensorMask":
        assert self.axis == 1
        if isinstance(tm, torch.Tensor):
            tm = TensorMask(tm)
        pair_v = [tm.value, self.value] if front else [self.value, tm.value]
        pair_m = [tm.mask, self.mask] if front else [self.mask, tm.mask]
        return TensorMask(torch.cat(pair_v, 1), torch.cat(pair_m, 1))

    def push(self, tm) -> "TensorMask":
        """Prepend frames (shift right)."""
        return self._joined(tm, front=True)

    def append(self, tm) -> "TensorMask":
        return self._joined(tm, front=False)

    def pop(self, n=1) -> "TensorMask":
        """Drop the last ``n`` frames; every sequence gets ``n`` shorter."""
        assert self.axis == 1
        return TensorMask.fromlength(self.value[:, :-n], self.length - n)

    def pop_left(self, n=1) -> "TensorMask":
        return TensorMask.fromlength(self.value[:, n:], self.length - n)

    # ------------------------------------------------------------ reductions / movement
    def mean(self) -> torch.Tensor:
        assert self.axis == 1
        v = self.flatten().apply_mask().value
        return (v / v.size(-1)).sum() / self.length.sum()

    def cuda(self) -> "TensorMask":
        return TensorMask(self.value.cuda(), self.mask.cuda())

    def to(self, device, non_blocking: bool = False) -> "TensorMask":
        return TensorMask(self.value.to(device, non_blocking=non_blocking),
                          self.mask.to(device, non_blocking=non_blocking))

    def detach(self) -> "TensorMask":
        return TensorMask(self.value.detach(), self.mask.detach())

    def batch_time_shuffle(self) -> "TensorMask":
        """Randomly permute the valid frames across batch and time."""
        assert self.axis == 1 and self.value.dim() == 3
        b, t, c = self.value.shape
        slots = torch.arange(b * t, device=self.device).reshape(b, t)[self.mask]
        slots = slots[torch.randperm(len(slots), device=self.device)]
        out = torch.zeros(b * t, c, dtype=self.value.dtype, device=self.device)
        out[slots] = self.value[self.mask]
        return TensorMask(out.reshape(b, t, c), self.mask).apply_mask()

    # ------------------------------------------------------------ arithmetic
    def _binary(self, other, op) -> "TensorMask":
        o = other.value if isinstance(other, TensorMask) else other
        return TensorMask(op(self.value, o), self.mask, axis=self.axis)

    def __truediv__(self, other):
        return self._binary(other, torch.true_divide)

    def __mul__(self, other):
        return self._binary(other, torch.mul)

    def __add__(self, other):
        return self._binary(other, torch.add)

    def __sub__(self, other):
        return self._binary(other, torch.sub)
