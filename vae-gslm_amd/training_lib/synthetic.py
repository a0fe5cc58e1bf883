"""Synthetic batches with the shapes of the reference's collate output
(data/dataset.py:385-429 -> utils/helpers.py:80-135): 50 Hz HuBERT-k-means
token ids, 80-bin mel frames (already rescaled to ~N(0,1), configs/...:157-159)
and a 2-4 s utterance crop.  Used by the benchmark, the smoke test and
``scripts/train.py --synthetic`` (no dataset ships with either repository).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from utils.tensormask import TensorMask


def make_batch(batch_size: int, seq_len: int, device, seed: int, vocab: int = 200, n_mels: int = 80,
               utt_len: int = 150, lengths: Optional[Sequence[int]] = None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    tokens = torch.randint(0, vocab, (batch_size, seq_len), generator=g)
    mel = torch.randn(batch_size, seq_len, n_mels, generator=g)
    utt = torch.randn(batch_size, utt_len, n_mels, generator=g)
    if lengths is None:
        mask = None
    else:
        mask = torch.arange(seq_len)[None] < torch.as_tensor(lengths)[:, None]
    dev = torch.device(device)          # "cpu": a host batch for training_lib.prefetch.DevicePrefetcher
    mk = None if mask is None else mask.to(dev)
    return {"tokens": TensorMask(tokens.to(dev), mk),
            "mel": TensorMask(mel.to(dev), mk),
            "cropped_mel_utt": TensorMask(utt.to(dev))}
