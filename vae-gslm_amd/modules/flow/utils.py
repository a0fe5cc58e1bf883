from typing import NamedTuple, Union

import torch

from utils.tensormask import TensorMask


class TensorLogdet(NamedTuple):
    tensor: Union[TensorMask, torch.Tensor]
    logdet: Union[float, torch.Tensor]
