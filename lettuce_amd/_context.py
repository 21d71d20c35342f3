"""Device / dtype / engine policy and tensor factories.

Drop-in for lettuce/_context.py:9-107.  ``use_native=True`` selects the HIP engine
(ROCm torch reports MI355X as a ``cuda`` device, so the reference's flag keeps its
meaning); there is no silent downgrade: a native context whose engine library is
missing fails when the first plan is built.
"""
from typing import List, Optional, Union

import numpy as np
import torch

__all__ = ["Context"]

_FLOATS = (torch.float16, torch.float32, torch.float64)


class Context:
    def __init__(self, device: Optional[Union[torch.device, str]] = None,
                 dtype: Optional[torch.dtype] = None, use_native: Optional[bool] = None):
        gpu = torch.cuda.is_available()
        if device is None:
            if use_native is None:
                use_native = gpu
                device = "cuda:0" if gpu else "cpu"
            else:
                assert not use_native or gpu, \
                    "cuda_native extension explicitly requested but cuda is not available!"
                device = "cuda:0"
        else:
            on_gpu = "cuda" in str(device)
            if on_gpu:
                assert gpu, "cuda device explicitly requested but cuda is not available!"
            else:
                assert "cpu" in str(device), \
                    f"lettuce is designed to work on cpu or cuda devices. {device} is not supported!"
            if use_native is None:
                use_native = on_gpu
            else:
                assert on_gpu or not use_native, \
                    "can not use explicitly requested cuda_native extension on explicitly requested cpu device!"
        dtype = dtype or torch.float32
        assert dtype in _FLOATS, \
            f"lettuce is designed to work with common float types (16, 32 and 64 bit). {dtype} is not supported!"
        self.device = torch.device(device)
        self.dtype = dtype
        self.use_native = bool(use_native)

    # -- factories (lettuce/_context.py:64-77) --------------------------------------------
    def _make(self, factory, size, args, dtype, kwargs):
        return factory(size, *args, **kwargs, device=self.device, dtype=(dtype or self.dtype))

    def empty_tensor(self, size: Union[List[int], torch.Size], *args, dtype=None, **kwargs):
        return self._make(torch.empty, size, args, dtype, kwargs)

    def zero_tensor(self, size: Union[List[int], torch.Size], *args, dtype=None, **kwargs):
        return self._make(torch.zeros, size, args, dtype, kwargs)

    def one_tensor(self, size: Union[List[int], torch.Size], *args, dtype=None, **kwargs):
        return self._make(torch.ones, size, args, dtype, kwargs)

    def convert_to_tensor(self, array, *args, dtype: Optional[torch.dtype] = None, **kwargs):
        """bool stays bool, uint8 stays uint8, everything else takes the context dtype
        (lettuce/_context.py:79-99)."""
        target = dtype
        if target is None:
            src = getattr(array, "dtype", None)
            if src in (bool, torch.bool):
                target = torch.bool
            elif src in (torch.uint8, np.uint8):
                target = torch.uint8
            else:
                target = self.dtype
        if isinstance(array, torch.Tensor):
            return array.to(*args, **kwargs, device=self.device, dtype=target)
        return torch.tensor(array, *args, **kwargs, device=self.device, dtype=target)

    @staticmethod
    def convert_to_ndarray(tensor: Union[torch.Tensor, List]) -> np.ndarray:
        if isinstance(tensor, torch.Tensor):
            return tensor.detach().cpu().numpy()
        return np.array(tensor)
