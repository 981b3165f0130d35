"""MI355X-native batched mean-field-game environment + actor-critic hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).
Hot path: hand-written HIP kernels behind the C ABI of include/mfg_hip.h (csrc/).
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
__version__ = '0.1.0'
