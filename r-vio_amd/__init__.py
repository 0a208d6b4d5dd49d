"""rvio_amd — MI355X-native robocentric MSCKF hot path (HIP kernels behind a C-ABI).

Layout:
  csrc/    hand-written HIP kernels (gfx950) + the C-ABI of include/rvio_hip.h
  abi.py   ctypes mirror of the C-ABI structs
  hip.py   loader/wrapper of librvio_hip.so (fails loudly if it is missing)
  synth.py deterministic synthetic EuRoC-shaped inputs (tests + bench)
  build.py in-tree hipcc build recipe
"""
from . import abi  # noqa: F401
from . import synth  # noqa: F401
