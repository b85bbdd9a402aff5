"""cuclark_amd — MI355X-native k-mer query engine behind CuCLARK's CuClarkDB contract.

Product code only: HIP kernels + C ABI (csrc/, lib/libmi_clark.so, include/mi_clark.h), the cuCLARK-compatible
CLI (exe/cuCLARK), and thin ctypes mirrors used by tests and bench.py.  Nothing here imports oracle/.
"""
from . import _lib  # noqa: F401
from ._lib import MicError, MIC_RESULT_WORDS, MIC_FLAG_ROW_OVERFLOW, MIC_FLAG_DENSE_PATH, MIC_ROW_INVALID  # noqa: F401
from .db import MiClarkDB  # noqa: F401
from . import host  # noqa: F401
