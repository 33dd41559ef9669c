"""ttcross_amd -- MI355X-native engine for the dtt_dmrgg greedy-cross sweep of aukeschaap/ttcross.

Python is plumbing only: `engine` binds the C-ABI of libttx.so (include/ttx.h) with ctypes, `drivers`
mirrors the command-line drivers test_crs_ising / test_crs_mvn / test_crs_stdnorm.  All computation happens
in hand-written HIP kernels (ttcross_amd/csrc); there is no CPU fallback.
"""
from .engine import (TTX_FUN_ISING, TTX_FUN_MVN, TTX_FUN_STDNORM, TTCross, TTXError, dtt_dmrgg, lib_path, load_library)  # noqa: F401
