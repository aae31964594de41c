"""`from kernels import *` for unmodified reference code."""
from sympgpr_amd.kernels import *  # noqa: F401,F403
