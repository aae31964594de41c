"""`from fortran.sympgpr import sympgpr` (python/functions/func.py:13)."""
from sympgpr_amd.fortran.sympgpr import sympgpr  # noqa: F401
