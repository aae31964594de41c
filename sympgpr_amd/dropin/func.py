"""`from func import ...` for unmodified reference drivers: put this directory on sys.path."""
from sympgpr_amd.func import *  # noqa: F401,F403
from sympgpr_amd.func import (applymap, applymap_henon, build_K, buildKreg, calcP, calcQ, gpsolve, guessP,  # noqa: F401
                              nll_chol, nll_chol_reg, quality, solve_cholesky)
