"""Stand-in for the reference's python/fortran package, so that
`from fortran.sympgpr import sympgpr` (python/functions/func.py:13) resolves to the HIP path."""
