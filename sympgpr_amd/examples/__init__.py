"""Per-driver call surfaces.

The reference has no single library: every example directory carries its own copy of func.py
whose functions differ in small ways (an extra positional argument, a third return value, a
different wrap, a fallback branch).  One module per driver directory reproduces the names and
signatures that driver imports, on top of the same device path:

    module                      reference file                                  kernel family
    --------------------------  ----------------------------------------------  -------------
    pendulum_implicit           python/01_pendulum/implicit/func.py             A
    pendulum_period_unknown     python/01_pendulum/implicit_period_unknown/func.py   D
    pendulum_explicit           python/01_pendulum/explicit/func_expl.py        B
    pert_pendulum               python/02_pert_pendulum/func.py                 A
    henon_heiles                python/03_henon_heiles/func.py                  C
    standard_map                python/04_standard_map/func.py                  A (implicit), B (explicit)
    tokamak                     python/05_tokamak/SympGPR/func.py               A
    tokamak_split               python/05_tokamak/Split_SympGPR/func.py         A

Physics and data generators of those files (intode, integrate_pendulum, energy, dydt_ivp,
symplEuler_pendulum, fieldlines) are outside the accelerated path and are not mirrored; the
tokamak maps take the flux-surface test `compute_r` as an optional callable.
"""
