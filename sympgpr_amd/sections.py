"""Independent per-section fits across GPUs -- "replicas only".

python/05_tokamak/Split_SympGPR/main.py:96-112 trains nphmap unrelated GP pairs, one per toroidal
section (`for i in range(0, nphmap): ... regGP(q[:, i], ...)`, `GP(xtrain[:, i], ...)`).  The fits
share nothing, so section m simply runs on rank m % world: no collective on the data path; one
all_gather of the (small) weight vectors afterwards so every rank can apply the full map.
"""
import numpy as np


def owned_sections(nphmap, rank, world):
    return list(range(rank, nphmap, world))


def _hip_fit(family, x, y, z, hyp, sig2n, reg):
    from .fit import SympFit
    with SympFit(family, x, y, z, hyp, sig2n, reg=reg) as f:
        f.run()
        return f.alpha(), f.nll()


def fit_sections(family, xtrain, ztrain, hyp, sig2n, reg=False, rank=0, world=1, fit_fn=None):
    """xtrain (2N x nphmap) columns (q || P), ztrain (n x nphmap) with n = 2N (N with reg=True),
    hyp (nphmap x 3) rows (lx, ly, sig).  Returns {m: (alpha_m, nll_m)} for the sections this rank
    owns.  fit_fn(family, x, y, z, hyp, sig2n, reg) -> (alpha, nll) defaults to the device fit."""
    xtrain, ztrain, hyp = np.asarray(xtrain), np.asarray(ztrain), np.atleast_2d(hyp)
    nphmap = xtrain.shape[1]
    N = xtrain.shape[0] // 2
    out = {}
    mine = owned_sections(nphmap, rank, world)
    if fit_fn is None and mine:
        # the sections of this rank in ONE launch (one workgroup per section) while the order allows it:
        # at the drivers' sizes (order 40 ... 160) a fit through a handle is all set-up time
        from .fit import batch_max_order, fit_batch
        if (N if reg else 2 * N) <= batch_max_order():
            sel = np.array(mine)
            al, nll, info = fit_batch(family, xtrain[:N, sel].T, xtrain[N:2 * N, sel].T, ztrain[:, sel].T, hyp[sel],
                                      sig2n, reg=reg)
            if np.any(info):
                raise np.linalg.LinAlgError("section %d: %d-th leading minor of the array is not positive definite"
                                            % (int(sel[np.nonzero(info)[0][0]]), int(info[np.nonzero(info)[0][0]])))
            return {int(m): (al[k].copy(), float(nll[k])) for k, m in enumerate(sel)}
    fit_fn = fit_fn or _hip_fit
    for m in mine:
        out[m] = fit_fn(family, xtrain[:N, m], xtrain[N:2 * N, m], ztrain[:, m], hyp[m], sig2n, reg)
    return out


def gather_sections(local, nphmap, group=None):
    """Every rank -> ([alpha_0 .. alpha_{nphmap-1}], [nll_0 ..]).  Collective (object all_gather:
    the payload is nphmap weight vectors, not matrices)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, local, group=group)
    merged = {}
    for p in parts:
        merged.update(p)
    missing = [m for m in range(nphmap) if m not in merged]
    if missing:
        raise RuntimeError("sections %s were fitted by no rank" % missing)
    return [merged[m][0] for m in range(nphmap)], [merged[m][1] for m in range(nphmap)]
