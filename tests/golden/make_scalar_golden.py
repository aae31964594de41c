#!/usr/bin/env python3
"""tests/golden/scalars.json: every generated scalar function of the four kernels*.f90 families
(19 each) evaluated by the REFERENCE's own compiled Fortran (oracle/_ref, build container only) at
a few random arguments.  Data only: inputs and expected outputs.
Usage:  python tests/golden/make_scalar_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import DL_NAMES, X_NAMES, Ref  # noqa: E402

BASE = ("kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num")


def main():
    ref = Ref()
    rng = np.random.default_rng(20261004)
    out = {"meta": {"source": "oracle/_ref (amdflang build of the reference's kernels*.f90)",
                    "arguments": "name(x_a, y_a, x_b, y_b, lx, ly[, p])"}}
    m = 12
    for fam in "ABCD":
        args = {"x_a": rng.uniform(-3, 3, m), "y_a": rng.uniform(-3, 3, m), "x_b": rng.uniform(-3, 3, m),
                "y_b": rng.uniform(-3, 3, m), "lx": rng.uniform(0.3, 2, m), "ly": rng.uniform(0.3, 2, m),
                "p": rng.uniform(0.3, 1.2, m)}
        vals = {}
        for name in BASE + tuple(DL_NAMES.values()) + tuple(X_NAMES.values()):
            vals[name] = [ref.scalar(fam, name, args["x_a"][i], args["y_a"][i], args["x_b"][i], args["y_b"][i],
                                     args["lx"][i], args["ly"][i], args["p"][i] if fam == "D" else None)
                          for i in range(m)]
        out[fam] = {"args": {k: v.tolist() for k, v in args.items()}, "values": vals}
    with open(os.path.join(HERE, "scalars.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote scalars.json:", {fam: len(out[fam]["values"]) for fam in "ABCD"})


if __name__ == "__main__":
    main()
