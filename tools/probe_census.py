"""Where do the workgroups of a small kernel B go while a persistent grid A holds most of the chip?
python tools/probe_census.py   (prints, per scenario, how many B workgroups started while A was still running)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
probe = L.load_probe_library()

def run(na, nb, lds_a=106504, lds_b=147464, spin_a=3000, spin_b=100, mask_a=None, mask_b=None, ta=512, tb=256, label=""):
    out = (C.c_ulonglong * (4 * (na + nb)))()
    ma = (C.c_uint * len(mask_a))(*mask_a) if mask_a else None
    mb = (C.c_uint * len(mask_b))(*mask_b) if mask_b else None
    L.check(probe.sgpr_probe_census(na, ta, lds_a, spin_a, ma, len(mask_a) if mask_a else 0,
                                    nb, tb, lds_b, spin_b, mb, len(mask_b) if mask_b else 0, out))
    a = np.array(out[:], dtype=np.uint64).reshape(-1, 4)
    A, B = a[:na], a[na:]
    t0 = A[:, 2].min()
    a_end = A[:, 3].min()
    def loc(x):
        hw = x[:, 1].astype(np.int64)
        return list(zip(x[:, 0].astype(int).tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 8) & 15).tolist()))
    a_cus = set(loc(A)); b_loc = loc(B)
    a_conc = int(((A[:, 2] - t0) < 5000).sum())          # started within 50 us of the first
    b_early = int((B[:, 2] < a_end).sum())
    per_xcc_a = np.bincount(A[:, 0].astype(int), minlength=8)
    print("%s A=%d B=%d: A workgroups started together %d on %d distinct CUs (per XCC %s); B started before A's first exit: %d of %d; "
          "B start (us after A) min %.0f max %.0f; B on CUs shared with A: %d" % (
              label, na, nb, a_conc, len(a_cus), per_xcc_a.tolist(), b_early, nb,
              (B[:, 2].min() - t0) / 100.0, (B[:, 2].max() - t0) / 100.0, sum(1 for l in b_loc if l in a_cus)), flush=True)
    return A, B

if __name__ == "__main__":
    for na, nb in ((248, 8), (240, 8), (240, 12), (240, 16), (232, 16), (224, 16), (224, 32), (192, 32)):
        run(na, nb, label="unmasked")
    # the same with B's 147 KiB replaced by a size that fits beside an A workgroup
    run(256, 16, lds_b=40000, label="B fits beside A")
    # CU masks: the probe_cumask bit order is printed by tools/probe_cumask.py; try one CU per XCC for B
    for bits in (0x00000001, 0x80000000):
        mb = [bits] * 8
        ma = [0xFFFFFFFF & ~bits] * 8
        run(248, 8, mask_a=ma, mask_b=mb, label="masked %08x" % bits)
        run(248, 16, mask_a=ma, mask_b=mb, spin_b=50, label="masked %08x (B oversubscribed)" % bits)
