"""CPU ORACLE (test infrastructure) for the structural histograms: numpy restatement of calculate_rdf / calculate_cdf of
/root/reference/scripts/lammps_distr.py ("distr").  Pinned by tests/golden/ref_distr.npz, which holds what the reference's
own two functions return (tests/golden/make_golden_distr.py)."""
import numpy as np

# br, distr:99-102
_B = [-1, 0, 1]
BR = np.array([[_B[i], _B[j], _B[k]] for i in range(3) for j in range(3) for k in range(3)], dtype=np.int8)


def calculate_rdf(natoms, box, pos, r):
    """distr:123-135"""
    rd = np.zeros(len(r), dtype=np.float32)
    for j in range(BR.shape[0]):
        dvm = pos - (pos + box * BR[j].reshape(1, -1)).reshape(-1, 1, 3)
        d = np.sqrt(np.sum(np.square(dvm), -1))
        rd[1:] += np.histogram(d, r)[0]
    return rd / natoms


def calculate_cdf(natoms, box, pos, rv):
    """distr:161-171"""
    cd = np.zeros(tuple(np.array(rv.shape[1:]) - 1) * 3, dtype=np.float32)
    for j in range(BR.shape[0]):
        dvm = pos - (pos + box * BR[j].reshape(1, -1)).reshape(-1, 1, 3)
        cd += np.histogramdd(dvm.reshape(-1, 3), rv)[0]
    return cd / natoms
