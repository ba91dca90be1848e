"""Golden fixtures for the structural histograms: runs the REFERENCE's own calculate_rdf / calculate_cdf
(/root/reference/scripts/lammps_distr.py) in this container — `numba` replaced by an identity `jit` stub, nothing else —
on small float32 samples and stores inputs + outputs as numbers in ref_distr.npz."""
import importlib.util
import os
import sys
import types

import numpy as np

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neuralmelting_amd import distr, lattice  # noqa: E402


def load_reference():
    nb = types.ModuleType('numba')

    def jit(*a, **k):
        if a and callable(a[0]):
            return a[0]
        return lambda f: f
    nb.jit = nb.njit = jit
    sys.modules['numba'] = nb
    spec = importlib.util.spec_from_file_location('lammps_distr_ref', '/root/reference/scripts/lammps_distr.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    mod = load_reference()
    rng = np.random.default_rng(42)
    out = {}
    cases = []
    # (a) 256-atom displaced fcc crystals at two densities + a random "gas", as float32 like lammps_parse.py writes them
    for sz, boxes in ((4, (6.17, 6.42)),):
        frac = lattice.fcc_fractional(sz)
        for b in boxes:
            x = (frac * b + 0.08 * (rng.random(frac.shape) - 0.5)).astype(np.float32)
            cases.append((x, np.float32(b)))
    cases.append((rng.random((256, 3)).astype(np.float32) * np.float32(6.3), np.float32(6.3)))
    # (b) a tiny sample whose displacements sit exactly on bin edges (0, +-l/2, l/2 in radius)
    l = np.float32(6.0)
    x = np.array([[0, 0, 0], [3, 0, 0], [0, 3, 0], [1.5, 1.5, 0], [0.375, 0.75, 1.125], [5.625, 0, 3]], dtype=np.float32)
    cases.append((x, l))
    groups = {'n256': [c for c in cases if len(c[0]) == 256], 'n6': [c for c in cases if len(c[0]) == 6]}
    for tag, cs in groups.items():
        pos = np.array([c[0] for c in cs])
        box = np.array([c[1] for c in cs], dtype=np.float32)
        natoms = np.full(len(cs), pos.shape[1], dtype=np.uint16)
        for sb, cb in ((64, 16), (17, 5)):
            nrho, dni, r, dn, rv = distr.calculate_spatial(natoms, box, sb, cb)
            rdf, cdf = [], []
            for i in range(len(cs)):
                rd = np.zeros(sb, dtype=np.float32)
                cd = np.zeros((cb, cb, cb), dtype=np.float32)
                rdf.append(np.array(mod.calculate_rdf(natoms[i], box[i], mod_br(mod), pos[i], r, rd)))
                cdf.append(np.array(mod.calculate_cdf(natoms[i], box[i], mod_br(mod), pos[i], rv, cd)))
            key = '%s_sb%d_cb%d_' % (tag, sb, cb)
            out[key + 'pos'], out[key + 'box'], out[key + 'natoms'] = pos, box, natoms
            out[key + 'r'], out[key + 'rv'] = r, rv
            out[key + 'rdf'], out[key + 'cdf'] = np.array(rdf), np.array(cdf)
    np.savez_compressed(os.path.join(HERE, 'ref_distr.npz'), **out)
    print('wrote ref_distr.npz', os.path.getsize(os.path.join(HERE, 'ref_distr.npz')), 'bytes;', sorted(out)[:4])


def mod_br(mod):
    b = [-1, 0, 1]
    return np.array([[b[i], b[j], b[k]] for i in range(3) for j in range(3) for k in range(3)], dtype=np.int8)


if __name__ == '__main__':
    main()
