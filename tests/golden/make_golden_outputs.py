"""Golden fixture for the whole output path (writer half): the REFERENCE's own init_outputs,
init_headers, write_outputs and consolidate_outputs (/root/reference/scripts/lammps_remcmc.py:153-316; module imported as
tests/golden/make_golden.py does) run on a synthetic STATE for three recorded cycles of a 2x3 grid; the two consolidated files
they leave are stored byte for byte in ref_outputs.npz with the numbers that went in.
tests/test_driver.py::test_output_files_equal_the_references replays it through the driver's writer."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    mod = G.load_reference()
    npn, ntn, sz, ncyc = 2, 3, 2, 3
    natoms = 4 * sz ** 3
    G.set_globals(mod, npn=npn, ntn=ntn, sz=sz, name='gout', nsmpl=5, cutoff=2, mod_=16)
    ns = npn * ntn
    rng = np.random.default_rng(33)
    rows = rng.standard_normal((ncyc, ns, 17)) * 10.0 ** rng.integers(-3, 4, (ncyc, ns, 17))
    rows[:, :, 8:14] = rng.integers(0, 17, (ncyc, ns, 6))
    rows[:, :, 14:17] = np.float32(rng.random((ncyc, ns, 3)))
    box = 3.0 + rng.random((ncyc, ns))
    x = (rng.random((ncyc, ns, 3 * natoms)) * 1.1 - 0.05) * box[:, :, None]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            mod.__dict__['PREF'] = os.getcwd() + '/%s.%s.%s.lammps' % ('gout', 'lj', 'fcc')       # remcmc:900
            mod.__dict__['OUTPUT'] = mod.init_outputs()                                            # remcmc:168-173
            mod.init_headers()                                                                     # remcmc:213-232
            for c in range(ncyc):
                state = []
                for k in range(ns):
                    r = rows[c, k]
                    state.append([natoms, x[c, k], np.zeros(3 * natoms), r[0], r[1], r[2], r[3], box[c, k], r[4], r[5], r[6], r[7],
                                  r[8], r[9], r[10], r[11], r[12], r[13], np.float32(r[14]), np.float32(r[15]), np.float32(r[16])])
                mod.__dict__['STATE'] = state
                mod.write_outputs()                                                                # remcmc:265-286
            mod.consolidate_outputs()                                                              # remcmc:289-316
            left = sorted(os.listdir(d))
            thrm = np.frombuffer(open(mod.PREF + '.thrm', 'rb').read(), dtype=np.uint8)
            traj = np.frombuffer(open(mod.PREF + '.traj', 'rb').read(), dtype=np.uint8)
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, 'ref_outputs.npz'), thrm=thrm, traj=traj, rows=rows, box=box, x=x,
                        shape=np.array([npn, ntn, sz, ncyc]))
    print(left, thrm.size, traj.size)


if __name__ == '__main__':
    main()
