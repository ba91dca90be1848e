"""Golden fixture for the restart interchange: the REFERENCE's own dump_samples_restart
(/root/reference/scripts/lammps_remcmc.py:821-828; module imported as tests/golden/make_golden.py does) writes a state list of
the shape its gen_mc_param returns (remcmc:743-745) and the file's bytes are stored in ref_restart.npz together with the numbers
that went in.  tests/test_driver.py::test_restart_file_written_by_the_reference_loads replays it."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    mod = G.load_reference()
    npn, ntn, natoms = 2, 2, 32
    G.set_globals(mod, npn=npn, ntn=ntn, name='gold')
    rng = np.random.default_rng(21)
    ns = npn * ntn
    x = rng.random((ns, 3 * natoms)) * 6.0
    v = rng.standard_normal((ns, 3 * natoms))
    sc = rng.random((ns, 9)) + 0.5                       # temp pe ke virial box vol dx dv dt
    state = []
    for k in range(ns):
        s = [natoms, x[k].copy(), v[k].copy(), sc[k, 0], -sc[k, 1] * 100, sc[k, 2], sc[k, 3], 6.0 + sc[k, 4], (6.0 + sc[k, 4]) ** 3,
             sc[k, 6] * 0.03, sc[k, 7] * 0.03, sc[k, 8] * 0.004] + [0.0] * 9
        state.append(mod.gen_mc_param(s[:18] + [np.float32(0.4), np.float32(0.6), np.float32(0.5)]))  # remcmc:726-745
    mod.__dict__['STATE'] = state
    mod.__dict__['STEP'] = 6
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            mod.dump_samples_restart()                                                        # remcmc:821-828
        finally:
            os.chdir(cwd)
        fn = os.path.join(d, 'gold.lj.fcc.lammps.rstrt.0007.npy')
        raw = np.frombuffer(open(fn, 'rb').read(), dtype=np.uint8)
    out = dict(rstrt=raw, x=x, v=v, box=np.array([s[7] for s in state]), d=np.array([s[9:12] for s in state], dtype=np.float64),
               th=np.array([[s[3], s[4], s[5], s[6], s[8]] for s in state], dtype=np.float64))
    np.savez_compressed(os.path.join(HERE, 'ref_restart.npz'), **out)
    print(raw.size, 'bytes', out['d'])


if __name__ == '__main__':
    main()
