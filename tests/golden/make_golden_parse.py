"""Golden fixture for the .thrm / .traj reader: runs the REFERENCE's own scripts/lammps_parse.py, unmodified, as a child
process in a scratch directory of this container (it needs numpy only) on a small synthetic run whose text was produced with
the reference's formatting expressions (remcmc:245, 254-256), and stores the input text plus every array the script wrote
(20 .npy files) in ref_parse.npz.  tests/test_parse.py::test_golden_from_the_reference_script replays it."""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = '/root/reference/scripts/lammps_parse.py'
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ('temp', 'pe', 'ke', 'virial', 'vol', 'dx', 'dv', 'dt', 'ntp', 'nap', 'ntv', 'nav', 'nth', 'nah', 'ap', 'av', 'ah',
         'natoms', 'box', 'pos')


def main():
    rng = np.random.default_rng(11)
    pn, tn, sn, natoms = 2, 3, 4, 32
    thrm, traj = [], []
    for k in range(pn * tn):
        thrm.append('# ---------------------\n# simulation parameters\n# ---------------------\n# nsmpl: %d\n' % sn
                    + '# | temp | pe | ke | virial | vol | dx | dv | dt | ntp | nap | ntv | nav | nth | nah | ap | av | ah |\n')
        for s in range(sn):
            row = rng.standard_normal(17) * 10.0 ** rng.integers(-4, 5, 17)
            row[8:14] = rng.integers(0, 129, 6)
            row[14:17] = np.float32(rng.random(3))
            if k == 1 and s == 2:
                row[3] = np.nan                                             # '%.4E' % nan -> 'NAN': loadtxt reads it back as nan
            thrm.append(17 * ' %.4E' % tuple(row) + '\n')                    # remcmc:245
            box = 5.5 + rng.random()
            x = (rng.random(3 * natoms) * 1.2 - 0.1) * box
            traj.append('%d %.4E\n' % (natoms, box))                         # remcmc:254
            for i in range(natoms):
                traj.append(3 * ' %.4E' % tuple(x[3 * i:3 * i + 3]) + '\n')   # remcmc:256
    thrm_txt, traj_txt = ''.join(thrm), ''.join(traj)
    out = {'thrm_txt': np.frombuffer(thrm_txt.encode(), dtype=np.uint8), 'traj_txt': np.frombuffer(traj_txt.encode(), dtype=np.uint8),
           'P': np.linspace(1, 8, pn, dtype=np.float32), 'T': np.linspace(0.25, 2.5, tn, dtype=np.float32)}
    with tempfile.TemporaryDirectory() as d:
        pre = os.path.join(d, 'g.lj.fcc.lammps')
        np.save(pre + '.virial.trgt.npy', out['P'])
        np.save(pre + '.temp.trgt.npy', out['T'])
        open(pre + '.thrm', 'w').write(thrm_txt)
        open(pre + '.traj', 'w').write(traj_txt)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
        subprocess.run([sys.executable, REF, '-n', 'g', '-e', 'LJ'], cwd=d, check=True, env=env)
        for n in NAMES:
            out['ref_' + n] = np.load(pre + '.%s.npy' % n)
    np.savez_compressed(os.path.join(HERE, 'ref_parse.npz'), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith('ref_')})


if __name__ == '__main__':
    main()
