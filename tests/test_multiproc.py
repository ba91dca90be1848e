"""N>1 path on CPU: two ranks (gloo) each own whole pressure rows; there is no data-path collective, only barriers
around the shared files.  The engine is the oracle stand-in (TEST ONLY) so this runs without a GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import OracleEngine
from neuralmelting_amd import remcmc

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(nproc, cwd, argv):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.join(HERE, '_mp_driver.py'), str(cwd)] + argv
    env = dict(os.environ, OMP_NUM_THREADS='2')
    subprocess.run(cmd, check=True, timeout=600, env=env, cwd=str(cwd))


@pytest.mark.parametrize('npn,world', [(2, 2), (3, 2), (1, 2)])  # (1, 2): one pressure row split across two ranks
def test_two_ranks_equal_one_rank(tmp_path, oracle, npn, world):
    # the split-row case uses a narrow temperature range so that the sweep does swap replicas across the two ranks
    argv = ('-bm -n mp -e LJ -ss 4 -pn %d -tn %d -sn 4 -sm 4 -rd 1' % (npn, 4 if npn == 1 else 2)).split()
    if npn == 1:
        argv += ['-tr', '1.0', '1.06']
    one = tmp_path / 'one'; two = tmp_path / 'two'
    one.mkdir(); two.mkdir()
    run = remcmc.Run(argv, cwd=str(one))
    run.make_engine = lambda: OracleEngine(oracle, run)
    run.main()
    launch(world, two, argv)
    for ext in ('.thrm', '.traj'):
        a = open(str(one / ('mp.lj.fcc.lammps' + ext))).read()
        b = open(str(two / ('mp.lj.fcc.lammps' + ext))).read()
        assert a == b                                     # byte-identical consolidated outputs
    ra = np.load(str(one / 'mp.lj.fcc.lammps.rstrt.0003.npy'), allow_pickle=True)
    rb = np.load(str(two / 'mp.lj.fcc.lammps.rstrt.0003.npy'), allow_pickle=True)
    assert ra.shape == rb.shape == (npn * (4 if npn == 1 else 2), 21)
    for sa, sb in zip(ra, rb):
        np.testing.assert_array_equal(sa[1], sb[1])
        assert [float(q) for q in sa[3:]] == [float(q) for q in sb[3:]]
    assert not [f for f in os.listdir(two) if '.part' in f]
