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


def launch(nproc, cwd, argv, script=None, extra_env=None):
    """torch.distributed.run; by default the helper that swaps the oracle stand-in in, else `script` (a module path)"""
    target = [os.path.join(HERE, '_mp_driver.py'), str(cwd)] if script is None else ['-m', script]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port())] + target + argv
    env = dict(os.environ, OMP_NUM_THREADS='2', **(extra_env or {}))
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


@pytest.mark.gpu
@pytest.mark.parametrize('npn,world', [(2, 2), (1, 2)])  # whole rows per rank; one pressure row split across the two ranks
def test_two_ranks_with_the_hip_engine_equal_one_rank(tmp_path, npn, world):
    """The N>1 path with the REAL engine: two ranks (one process each, launched by torch.distributed.run before anything touches
    the GPU) share the one card of this box, so the process group is gloo (RCCL refuses two ranks on one device); everything else
    is the production path — each rank's own context and stream, the device-side exchange for whole rows, the all-gather +
    identical-sweep + re-seat exchange when a row spans ranks — and must leave byte-identical files to a single-rank run."""
    tn = 4 if npn == 1 else 2
    argv = ('-bm -n mpg -e LJ -ss 4 -pn %d -tn %d -sn 4 -sm 8 -rd 1' % (npn, tn)).split()
    if npn == 1:
        argv += ['-tr', '1.0', '1.06']      # a narrow temperature range: the sweep does swap replicas across the two ranks
    one = tmp_path / 'one'; two = tmp_path / 'two'
    one.mkdir(); two.mkdir()
    remcmc.Run(argv, cwd=str(one)).main()
    launch(world, two, argv, script='neuralmelting_amd.remcmc', extra_env={'NM_DIST_BACKEND': 'gloo', 'PYTHONPATH': os.path.dirname(HERE)})
    for ext in ('.thrm', '.traj'):
        assert open(str(one / ('mpg.lj.fcc.lammps' + ext))).read() == open(str(two / ('mpg.lj.fcc.lammps' + ext))).read()
    ra = np.load(str(one / 'mpg.lj.fcc.lammps.rstrt.0003.npy'), allow_pickle=True)
    rb = np.load(str(two / 'mpg.lj.fcc.lammps.rstrt.0003.npy'), allow_pickle=True)
    assert ra.shape == rb.shape == (npn * tn, 21)
    for sa, sb in zip(ra, rb):
        np.testing.assert_array_equal(sa[1], sb[1])
        np.testing.assert_array_equal(sa[2], sb[2])
        assert [float(q) for q in sa[3:]] == [float(q) for q in sb[3:]]
    if npn == 1:                                # the split-row sweep really moved configurations between the ranks
        th = np.loadtxt(str(two / 'mpg.lj.fcc.lammps.thrm'))
        assert len(th) == npn * tn * 4


def test_a_failing_rank_ends_the_whole_job(tmp_path, oracle):
    """An error on ONE rank (injected on rank 1 before it reaches its first barrier) must not leave rank 0 waiting in that
    barrier for ever: the failing rank leaves with a failure code and the launcher ends the job (CPU, gloo, oracle stand-in)."""
    argv = '-bm -n mpf -e LJ -ss 4 -pn 2 -tn 2 -sn 2 -sm 4'.split()
    with pytest.raises(subprocess.CalledProcessError):
        launch(2, tmp_path, argv, extra_env={'NM_TEST_FAIL_RANK': '1'})
    launch(2, tmp_path, argv)            # the same job without the injection runs through


def _bench_line(tmp_path, nproc, argv):
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(HERE, '_mp_bench.py'), '--gpus', str(nproc)] + argv
    r = subprocess.run(cmd, timeout=600, cwd=str(tmp_path), capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS='2'))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])


def test_bench_line_at_two_ranks_carries_the_metrics_own_grid_and_the_weak_leg(tmp_path, oracle):
    """bench.py --gpus 2 as the driver launches it (here: gloo, CPU, the oracle stand-in as the engine; a tiny custom grid).  `value` must be
    the STRONG-scaling rate of the preset's own grid — the grid BASELINE's metric names, dealt out over the ranks — and the weak-scaling rate
    of the grid with twice the pressure rows must ride along under `weak`; each with its replicas_total and the world size the backend saw,
    and each consistent with its own ms_per_step."""
    d = _bench_line(tmp_path, 2, '--sz 4 --rows 2 --tn 2 --mod 2 --steps 2 --warmup 1 --equil 0 --no-cpu'.split())
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong'
    c = d['config']
    assert c['replicas_total'] == 4 and c['replicas_per_gpu'] == 2 and c['world_size_seen_by_backend'] == 2 and c['backend'] == 'gloo'
    assert '2x2 PxT grid over 2 GPUs' in d['metric']
    assert abs(d['value'] - c['sweeps_per_step'] / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    w = d['weak']
    assert w['scaling'] == 'weak' and w['replicas_total'] == 8 and w['replicas_per_gpu'] == 4 and w['world_size_seen_by_backend'] == 2
    assert '4x2 PxT grid over 2 GPUs' in w['metric']
    assert abs(w['value'] - w['sweeps_per_step'] / (w['ms_per_step'] * 1e-3)) < 1e-6 * w['value']
    # one leg only when asked for
    d1 = _bench_line(tmp_path, 2, '--sz 4 --rows 2 --tn 2 --mod 2 --steps 2 --warmup 1 --equil 0 --no-cpu --scaling weak'.split())
    assert d1['scaling'] == 'weak' and 'weak' not in d1 and d1['config']['replicas_total'] == 8


def test_bench_line_at_one_rank_is_the_preset_as_it_stands(tmp_path, oracle):
    d = _bench_line(tmp_path, 1, '--sz 4 --rows 2 --tn 2 --mod 2 --steps 2 --warmup 1 --equil 0 --no-cpu'.split())
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and 'weak' not in d and d['config']['replicas_total'] == 4


@pytest.mark.gpu
def test_split_row_exchange_on_rccl_with_one_rank(tmp_path):
    """The collective (split-row) exchange on the nccl backend = RCCL.  A one-GPU box cannot hold two RCCL ranks, so the path is
    forced (NM_FORCE_SPLIT_ROWS=1) in a world of one launched the way the driver launches ranks: the (E_tot, V) all-gather then
    runs on CUDA tensors over RCCL, the host-side sweep decides, the swapped replicas are re-seated one by one — and the files
    must equal, byte for byte, those of the same run with the device-side exchange."""
    argv = '-bm -n sp -e LJ -ss 4 -pn 2 -tn 4 -sn 5 -sm 8 -rd 1 -tr 1.0 1.06'.split()   # narrow range: the sweep does swap
    one = tmp_path / 'one'; two = tmp_path / 'two'
    one.mkdir(); two.mkdir()
    remcmc.Run(argv, cwd=str(one)).main()
    launch(1, two, argv, script='neuralmelting_amd.remcmc',
           extra_env={'NM_FORCE_SPLIT_ROWS': '1', 'NM_DIST_BACKEND': 'nccl', 'PYTHONPATH': os.path.dirname(HERE), 'NM_LOG_EXCHANGE': str(two / 'xlog')})
    for ext in ('.thrm', '.traj'):
        assert open(str(one / ('sp.lj.fcc.lammps' + ext))).read() == open(str(two / ('sp.lj.fcc.lammps' + ext))).read()
    ra = np.load(str(one / 'sp.lj.fcc.lammps.rstrt.0004.npy'), allow_pickle=True)
    rb = np.load(str(two / 'sp.lj.fcc.lammps.rstrt.0004.npy'), allow_pickle=True)
    for sa, sb in zip(ra, rb):
        np.testing.assert_array_equal(sa[1], sb[1])
        np.testing.assert_array_equal(sa[2], sb[2])
        assert [float(q) for q in sa[3:]] == [float(q) for q in sb[3:]]
    log = open(str(two / 'xlog')).read().split()
    assert log[0] == 'nccl' and log[1] == 'cuda' and int(log[2]) > 0     # backend, all-gather on CUDA tensors, replicas re-seated


@pytest.mark.gpu
def test_bench_under_torchrun_uses_rccl(tmp_path):
    """bench.py launched the way the driver launches it (torch.distributed.run, one rank per GPU, backend nccl = RCCL): on this
    one-GPU box that is a world of one, which still exercises the RCCL process group, its barrier and the max/sum all-reduce"""
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(os.path.dirname(HERE), 'bench.py'), '--gpus', '1', '--steps', '2',
           '--warmup', '1', '--no-cpu', '--scaling', 'strong']
    r = subprocess.run(cmd, timeout=600, cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 1 and d['config']['backend'] == 'nccl' and d['config']['world_size_seen_by_backend'] == 1
    assert d['scaling'] == 'strong' and d['config']['replicas_total'] == 64 and d['value'] > 1e5
