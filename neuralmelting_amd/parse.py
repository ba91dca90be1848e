"""Reader of the sampler's text output with lammps_parse.py's command line and outputs.

Mirrors /root/reference/scripts/lammps_parse.py ("parse"): loads <PREFIX>.virial.trgt.npy / .temp.trgt.npy for the grid
shape, turns the consolidated <PREFIX>.thrm into the 17 float32 arrays of shape (PN, TN, SN) (parse:48-85) and the
consolidated <PREFIX>.traj into .natoms (uint16, (PN, TN, SN)), .box (float32, flat, parse:92 does not reshape it) and .pos
(float32, (PN, TN, SN, natoms, 3)) (parse:88-104).  The text is read by the multi-threaded reader of include/nm_parse.h
instead of np.loadtxt and a Python list of every line.

    python -m neuralmelting_amd.parse -v -n remcmc_init -e LJ
"""
import argparse
import ctypes as C
import os

import numpy as np

from . import _lib as B

LAT = {'Ti': 'bcc', 'Al': 'fcc', 'Ni': 'fcc', 'Cu': 'fcc', 'LJ': 'fcc'}
# column order of the .thrm rows (remcmc:208) = suffixes of the files written (parse:69-85)
COLUMNS = ('temp', 'pe', 'ke', 'virial', 'vol', 'dx', 'dv', 'dt', 'ntp', 'nap', 'ntv', 'nav', 'nth', 'nah', 'ap', 'av', 'ah')


def parse_args(argv=None):
    """lammps_parse.py's flags (parse:14-20)"""
    p = argparse.ArgumentParser()
    p.add_argument('-v', '--verbose', help='verbose output', action='store_true')
    p.add_argument('-n', '--name', help='name of simulation', type=str, default='remcmc_init')
    p.add_argument('-e', '--element', help='element choice', type=str, default='LJ')
    return p.parse_args(argv)


def _fail(what):
    raise RuntimeError('%s: %s' % (what, B.load().nm_parse_last_error().decode()))


def read_thrm(path, nthreads=0):
    """np.loadtxt(path, dtype=np.float32) for a .thrm file: (rows, 17) float32"""
    L = B.load()
    n = C.c_long(0)
    if L.nm_parse_thrm(os.fsencode(path), None, 0, C.byref(n), nthreads) != 0:
        _fail('nm_parse_thrm')
    rows = np.empty((n.value, 17), np.float32)
    if L.nm_parse_thrm(os.fsencode(path), rows.ctypes.data_as(B.c_float_p), n.value, C.byref(n), nthreads) != 0:
        _fail('nm_parse_thrm')
    return rows


def read_traj(path, nthreads=0):
    """the three flat arrays of parse:88-94: natoms (frames,) uint16, box (frames,) float32, x (coordinate lines, 3) float32"""
    L = B.load()
    nf, nr = C.c_long(0), C.c_long(0)
    if L.nm_parse_traj(os.fsencode(path), None, None, None, 0, 0, C.byref(nf), C.byref(nr), nthreads) != 0:
        _fail('nm_parse_traj')
    natoms = np.empty(nf.value, np.uint16)
    box = np.empty(nf.value, np.float32)
    x = np.empty((nr.value, 3), np.float32)
    if L.nm_parse_traj(os.fsencode(path), natoms.ctypes.data_as(C.POINTER(C.c_uint16)), box.ctypes.data_as(B.c_float_p),
                       x.ctypes.data_as(B.c_float_p), nf.value, nr.value, C.byref(nf), C.byref(nr), nthreads) != 0:
        _fail('nm_parse_traj')
    return natoms, box, x


def main(argv=None):
    args = parse_args(argv)
    el = args.element
    prefix = os.getcwd() + '/' + '%s.%s.%s.lammps' % (args.name, el.lower(), LAT[el])
    p = np.load(prefix + '.virial.trgt.npy')
    t = np.load(prefix + '.temp.trgt.npy')
    pn, tn = p.size, t.size
    if args.verbose:
        print('parsing data for %s for %d pressure indices and %d temperature indices' % (el.lower(), pn, tn))
    rows = read_thrm(prefix + '.thrm')
    sn = 0
    for c, name in enumerate(COLUMNS):
        a = np.ascontiguousarray(rows[:, c]).reshape(pn, tn, -1)
        sn = a.shape[2]
        np.save(prefix + '.%s.npy' % name, a)
    if args.verbose:
        print('%d thermodynamic property steps parsed' % (pn * tn * sn))
    del rows
    natoms, box, x = read_traj(prefix + '.traj')
    natoms = natoms.reshape(pn, tn, -1)
    x = x.reshape(pn, tn, natoms.shape[2], natoms[0, 0, 0], 3)
    if args.verbose:
        print('%d trajectory steps parsed' % (pn * tn * sn))
    np.save(prefix + '.natoms.npy', natoms)
    np.save(prefix + '.box.npy', box)
    np.save(prefix + '.pos.npy', x)
    if args.verbose:
        print('all properties pickled')


if __name__ == '__main__':
    main()
