"""Structural histograms on the GPU with lammps_distr.py's command line and outputs.

Mirrors /root/reference/scripts/lammps_distr.py ("distr"): loads <PREFIX>.natoms/.box/.pos.npy written by
lammps_parse.py, computes the per-sample radial distribution over the 27 periodic images (calculate_rdf, distr:123-135)
and the 3-D histogram of pair displacement vectors (calculate_cdf, distr:161-171) — here one kernel launch over all
samples (include/nm_distr.h) instead of a Dask/joblib map of numpy calls — and writes the same .dni/.r/.rdf/.dn/.rv/.cdf
files with the same shapes and dtypes.

    python -m neuralmelting_amd.distr -v -n remcmc_init -e LJ -sb 64 -cb 16
"""
import argparse
import ctypes as C
import os

import numpy as np

from . import _lib as B

LAT = {'Ti': 'bcc', 'Al': 'fcc', 'Ni': 'fcc', 'Cu': 'fcc', 'LJ': 'fcc'}


def parse_args(argv=None):
    """lammps_distr.py's flags (distr:15-53); the cluster flags are accepted and ignored"""
    p = argparse.ArgumentParser()
    p.add_argument('-v', '--verbose', action='store_true')
    p.add_argument('-p', '--parallel', action='store_true')
    p.add_argument('-c', '--client', action='store_true')
    p.add_argument('-d', '--distributed', action='store_true')
    p.add_argument('-q', '--queue', type=str, default='jobqueue')
    p.add_argument('-a', '--allocation', type=str, default='startup')
    p.add_argument('-nn', '--nodes', type=int, default=1)
    p.add_argument('-np', '--procs_per_node', type=int, default=16)
    p.add_argument('-w', '--walltime', type=int, default=2)
    p.add_argument('-m', '--memory', type=int, default=32)
    p.add_argument('-nw', '--workers', type=int, default=16)
    p.add_argument('-nt', '--threads', type=int, default=1)
    p.add_argument('-mt', '--method', type=str, default='fork')
    p.add_argument('-n', '--name', type=str, default='remcmc_init')
    p.add_argument('-e', '--element', type=str, default='LJ')
    p.add_argument('-sb', '--spherical_bins', type=int, default=64)
    p.add_argument('-cb', '--cartesian_bins', type=int, default=16)
    return p.parse_args(argv)


def calculate_spatial(natoms, box, sbins, cbins):
    """the domains of calculate_spatial (distr:73-120), same numpy expressions: returns nrho, dni, r, dn, rv"""
    nrho = np.divide(natoms, np.power(box, 3))
    l = np.min(box)
    mr = 1 / 2
    r = np.linspace(1e-16, mr, sbins)
    dr = r[1] - r[0]
    dv = 4 * np.pi * np.square(r) * dr
    r = r * l
    dv = dv * l ** 3
    dni = np.multiply(nrho[:, np.newaxis], dv[np.newaxis, :])
    cb = np.array(3 * (cbins + 1,))
    rv = np.array([np.linspace(0, l, cb[i]) for i in range(len(cb))])
    rv -= l / 2
    drv = rv[0, 1] - rv[0, 0]
    dn = nrho * drv ** 3
    return nrho, dni, r, dn, rv


def histograms(natoms, box, pos, r, rv, device=0, want_rdf=True, want_cdf=True):
    """raw counts of calculate_rdf / calculate_cdf for all samples, divided by natoms as the reference does:
    rdf[ns][sbins] float32, cdf[ns][cb][cb][cb] float32"""
    L = B.load()
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    box = np.ascontiguousarray(box, dtype=np.float32)
    ns, n = pos.shape[0], pos.shape[1]
    r = np.ascontiguousarray(r, dtype=np.float64)
    ve = np.ascontiguousarray(rv[0], dtype=np.float64)
    if not (np.array_equal(rv[0], rv[1]) and np.array_equal(rv[0], rv[2])):
        raise ValueError('the three cartesian axes must share their bin edges (they do in lammps_distr.py)')
    sb, cb = len(r), len(ve) - 1
    rdf = np.zeros((ns, sb), dtype=np.float32) if want_rdf else None
    cdf = np.zeros((ns, cb, cb, cb), dtype=np.float32) if want_cdf else None
    fp = lambda a: None if a is None else a.ctypes.data_as(B.c_float_p)
    rc = L.nm_distr_histograms(device, ns, n, fp(pos), fp(box), sb, r.ctypes.data_as(B.c_double_p), cb,
                               ve.ctypes.data_as(B.c_double_p), fp(rdf), fp(cdf))
    if rc != 0:
        raise RuntimeError('nm_distr_histograms failed (%d): %s' % (rc, L.nm_distr_last_error().decode()))
    na = np.asarray(natoms).reshape(-1)
    if want_rdf:
        rdf = rdf / na[:, None]                           # rd/natoms (distr:135): float32 / uint16 -> float32
    if want_cdf:
        cdf = cdf / na[:, None, None, None]               # cd/natoms (distr:171)
    return rdf, cdf


def main(argv=None):
    a = parse_args(argv)
    prefix = os.getcwd() + '/' + '%s.%s.%s.lammps' % (a.name, a.element.lower(), LAT[a.element])
    P = np.load(prefix + '.virial.trgt.npy')
    T = np.load(prefix + '.temp.trgt.npy')
    pn, tn = P.size, T.size
    natoms = np.load(prefix + '.natoms.npy').reshape(-1)                      # load_data, distr:63-70
    box = np.load(prefix + '.box.npy').reshape(-1)
    pos = np.load(prefix + '.pos.npy').reshape(-1, natoms[0], 3)
    ns = natoms.size
    nrho, dni, r, dn, rv = calculate_spatial(natoms, box, a.spherical_bins, a.cartesian_bins)
    rns = np.int32(ns / (pn * tn))
    if a.verbose:
        print('computing %s %s samples' % (ns, a.element.lower()))
    rdf, cdf = histograms(natoms, box, pos, r, rv, device=int(os.environ.get('LOCAL_RANK', '0')))
    g = np.divide(np.array(rdf, dtype=np.float32), dni)                        # distr:305-311
    np.save(prefix + '.dni.npy', dni.reshape(pn, tn, rns, r.size))
    np.save(prefix + '.r.npy', r)
    np.save(prefix + '.rdf.npy', g.reshape(pn, tn, rns, r.size))
    c = np.divide(np.array(cdf, dtype=np.float32), dn[:, np.newaxis, np.newaxis, np.newaxis])   # distr:361-362
    np.save(prefix + '.dn.npy', dn)
    np.save(prefix + '.rv.npy', rv)
    np.save(prefix + '.cdf.npy', c.reshape(pn, tn, rns, *(3 * (rv.shape[1] - 1,))))
    if a.verbose:
        print('all properties pickled')


if __name__ == '__main__':
    main()
