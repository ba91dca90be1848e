"""structural histograms: oracle and HIP kernels against what the reference's own calculate_rdf /
calculate_cdf return (tests/golden/ref_distr.npz).  Integer counts: bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'ref_distr.npz'))
KEYS = sorted({k.rsplit('_', 1)[0] for k in G.files})


@pytest.mark.parametrize('key', KEYS)
def test_oracle_matches_reference_outputs(key):
    from oracle import distr_oracle as D
    pos, box, natoms, r, rv = (G[key + '_' + n] for n in ('pos', 'box', 'natoms', 'r', 'rv'))
    for i in range(len(pos)):
        np.testing.assert_array_equal(D.calculate_rdf(natoms[i], box[i], pos[i], r), G[key + '_rdf'][i])
        np.testing.assert_array_equal(D.calculate_cdf(natoms[i], box[i], pos[i], rv), G[key + '_cdf'][i])


@pytest.mark.parametrize('key', KEYS)
def test_spatial_domains_match(key):
    from neuralmelting_amd import distr
    sb = len(G[key + '_r']); cb = G[key + '_rv'].shape[1] - 1
    nrho, dni, r, dn, rv = distr.calculate_spatial(G[key + '_natoms'], G[key + '_box'], sb, cb)
    np.testing.assert_array_equal(r, G[key + '_r'])
    np.testing.assert_array_equal(rv, G[key + '_rv'])


@pytest.mark.gpu
@pytest.mark.parametrize('key', KEYS)
def test_hip_histograms_bit_exact(key):
    from neuralmelting_amd import distr
    pos, box, natoms, r, rv = (G[key + '_' + n] for n in ('pos', 'box', 'natoms', 'r', 'rv'))
    rdf, cdf = distr.histograms(natoms, box, pos, r, rv)
    assert rdf.dtype == np.float32 and cdf.dtype == np.float32
    np.testing.assert_array_equal(rdf, G[key + '_rdf'])
    np.testing.assert_array_equal(cdf, G[key + '_cdf'])
    assert rdf.sum() > 0 and cdf.sum() > 0


@pytest.mark.gpu
def test_hip_histograms_many_samples_vs_oracle():
    """more samples than one launch chunk would need + totals: every pair of every image lands in exactly one cdf bin or outside"""
    from neuralmelting_amd import distr
    from oracle import distr_oracle as D
    rng = np.random.default_rng(3)
    ns, n = 40, 256
    box = (6.0 + rng.random(ns)).astype(np.float32)
    pos = (rng.random((ns, n, 3)) * box[:, None, None]).astype(np.float32)
    natoms = np.full(ns, n, dtype=np.uint16)
    nrho, dni, r, dn, rv = distr.calculate_spatial(natoms, box, 64, 16)
    rdf, cdf = distr.histograms(natoms, box, pos, r, rv)
    for i in (0, 17, 39):
        np.testing.assert_array_equal(rdf[i], D.calculate_rdf(natoms[i], box[i], pos[i], r))
        np.testing.assert_array_equal(cdf[i], D.calculate_cdf(natoms[i], box[i], pos[i], rv))
    assert (cdf.reshape(ns, -1).sum(1) * n <= 27 * n * n).all()


@pytest.mark.gpu
def test_distr_cli_files(tmp_path):
    """python -m neuralmelting_amd.distr on lammps_parse-style inputs writes the reference's six files with its shapes"""
    from neuralmelting_amd import distr
    rng = np.random.default_rng(5)
    pn, tn, sn, n = 2, 2, 3, 256
    pref = str(tmp_path / 'd1.lj.fcc.lammps')
    np.save(pref + '.virial.trgt.npy', np.linspace(1, 8, pn, dtype=np.float32))
    np.save(pref + '.temp.trgt.npy', np.linspace(0.25, 2.5, tn, dtype=np.float32))
    box = (6.0 + rng.random(pn * tn * sn)).astype(np.float32)
    np.save(pref + '.natoms.npy', np.full((pn, tn, sn), n, dtype=np.uint16))
    np.save(pref + '.box.npy', box)
    np.save(pref + '.pos.npy', (rng.random((pn, tn, sn, n, 3)) * 6.0).astype(np.float32))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        distr.main(['-n', 'd1', '-e', 'LJ', '-sb', '32', '-cb', '8'])
    finally:
        os.chdir(cwd)
    # dtypes follow the same numpy expressions as the reference (float32 counts / float64 dni -> float64; dn's dtype depends
    # on the numpy version's promotion of np.linspace(0, float32, n))
    assert np.load(pref + '.rdf.npy').shape == (pn, tn, sn, 32) and np.load(pref + '.rdf.npy').dtype == np.float64
    assert np.load(pref + '.cdf.npy').shape == (pn, tn, sn, 8, 8, 8)
    assert np.load(pref + '.dni.npy').shape == (pn, tn, sn, 32) and np.load(pref + '.r.npy').shape == (32,)
    assert np.load(pref + '.rv.npy').shape == (3, 9) and np.load(pref + '.dn.npy').shape == (pn * tn * sn,)
