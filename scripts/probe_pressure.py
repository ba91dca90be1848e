"""scratch: NPT virial pressure incl. the impulsive term of the unshifted cutoff vs the imposed pressure (C2 grid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice
N = 256
P = np.linspace(1.0, 8.0, 8, dtype=np.float32)
T = np.linspace(0.25, 2.5, 8, dtype=np.float32)
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125)
e = nm.Engine(N, P, T)
e.set_state(x, v, box, d)
mod, burn, cycles = 64, 24, 96
rc, dl = 2.5, float(os.environ.get("DL", "0.02"))
urc = 4.0 * (rc ** -12 - rc ** -6)
samples, shell = [], []
iu = np.triu_indices(N, 1)
for step in range(burn + cycles):
    e.set_step(step); e.run_block(mod)
    if step >= burn:
        samples.append(e.thermo())
        xs, _, bs, _ = e.get_state(velocities=False)
        ns = []
        for k in range(64):
            q = xs[k].reshape(N, 3)
            dd = q[:, None, :] - q[None, :, :]
            dd -= bs[k] * np.rint(dd / bs[k])
            r = np.sqrt((dd * dd).sum(-1)[iu])
            ns.append(((r > rc - dl) & (r < rc + dl)).sum())
        shell.append(ns)
    e.adapt(); e.exchange(count=False)
e.close()
r = np.array(samples); shell = np.array(shell, dtype=float)
tkin, press, vol = r[:, :, 0], r[:, :, 3], r[:, :, 4]
Tj = np.tile(T.astype(np.float64), 8)[None, :]
Pi = np.repeat(P.astype(np.float64), 8)
pvir = press + (N * Tj - (N - 1.0) * tkin) / vol
pimp = urc * rc / (3.0 * vol) * shell / (2 * dl)
pest = pvir + pimp
nb = 8
for name, arr in (('virial', pvir), ('virial+impulsive', pest)):
    bm = arr.reshape(nb, cycles // nb, 64).mean(1)
    se = bm.std(0, ddof=1) / np.sqrt(nb)
    z = (bm.mean(0) - Pi) / se
    print(name, 'mean z', z.mean(), 'max |z|', np.abs(z).max(), 'mean diff', (bm.mean(0) - Pi).mean())
    print(np.round((bm.mean(0) - Pi).reshape(8, 8), 3))
    print(np.round(z.reshape(8, 8), 1))
print('tkin ratio', (tkin / Tj).mean(), (tkin / Tj).std(ddof=1) / np.sqrt(tkin.size))
print('density', np.round((N / vol.mean(0)).reshape(8, 8), 3))
