"""dev probe (experiment build): timeline of slot 0's evaluations across its 4 workgroups x 8 waves, 100 MHz clock"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['NM_HIP_LIB'] = os.environ.get('NM_EXP_LIB', os.path.join(ROOT, 'neuralmelting_amd', 'libnm_hip_exp.so'))
import numpy as np
import neuralmelting_amd as nm
from neuralmelting_amd import lattice, _lib

ROWS = int(os.environ.get('NM_TL_ROWS', '8'))   # 8 rows x 8 = 64 replicas (Q = 4); 4 rows = 32 replicas (Q = 8)
P = np.linspace(1, 8, 8, dtype=np.float32); T = np.linspace(.25, 2.5, 8, dtype=np.float32)
if os.environ.get('NM_TL_TREV'):   # the temperature grid upside down: the timeline's slot 0 is then the HOTTEST replica of the first pressure row
    T = T[::-1].copy()
x, v, box, d = lattice.init_states(4, P, T, 0.03125, 0.03125, row0=0, nrows=ROWS)
EQ = int(os.environ.get('NM_TL_EQUIL', '0'))   # > 0: the bench's moves (PMC / VMC 0.125 each, 8 steps), that many cycles of 128 moves first, with the
# exchange — the timeline is then that of an equilibrated replica (slot 0 after the last sweep), list rebuilds included
if EQ:
    e = nm.Engine(256, P, T, row0=0, nrows=ROWS)
    e.set_state(x, v, box, d)
    for s in range(EQ):
        e.set_step(s); e.run_block(128); e.adapt(); e.exchange(count=False)
else:
    e = nm.Engine(256, P, T, ppos=0.0, pvol=0.0, nstps=16, row0=0, nrows=ROWS)
    e.set_state(x, v, box, d)
    for s in range(3):
        e.set_step(s); e.run_block(24); e.adapt()
e.synchronize()
L = _lib.load()
L.nm_tline_get.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros((8, 8, 512, 8), dtype=np.uint64)   # [q][wave][eval][point]
L.nm_tline_get(e.h, buf.ctypes.data)
Q = e.cus_per_replica
t = buf[:Q].astype(np.int64)
n = int((t[0, 0, :, 0] > 0).sum())
print('evals recorded', n, 'Q', Q)
names = ['entry', 'after check', 'after rebuild', 'pair loop done', 'exchange done(+barrier)', 'granules read']
ev = range(20, min(n - 1, 400))
def span(a, b, red=np.max):
    return np.array([red(t[:, :, k, b]) - np.min(t[:, :, k, a]) for k in ev]) * 10.0   # ns
print('per evaluation (ns), all 32 waves of the cluster on one axis:')
print('  first entry -> last wave past check    %7.0f' % np.median(span(0, 1)))
print('  last check -> last pair loop done      %7.0f' % np.median(np.array([np.max(t[:, :, k, 3]) - np.max(t[:, :, k, 2]) for k in ev]) * 10.0))
print('  pair loop per wave: median %7.0f  max-over-waves %7.0f  min-over-waves %7.0f' % (
    np.median((t[:, :, list(ev), 3] - t[:, :, list(ev), 2]) * 10.0),
    np.median(np.max(t[:, :, list(ev), 3] - t[:, :, list(ev), 2], axis=(0, 1)) * 10.0),
    np.median(np.min(t[:, :, list(ev), 3] - t[:, :, list(ev), 2], axis=(0, 1)) * 10.0)))
d = t[:, :, list(ev), :]
print('  inside the force-only pair loop, per wave (median / max over waves): prologue issued %5.0f / %5.0f   neighbours %5.0f / %5.0f   epilogue %5.0f / %5.0f' % (
    np.median((d[..., 1] - d[..., 2]) * 10.0), np.median(np.max(d[..., 1] - d[..., 2], axis=(0, 1)) * 10.0),
    np.median((d[..., 5] - d[..., 1]) * 10.0), np.median(np.max(d[..., 5] - d[..., 1], axis=(0, 1)) * 10.0),
    np.median((d[..., 3] - d[..., 5]) * 10.0), np.median(np.max(d[..., 3] - d[..., 5], axis=(0, 1)) * 10.0)))
print('  pair done -> granules read (per wave)  median %7.0f  max %7.0f' % (
    np.median((t[:, :, list(ev), 5] - t[:, :, list(ev), 3]) * 10.0), np.median(np.max(t[:, :, list(ev), 5] - t[:, :, list(ev), 3], axis=(0, 1)) * 10.0)))
print('  last pair done -> last exchange done   %7.0f' % np.median(np.array([np.max(t[:, :, k, 4]) - np.max(t[:, :, k, 3]) for k in ev]) * 10.0))
print('  exchange done -> next entry (kick etc) %7.0f' % np.median(np.array([np.min(t[:, :, k + 1, 0]) - np.max(t[:, :, k, 4]) for k in ev]) * 10.0))
print('  entry -> next entry                    %7.0f' % np.median(np.array([np.min(t[:, :, k + 1, 0]) - np.min(t[:, :, k, 0]) for k in ev]) * 10.0))
print('  hand-over: last pair done -> granules in (per wave) median %7.0f max %7.0f ; -> past its barrier %7.0f' % (
    np.median(np.array([np.median(t[:, :, k, 6]) - np.max(t[:, :, k, 3]) for k in ev]) * 10.0),
    np.median(np.array([np.max(t[:, :, k, 6]) - np.max(t[:, :, k, 3]) for k in ev]) * 10.0),
    np.median(np.array([np.max(t[:, :, k, 7]) - np.max(t[:, :, k, 3]) for k in ev]) * 10.0)))
print('  barrier after the pair loop: last pair done -> last wave past it %7.0f' % np.median(np.array([np.max(t[:, :, k, 4]) - np.max(t[:, :, k, 3]) for k in ev]) * 10.0))
print('  hand-over barrier -> next entry %7.0f' % np.median(np.array([np.min(t[:, :, k + 1, 0]) - np.max(t[:, :, k, 7]) for k in ev]) * 10.0))
wg_done = np.array([[np.max(t[q, :, k, 3]) for q in range(Q)] for k in ev])
print('  skew between workgroups at pair-loop end (max-min) %7.0f' % np.median((wg_done.max(1) - wg_done.min(1)) * 10.0))
flat = buf.reshape(-1)
print('shader clock held during the block: %.0f MHz' % (float(flat[-2]) / float(flat[-1]) * 100.0))
e.close()
# per wave of workgroup q: medians of the stamps relative to the cluster's last pair-loop end
e_ = list(ev)
last3 = np.max(t[:, :, e_, 3], axis=(0, 1))
for q in range(Q):
    print('wg %d  wave: pair-end  barrierA  granules-in  barrierB  next-entry  (ns after the last pair-loop end of the cluster)' % q)
    for w in range(8):
        vals = [np.median((t[q, w, e_, k] - last3) * 10.0) for k in (3, 4, 6, 7)]
        nxt = np.median((t[q, w, [k + 1 for k in e_], 0] - last3) * 10.0)
        print('      %d   %7.0f %7.0f %7.0f %7.0f %7.0f' % (w, vals[0], vals[1], vals[2], vals[3], nxt))
# evaluations that rebuilt their list: entry -> "after rebuild" stamp of the workgroup's slowest wave, sorted (the short ones are plain entries)
ent = np.array([np.max(t[0, :, k, 2]) - np.min(t[0, :, k, 0]) for k in e_]) * 10.0
srt = np.sort(ent)
print('entry -> past the rebuild decision / rebuild, workgroup 0, ns: median %.0f; evaluations above 1 us: %d of %d, their median %.0f, min %.0f'
      % (np.median(ent), int((ent > 1000).sum()), len(ent), np.median(ent[ent > 1000]) if (ent > 1000).any() else 0, ent[ent > 1000].min() if (ent > 1000).any() else 0))
