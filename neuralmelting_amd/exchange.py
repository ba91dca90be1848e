"""Replica exchange when a pressure row is split across GPUs (NP < number of ranks).

With whole rows per GPU the sweep is local and runs on the device (nm_exchange).  When a row spans ranks the sweep's
inputs — (E_tot, V) per slot, 16 bytes each — are all-gathered over RCCL (xGMI), every rank runs the identical sweep with the
shared Philox stream (global pair index, same draws as nm_exchange_kernel) and so knows the same permutation; only the
replicas that changed slot then move: inside a rank as a local re-seat, between ranks point to point (one send / receive of
6 N + 9 doubles per replica: 12 KB at 256 atoms, 98 KB at 2048), never the other replicas' state; on each rank the replicas that
leave are read in one batch and the ones that arrive written in one batch (nm_get_slots / nm_set_slots: one settle, one wait each).  All messages are
latency-sized."""
import os

import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
S_EXCH = 7


def philox4x32_10(ctr, key):
    """Philox4x32-10 (same constants and round structure as csrc/nm_device.h)"""
    c0, c1, c2, c3 = [int(v) & 0xFFFFFFFF for v in ctr]
    k0, k1 = [int(v) & 0xFFFFFFFF for v in key]
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & 0xFFFFFFFF, p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u01(hi, lo):
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def sweep(npn, nt, seed, step, etot, vol, et, pf):
    """replica_exchange (remcmc:776-803) over the whole grid; returns (perm, swaps): perm[k] = slot whose state list
    entries [0..11] end up in slot k.  Bit-for-bit the decisions of nm_exchange_kernel."""
    etot = np.array(etot, dtype=np.float64).copy()
    vol = np.array(vol, dtype=np.float64).copy()
    perm = np.arange(npn * nt)
    ppr = nt * (nt - 1) // 2
    swaps = 0
    for u in range(npn):
        q = 0
        for v in range(nt - 1, -1, -1):
            for w in range(v):
                i, j = u * nt + v, u * nt + w
                de = etot[i] - etot[j]
                dv = vol[i] - vol[j]
                dh = de * (1.0 / et[i] - 1.0 / et[j]) + (pf[i] - pf[j]) * dv
                o = philox4x32_10((u * ppr + q, S_EXCH, 0, step), (seed, 0xFFFFFFFF))
                uu = u01(o[0], o[1])
                with np.errstate(over='ignore'):
                    e = np.exp(dh)
                mm = e if e != e else min(e, 1.0)
                if uu <= mm:
                    swaps += 1
                    etot[[i, j]] = etot[[j, i]]
                    vol[[i, j]] = vol[[j, i]]
                    perm[[i, j]] = perm[[j, i]]
                q += 1
    return perm, swaps


def allgather(arr, group_info):
    """all-gather equally shaped float64 arrays over the process group (RCCL on GPUs, gloo on CPUs)"""
    import torch
    import torch.distributed as dist
    world, use_cuda = group_info
    if world == 1 and not dist.is_initialized():   # a single process without a group (forced split-row mode in tests)
        return np.array(arr, dtype=np.float64)
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    if use_cuda:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy().reshape(1, *arr.shape) for o in out]).reshape(-1, *arr.shape[1:])


def transfers(perm, nloc):
    """what has to move for `perm` when every rank holds `nloc` consecutive slots: a list of (dst_slot, src_slot) for every slot
    whose content changes, in ascending dst order — the order both ends of every point-to-point message use"""
    return [(int(k), int(s)) for k, s in enumerate(perm) if int(s) != k]


def exchange_split(eng, step, npn, nt, seed, k0, natoms, et, pf, group_info, rank=None):
    """replica_exchange (remcmc:776-803) for a context that holds a partial pressure row (slots k0 .. k0+eng.nslots of the grid).
    Returns the number of swaps of the whole sweep.  Entries [0..11] of the state lists travel (remcmc:798): x, v, box, dx dv dt
    and the thermo scalars — for the replicas that swapped only."""
    import torch
    import torch.distributed as dist
    world, use_cuda = group_info
    nloc = eng.nslots
    rows = eng.thermo()
    ev = allgather(np.stack([rows[:, 1] + rows[:, 2], rows[:, 4]], axis=1), group_info)          # (E_tot, V): 16 B per slot
    perm, swaps = sweep(npn, nt, seed, step, ev[:, 0], ev[:, 1], et, pf)
    if not swaps:
        return swaps
    moves = transfers(perm, nloc)
    me = k0 // nloc if rank is None else rank
    n3 = 3 * natoms
    width = 2 * n3 + 9
    # what this rank gives away or re-seats locally: read in ONE batch (nm_get_slots: one settle, one wait) before anything is overwritten
    mine = sorted({src for _, src in moves if src // nloc == me})
    out = {}
    if mine:
        x, v, box, d, th = eng.get_slots([s_ - k0 for s_ in mine])
        for q, s_ in enumerate(mine):
            out[s_] = np.concatenate([x[q], v[q], box[q:q + 1], d[q], th[q]])
    recv, ops = {}, []
    for dst, src in moves:                              # ascending dst on every rank: matching order of sends and receives
        rs, rd = src // nloc, dst // nloc
        if rs == rd:
            continue
        if rs == me:
            t = torch.from_numpy(out[src])
            ops.append(dist.P2POp(dist.isend, t.cuda() if use_cuda else t, rd))
        elif rd == me:
            t = torch.empty(width, dtype=torch.float64, device='cuda' if use_cuda else 'cpu')
            recv[dst] = t
            ops.append(dist.P2POp(dist.irecv, t, rs))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    arrive = [(dst, src) for dst, src in moves if dst // nloc == me]
    log = os.environ.get('NM_LOG_EXCHANGE')  # tests: which backend and device the collective used, how many replicas were re-seated
    if log and me == 0:
        with open(log, 'a') as f:
            f.write('%s %s %d\n' % (dist.get_backend() if dist.is_initialized() else 'none', 'cuda' if use_cuda else 'cpu', len(arrive)))
    if arrive:                                          # everything that lands here, again as ONE batch (nm_set_slots)
        rows_in = np.stack([out[src] if src // nloc == me else recv[dst].cpu().numpy() for dst, src in arrive])
        eng.set_slots([dst - k0 for dst, _ in arrive], rows_in[:, :n3], rows_in[:, n3:2 * n3], rows_in[:, 2 * n3],
                      rows_in[:, 2 * n3 + 1:2 * n3 + 4], rows_in[:, 2 * n3 + 4:2 * n3 + 9])
    return swaps
