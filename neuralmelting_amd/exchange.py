"""Replica exchange when a pressure row is split across GPUs (NP < number of ranks).

With whole rows per GPU the sweep is local and runs on the device (nm_exchange).  When a row spans ranks the sweep's
inputs — (E_tot, V) per slot, 16 bytes each — are all-gathered over RCCL (xGMI), every rank runs the identical sweep with the
shared Philox stream (global pair index, same draws as nm_exchange_kernel), and the configurations that changed owner are
all-gathered and re-seated.  All messages are latency-sized."""
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
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    if use_cuda:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy().reshape(1, *arr.shape) for o in out]).reshape(-1, *arr.shape[1:])
