"""bench.py — MC sweeps/sec of the NPT-HMC + replica-exchange hot path (BASELINE.json metric).

A step = one cycle of the reference's main loop (remcmc:977-995) with outputs off (-sc == -sn, as run.sh:7):
gen_samples (MOD moves per replica) -> gen_mc_params -> replica_exchange.  One sweep = one move_mc call on one
replica, so a step is NS*MOD sweeps.  Workload at N=1: BASELINE configs[1] — LJ, 4^3 cells (256 atoms), 8x8 PxT grid,
64 replicas resident in HBM.  For N>1 every rank owns 8 pressure rows x 8 temperatures of an (8N)x8 grid
(weak scaling; the exchange never crosses pressure rows, so there is no data-path collective).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VEC_PEAK_TF = 78.6    # MI355X fp64 vector peak (spec; = FP32 vector 157.3 / 2)
FLOP_PER_PAIR = 40.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--sz', type=int, default=4, help='supercell size (-ss)')
    ap.add_argument('--rows', type=int, default=8, help='pressure rows per GPU')
    ap.add_argument('--tn', type=int, default=8, help='temperatures (-tn)')
    ap.add_argument('--mod', type=int, default=128, help='moves per block (-sm)')
    ap.add_argument('--el', type=str, default='LJ', help="element (-e): LJ, or Al = BASELINE config 4 (Sutton-Chen EAM, metal units)")
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-cycles', type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...'
                             % (args.gpus, args.gpus))
    import torch
    dist = None
    if world > 1 or 'RANK' in os.environ:  # under torch.distributed.run: one rank per GPU over RCCL (also at N=1)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # NM_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share cards;
        # RCCL refuses two ranks on one device).  The driver's runs use RCCL.
        backend = os.environ.get('NM_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            dist.init_process_group(backend)

    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice

    npn = args.rows * world
    P = np.linspace(1.0, 8.0, npn, dtype=np.float32)
    T = np.linspace(0.25, 2.5, args.tn, dtype=np.float32) if args.el == 'LJ' else np.linspace(256.0, 2560.0, args.tn, dtype=np.float32)
    natoms = 4 * args.sz ** 3
    row0 = rank * args.rows
    x, v, box, d = lattice.init_states(args.sz, P, T, 0.03125, 0.03125, el=args.el, row0=row0, nrows=args.rows)
    eng = nm.Engine(natoms, P, T, element=args.el, device=local, row0=row0, nrows=args.rows, ppos=0.125, pvol=0.125, nstps=8,
                    bulk=True, seed=256)
    eng.set_state(x, v, box, d)
    ns = eng.nslots

    def cycle(step, last=False):
        eng.set_step(step)
        eng.run_block(args.mod)
        eng.adapt()
        if not last:
            eng.exchange(count=False)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step = 0
    for _ in range(args.warmup):
        cycle(step)
        step += 1
    fence()
    eng.timing_reset()
    eng.stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cycle(step)
        step += 1
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    launches, kms = eng.timing()
    st = eng.stats()
    # one more block outside the timed region, read before gen_mc_params zeroes the counters: the per-replica acceptance
    # ratios and U, V the metric's definition asks to see next to the rate
    eng.set_step(step)
    eng.run_block(args.mod)
    last = eng.thermo()
    sweeps_total = world * ns * args.mod * args.steps
    value = sweeps_total / dt

    out = None
    if rank == 0:
        # dominant kernel: nm_block_kernel, one launch per step, ns*mod sweeps per launch
        phmc = 1.0 - 0.125 - 0.125
        bytes_per_sweep = 48.0 * natoms + 24.0 * natoms * phmc          # SURVEY.md §8d compulsory HBM bytes
        k_avg_s = (kms / max(launches, 1)) * 1e-3
        sweeps_per_launch = ns * args.mod
        achieved_gbs = bytes_per_sweep * sweeps_per_launch / k_avg_s / 1e9
        evals = st[:, 0].sum()
        mean_pairs = st[:, 3].sum() / max(st[:, 2].sum(), 1.0)
        flops = evals * mean_pairs * FLOP_PER_PAIR
        tf = flops / (kms * 1e-3) / 1e12
        out = {
            'metric': 'MC sweeps/sec (whole node), %s 4^3 cells, 8x8 PxT grid' % args.el,
            'value': value, 'unit': 'sweeps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s %d^3 cells (%d atoms), %dx%d PxT grid per GPU, MOD=%d, bulk PMC 0.125 / VMC 0.125 / '
                                   'HMC 0.75 x %d steps, outputs off' % (args.el, args.sz, natoms, args.rows, args.tn, args.mod, 8),
                       'replicas_per_gpu': ns, 'sweeps_per_step': world * ns * args.mod, 'parallelism': 'rows/gpu'},
            'roofline': {'bound': 'hbm', 'achieved': achieved_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved_gbs / HBM_PEAK_GBS, 'traffic': measured_traffic(natoms, ns, args.mod),
                         'kernel': 'nm_block_kernel', 'kernel_avg_ms': k_avg_s * 1e3, 'launches': launches,
                         'algorithmic_bytes_per_launch': bytes_per_sweep * sweeps_per_launch,
                         'note': 'LDS-resident by design: the binding ceiling is fp64 VALU/latency on the CUs that hold a '
                                 'replica, see fp64'},
            # executed work (evaluations the kernel really made) and SURVEY.md §8d's algorithmic count
            # EVALS = PPOS + PVOL + PHMC (NSTPS + 1) = 7.0 per sweep at the defaults (the kernel makes fewer: forces are kept
            # across accepted and rejected moves)
            'fp64': {'achieved': tf, 'peak': FP64_VEC_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / FP64_VEC_PEAK_TF,
                     'algorithmic_evals_per_sweep': 0.125 + 0.125 + phmc * 9.0,
                     'achieved_algorithmic': (0.125 + 0.125 + phmc * 9.0) * sweeps_per_launch * mean_pairs * FLOP_PER_PAIR
                                             / k_avg_s / 1e12,
                     'evals_per_sweep': evals / (ns * args.mod * args.steps), 'mean_pairs_per_eval': mean_pairs,
                     'flop_per_pair': FLOP_PER_PAIR, 'list_rebuilds_per_sweep': st[:, 1].sum() / (ns * args.mod * args.steps),
                     'cus_per_replica': eng.cus_per_replica, 'cus_occupied': ns * eng.cus_per_replica, 'cus_total': 256},
        }
        out['replicas'] = {'note': 'rank 0, block after the timed region, slot k = i*NT + j (pressure i, temperature j)',
                           'accept_pmc': [round(float(a), 3) for a in last[:, 14]],
                           'accept_vmc': [round(float(a), 3) for a in last[:, 15]],
                           'accept_hmc': [round(float(a), 3) for a in last[:, 16]],
                           'pe_per_atom': [round(float(a) / natoms, 4) for a in last[:, 1]],
                           'vol_per_atom': [round(float(a) / natoms, 4) for a in last[:, 4]]}
        if not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(eng, natoms, args, T, P, row0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()
    if rank == 0:
        print(json.dumps(out))


def measured_traffic(natoms, ns, mod):
    """HBM bytes per launch of nm_block_kernel from the committed rocprofv3 PMC passes (profiles/), for the workload
    they were taken on; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  None for other workloads."""
    f = os.path.join(ROOT, 'profiles', 'r01_pmc_block_kernel_final.json')
    if not (os.path.isfile(f) and natoms == 256 and ns == 64 and mod == 128):
        return None
    d = json.load(open(f))
    return (2.0 * d['FETCH_SIZE']['mean'] + d['WRITE_SIZE']['mean']) * 1024.0


def cpu_baseline(eng, natoms, args, T, P, row0):
    """the oracle (C restatement, OpenMP over replicas) timed on the host cores on the engine's current states"""
    from oracle import oracle as O
    from neuralmelting_amd import lattice
    O.build()
    x, v, box, d = eng.get_state()
    et, pf = eng.constants()
    ns = eng.nslots
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, ns)  # one replica per thread: more threads than replicas would idle
    t0 = time.perf_counter()
    for c in range(args.cpu_cycles):
        kw = dict(units=1, mass=lattice.MASS['Al'], pot=1) if args.el == 'Al' else {}
        tq = np.tile(T.astype(np.float64), args.rows)
        out = O.run_blocks(x, v, box, d, tq, et, pf, natoms=natoms, mod=args.mod, nstps=8, bulk=True, ppos=0.125,
                           pvol=0.125, lat=lattice.LAT[args.el][1], seed=256, slot0=row0 * len(T), step=1000 + c, nthreads=cores, **kw)
        x, v, box = out['x'], out['v'], out['box']
    dt = time.perf_counter() - t0
    return {'value': ns * args.mod * args.cpu_cycles / dt, 'unit': 'sweeps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d cycles of the same %d-replica workload (MOD=%d) from the GPU run\'s post-warm-up states, '
                      'OpenMP over replicas, %.1f s wall' % (args.cpu_cycles, ns, args.mod, dt)}


if __name__ == '__main__':
    main()
