"""bench.py — MC sweeps/sec of the NPT-HMC + replica-exchange hot path (BASELINE.json metric).

A step = one cycle of the reference's main loop (remcmc:977-995) with outputs off (-sc == -sn, as run.sh:7):
gen_samples (MOD moves per replica) -> gen_mc_params -> replica_exchange.  One sweep = one move_mc call on one
replica, so a step is NS*MOD sweeps.

Workloads (--config; BASELINE.json `configs`):
  C2    LJ 4^3 cells (256 atoms), 8x8 PxT grid: the configuration the metric is quoted on (default)
  C3    LJ 6^3 cells (864 atoms), 16x16 grid over 8 GPUs: one GPU's share = 2 pressure rows x 16 temperatures
  C4    Al (Sutton-Chen EAM, metal units) 4^3 cells, 8x8 grid
  C5    LJ 8^3 cells (2048 atoms), 32x32 grid over 8 GPUs: one GPU's share = 4 rows x 32 temperatures
  runsh LJ 5^3 cells (500 atoms), 32x32 grid: the reference's own production setting (run.sh:1,7,10), whole grid on one GPU
One GPU: the preset as it stands.  --gpus N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): BOTH scaling legs
run back to back — `value` is the STRONG-scaling rate of the preset's own grid (BASELINE.json's metric: "8x8 PxT grid, 1/2/4/8 GPUs";
C2: whole pressure rows per rank while N <= 8, else even slot ranges with the split-row exchange over RCCL,
neuralmelting_amd/exchange.py), and the WEAK-scaling rate of the grid with N times as many pressure rows (every rank holds `rows`
rows, per-GPU work fixed; the exchange never leaves a pressure row, so there is no data-path collective) is reported under the key
`weak`; each leg carries its own replicas_total and world_size_seen_by_backend.  --scaling weak|strong runs one leg only.
--record adds a timed region with the reference's outputs on (thermo rows and trajectory frames written every cycle).
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
FP64_VEC_PEAK_TF = 78.6    # MI355X fp64 vector peak = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (spec)
FLOP_PER_PAIR = 40.0

# name: (element, supercell, pressure rows per GPU, rows of the whole grid, temperatures, MOD, description)
CONFIGS = {
    'C2': ('LJ', 4, 8, 8, 8, 128, 'BASELINE config 2: LJ 4^3 cells (256 atoms), 8x8 PxT grid'),
    'C3': ('LJ', 6, 2, 16, 16, 128, "BASELINE config 3, one GPU's share: LJ 6^3 cells (864 atoms), 2 of the 16x16 grid's pressure rows"),
    'C4': ('Al', 4, 8, 8, 8, 128, 'BASELINE config 4: Al (Sutton-Chen EAM) 4^3 cells (256 atoms), 8x8 PxT grid'),
    'C5': ('LJ', 8, 4, 32, 32, 128, "BASELINE config 5, one GPU's share: LJ 8^3 cells (2048 atoms), 4 of the 32x32 grid's pressure rows"),
    'runsh': ('LJ', 5, 32, 32, 32, 128, "the reference's run.sh setting: LJ 5^3 cells (500 atoms), 32x32 PxT grid"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--config', type=str, default='C2', choices=sorted(CONFIGS), help='workload preset (see the module docstring)')
    ap.add_argument('--scaling', type=str, default=None, choices=('weak', 'strong'),
                    help='default: one GPU runs the preset as it stands; at --gpus N > 1 BOTH legs run — `value` is the strong-scaling rate of the '
                         "preset's own grid (the grid BASELINE's metric names), the weak-scaling rate of the N-times-larger grid is reported "
                         'under the key `weak`.  Given explicitly, only that leg runs.')
    ap.add_argument('--sz', type=int, default=None, help='override: supercell size (-ss)')
    ap.add_argument('--rows', type=int, default=None, help='override: pressure rows per GPU')
    ap.add_argument('--tn', type=int, default=None, help='override: temperatures (-tn)')
    ap.add_argument('--mod', type=int, default=None, help='override: moves per block (-sm)')
    ap.add_argument('--el', type=str, default=None, help='override: element (-e), LJ or Al')
    ap.add_argument('--iterative', action='store_true', help="the reference's default position move (no -bm): N single-atom trials")
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--equil', type=int, default=30, help='cycles since the lattice start after which the chains count as equilibrated: '
                    'after the timed window (cycles warmup .. warmup+steps) the run continues untimed up to this cycle, and until the '
                    'mean HMC acceptance is within 0.4-0.6, then times `steps` cycles again (the sustained rate = `value`); 0 = window only')
    ap.add_argument('--force-split', action='store_true', help='strong scaling: use the split-row (collective) exchange even where whole rows would do')
    ap.add_argument('--cpu-seconds', type=float, default=6.0, help='target wall time of each cpu_baseline leg')
    ap.add_argument('--record', action='store_true',
                    help="one more timed region of `steps` cycles with the reference's outputs ON (-sc 0, remcmc:983-985): every cycle's thermo row "
                         'and trajectory frame of every replica goes D2H and is appended to its .thrm / .traj file (nm_append_outputs), '
                         'overlapped with the next block as the driver does; `value` is then the recorded rate, `record` holds both rates '
                         'and the I/O share')
    ap.add_argument('--record-dir', type=str, default=None, help='where --record writes (default: a temporary directory, removed afterwards)')
    args = ap.parse_args(argv)
    if args.iterative:
        # The reference's iterative position move never undoes a rejected trial while its step size keeps growing (remcmc:522-545,
        # 733-737): from about cycle 10 on atoms overlap, energies reach 1e12-1e18 per atom and sooner or later leave the floating-point
        # range (NM_ST_NONFINITE; LAMMPS would stop the reference there).  That mode has no equilibrated regime to time: window only.
        args.equil = 0
        if (args.warmup, args.steps) == (ap.get_default('warmup'), ap.get_default('steps')):
            args.warmup, args.steps = 3, 5   # cycles 3-7: before the chains leave the floating-point range
    return args


def main(argv=None, make_engine=None):
    """make_engine: tests only (tests/_mp_bench.py hands in a stand-in with the Engine interface so that the N > 1 logic runs on CPUs
    over gloo); the product path builds neuralmelting_amd.Engine and fails loudly without libnm_hip.so / a HIP device."""
    args = parse_args(argv)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...'
                             % (args.gpus, args.gpus))
    import torch
    have_gpu = torch.cuda.is_available()
    dist = None
    backend = None
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
            if have_gpu:
                local = local % max(torch.cuda.device_count(), 1)
                torch.cuda.set_device(local)
            dist.init_process_group(backend)
    env = dict(rank=rank, world=world, local=local, dist=dist, backend=backend, have_gpu=have_gpu, make_engine=make_engine)

    # Which legs.  One GPU: the preset as it stands (`scaling` "weak" by the contract's definition: per-GPU work fixed).  N > 1 without
    # --scaling: the metric's own grid dealt out over the ranks (strong scaling: BASELINE.json's "8x8 PxT grid, 1/2/4/8 GPUs") is
    # `value`; the grid of N times as many pressure rows (weak scaling) rides along under `weak`.
    if args.scaling is not None:
        out = run_leg(args, env, args.scaling, with_cpu=not args.no_cpu)
    elif world == 1:
        out = run_leg(args, env, 'weak', with_cpu=not args.no_cpu)
    else:
        out = run_leg(args, env, 'strong', with_cpu=not args.no_cpu)
        weak = run_leg(args, env, 'weak', with_cpu=False)
        if rank == 0:
            out['weak'] = {k: weak[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'scaling', 'window', 'sustained')}
            out['weak'].update({k: weak['config'][k] for k in ('workload', 'replicas_per_gpu', 'replicas_total', 'sweeps_per_step',
                                                               'parallelism', 'world_size_seen_by_backend', 'backend')})
            out['weak']['roofline_frac'] = weak['roofline']['frac']
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    return out


def run_leg(args, env, scaling, with_cpu):
    """one workload (grid) on all ranks: warm-up, the timed window, equilibration, the timed sustained region; rank 0 returns the line"""
    import torch
    rank, world, local, dist, backend = env['rank'], env['world'], env['local'], env['dist'], env['backend']
    el, sz, rows, np_cfg, tn, mod, desc = CONFIGS[args.config]
    custom = any(v is not None for v in (args.sz, args.rows, args.tn, args.mod, args.el))
    el, sz, rows, tn, mod = args.el or el, args.sz or sz, args.rows or rows, args.tn or tn, args.mod or mod
    if custom:
        np_cfg, desc = rows, 'custom workload'

    import neuralmelting_amd as nm
    from neuralmelting_amd import lattice, exchange as X
    make_engine = env['make_engine'] or nm.Engine

    # ---- which replicas this rank holds
    split = False
    if scaling == 'weak':
        npn = rows * world
        row0, nrows = rank * rows, rows
    else:
        npn = np_cfg if not custom else rows
        if npn % world == 0 and not args.force_split:
            nrows = npn // world
            row0 = rank * nrows
        elif (npn * tn) % world == 0:  # fewer rows than ranks: even slot ranges, a pressure row spans ranks
            split = True
        else:
            raise SystemExit('strong scaling: %d x %d replicas do not divide over %d ranks' % (npn, tn, world))
    P = np.linspace(1.0, 8.0, npn, dtype=np.float32)
    T = np.linspace(0.25, 2.5, tn, dtype=np.float32) if el == 'LJ' else np.linspace(256.0, 2560.0, tn, dtype=np.float32)
    natoms = 4 * sz ** 3
    kw = dict(element=el, device=local, ppos=0.125, pvol=0.125, nstps=8, bulk=not args.iterative, seed=256)
    if split:
        nloc = npn * tn // world
        k0 = rank * nloc
        r0, r1 = k0 // tn, (k0 + nloc - 1) // tn
        x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=r0, nrows=r1 - r0 + 1)
        a = k0 - r0 * tn
        x, v, box, d = x[a:a + nloc], v[a:a + nloc], box[a:a + nloc], d[a:a + nloc]
        eng = make_engine(natoms, P, T, slot0=k0, nslots=nloc, **kw)
    else:
        x, v, box, d = lattice.init_states(sz, P, T, 0.03125, 0.03125, el=el, row0=row0, nrows=nrows)
        eng = make_engine(natoms, P, T, row0=row0, nrows=nrows, **kw)
        k0 = row0 * tn
    eng.set_state(x, v, box, d)
    ns = eng.nslots
    if split:
        from neuralmelting_amd import remcmc
        et_all = np.array([remcmc.init_constant(P, T, el, *divmod(k, tn))[0] for k in range(npn * tn)])
        pf_all = np.array([remcmc.init_constant(P, T, el, *divmod(k, tn))[1] for k in range(npn * tn)])
        info = (world if dist is not None else 1, backend == 'nccl')

    def exchange_split(step):
        """a pressure row spans ranks: all-gather (E_tot, V), identical sweep everywhere, the swapped replicas move (exchange.py)"""
        X.exchange_split(eng, step, npn, tn, 256, k0, natoms, et_all, pf_all, info, rank=rank)

    recorder = None

    def cycle(step):
        eng.set_step(step)
        eng.run_block(mod)
        if recorder is not None:
            recorder.cycle(eng)          # this cycle's outputs are snapshotted behind the block (in front of gen_mc_params, which zeroes the counters)
        eng.adapt()
        if split:
            exchange_split(step)
        else:
            eng.exchange(count=False)
        if recorder is not None:
            recorder.drain(eng)          # the PREVIOUS cycle's snapshot is fetched and handed to the writer thread while this block runs

    def fence():
        eng.synchronize()
        if env['have_gpu']:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if env['have_gpu']:
                torch.cuda.synchronize()

    def reduce_(vals, op):
        tt = torch.tensor(vals, dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=op)
        return [float(q) for q in tt.tolist()]

    def timed(nsteps, step):
        """exactly nsteps cycles between two fences; (seconds = max over ranks, replicas of all ranks, kernel launches, kernel ms, stats)"""
        fence()
        eng.timing_reset()
        eng.stats(reset=True)
        heals0 = getattr(eng, 'heals', 0)
        t0 = time.perf_counter()
        if recorder is None and not split and hasattr(eng, 'run_cycles') and not os.environ.get('NM_BENCH_SINGLE_CALLS'):
            # outputs off: the K cycles as ONE call (nm_run_cycles: where the grid runs as clusters, one launch in which only the replicas of a
            # pressure row wait for one another; the same chains, bit for bit, as the loop of single calls below)
            eng.set_step(step)
            eng.run_cycles(nsteps, mod)
            step += nsteps
        else:
            for _ in range(nsteps):
                cycle(step)
                step += 1
        if recorder is not None:
            recorder.flush()
        fence()
        dt = time.perf_counter() - t0
        ns_total = ns
        if dist is not None:
            dt = reduce_([dt], dist.ReduceOp.MAX)[0]
            ns_total = int(round(reduce_([float(ns)], dist.ReduceOp.SUM)[0]))
        launches, kms = eng.timing()
        return step, dt, ns_total, launches, kms, eng.stats(), getattr(eng, 'heals', 0) - heals0

    def hmc_acceptance(step):
        """mean HMC acceptance over this rank's replicas of one more (untimed) cycle; agreed over the ranks"""
        eng.set_step(step)
        eng.run_block(mod)
        a = float(eng.thermo()[:, 16].mean())
        eng.adapt()
        if split:
            exchange_split(step)
        else:
            eng.exchange(count=False)
        if dist is not None:
            a = reduce_([a], dist.ReduceOp.SUM)[0] / dist.get_world_size()
        return a

    step = 0
    for _ in range(args.warmup):
        cycle(step)
        step += 1
    # ---- the window right after the warm-up the caller asked for (what rounds 1 and 2 reported)
    step, dt_w, ns_total, launches_w, kms_w, st_w, heals_w = timed(args.steps, step)
    # ---- untimed equilibration, then the same number of timed cycles: the sustained rate
    equil = 0
    if args.equil > 0:
        acc = None
        while step < args.equil or acc is None or not (0.4 <= acc <= 0.6):
            if step >= args.equil + 40:   # adaptation oscillates around 0.5: do not wait for ever
                break
            acc = hmc_acceptance(step)
            step += 1
            equil += 1
        step, dt, ns_total, launches, kms, st, heals_s = timed(args.steps, step)
    else:
        dt, launches, kms, st, heals_s = dt_w, launches_w, kms_w, st_w, heals_w
    # ---- the same number of cycles once more with the reference's outputs ON (--record)
    rec = None
    if args.record:
        import shutil
        import tempfile
        rdir = args.record_dir or tempfile.mkdtemp(prefix='nm_record_')
        os.makedirs(rdir, exist_ok=True)
        recorder = Recorder(rdir, k0, ns, natoms)
        for _ in range(4):        # warm-up of the recording path, untimed like every other warm-up: the first snapshots allocate the side stream and the
            cycle(step)           # device / pinned host buffers, and the block kernel itself runs 2-3 % slower for the first few cycles that touch them
            step += 1             # (scripts/probe_record.py: 5.85 ms per block in the first ten recorded cycles of a context, 5.66-5.71 ms in every
        recorder.flush()          # later region, recording or not)
        recorder.d2h_s = recorder.write_s = 0.0
        recorder.bytes = 0
        step, dt_r, _, launches_r, kms_r, st_r, _ = timed(args.steps, step)
        rec = recorder.summary(args.steps)
        recorder = None
        if args.record_dir is None:
            shutil.rmtree(rdir, ignore_errors=True)
    world_seen = dist.get_world_size() if dist is not None else 1

    # one more block outside the timed region, read before gen_mc_params zeroes the counters: the per-replica acceptance
    # ratios and U, V the metric's definition asks to see next to the rate
    eng.set_step(step)
    eng.run_block(mod)
    last = eng.thermo()

    out = None
    if rank == 0:
        # dominant kernel: nm_block_kernel, one launch per step, ns*mod sweeps per launch
        phmc = 1.0 - 0.125 - 0.125
        evals_alg = 0.125 + 0.125 + phmc * 9.0                             # EVALS = PPOS + PVOL + PHMC (NSTPS + 1) = 7.0
        bytes_per_sweep = 48.0 * natoms + 24.0 * natoms * phmc             # SURVEY.md §8d compulsory HBM bytes
        sweeps_per_launch = ns * mod

        def numbers(dt_, launches_, kms_, st_):
            """rate, kernel time and roofline figures of one timed region of args.steps cycles"""
            k_avg_s = max((kms_ / max(launches_, 1)) * 1e-3, 1e-12)
            evals = st_[:, 0].sum()
            mean_pairs = st_[:, 3].sum() / max(st_[:, 2].sum(), 1.0)
            cycles_per_launch = args.steps / max(launches_, 1)             # 1 with single calls, `steps` when nm_run_cycles made one launch of the region
            sweeps_per_launch = ns * mod * cycles_per_launch
            flop_alg_launch = evals_alg * sweeps_per_launch * mean_pairs * FLOP_PER_PAIR
            blk_ms = st_[:, 4] / np.maximum(st_[:, 6], 1.0) * 1e-5          # per slot: mean time of its blocks (100 MHz ticks -> ms)
            return {'value': ns_total * mod * args.steps / dt_, 'ms_per_step': dt_ / args.steps * 1e3,
                    'kernel_avg_ms': k_avg_s * 1e3, 'launches': launches_, 'cycles_per_launch': cycles_per_launch,
                    'algorithmic_bytes_per_launch': bytes_per_sweep * sweeps_per_launch,
                    'achieved': flop_alg_launch / k_avg_s / 1e12, 'frac': flop_alg_launch / k_avg_s / 1e12 / FP64_VEC_PEAK_TF,
                    'algorithmic_flop_per_launch': flop_alg_launch,
                    'executed': evals * mean_pairs * FLOP_PER_PAIR / max(kms_ * 1e-3, 1e-12) / 1e12,
                    'frac_executed': evals * mean_pairs * FLOP_PER_PAIR / max(kms_ * 1e-3, 1e-12) / 1e12 / FP64_VEC_PEAK_TF,
                    'evals_per_sweep': evals / (ns * mod * args.steps), 'mean_pairs_per_eval': mean_pairs,
                    'list_rebuilds_per_sweep': st_[:, 1].sum() / (ns * mod * args.steps),
                    'hbm_gbs_algorithmic': bytes_per_sweep * sweeps_per_launch / k_avg_s / 1e9,
                    # a launch lasts as long as its slowest replica: mean and maximum over the slots of their mean block time
                    'slot_block_ms_mean': float(blk_ms.mean()), 'slot_block_ms_max': float(blk_ms.max()),
                    'clusters_on_one_xcd': float(st_[:, 5].sum() / max(st_[:, 6].sum(), 1.0)) if eng.cus_per_replica > 1 else None}

        win = numbers(dt_w, launches_w, kms_w, st_w)
        sus = numbers(dt, launches, kms, st) if args.equil > 0 else win
        prof = measured_profile((args.config + ('_iter' if args.iterative else '')) if not custom else None, ns, mod, sus['cycles_per_launch'])
        grid = '%dx%d PxT grid%s' % (npn, tn, '' if world == 1 else ' over %d GPUs' % world)
        metric = 'MC sweeps/sec (whole node), %s %d^3 cells, %s' % (el, sz, grid)
        keys = ('value', 'ms_per_step', 'kernel_avg_ms', 'cycles_per_launch', 'frac', 'frac_executed', 'evals_per_sweep', 'list_rebuilds_per_sweep',
                'slot_block_ms_mean', 'slot_block_ms_max')
        out = {
            # value = the SUSTAINED rate: `steps` timed cycles of equilibrated chains (HMC accepting about half its trajectories, step
            # sizes adapted).  `window` = the same number of cycles timed right after `warmup` cycles from the lattice start, where
            # trajectories are still short and lists are rebuilt less often: what rounds 1 and 2 reported as value.
            'metric': metric, 'value': sus['value'], 'unit': 'sweeps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': sus['ms_per_step'], 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'equilibration_cycles': equil, 'timed_from_cycle': step - args.steps * (2 if args.record else 1) - (4 if args.record else 0) if args.equil > 0 else args.warmup,
            'window': {k: win[k] for k in keys},
            'sustained': {k: sus[k] for k in keys},
            'config': {'workload': '%s; %s %d^3 cells (%d atoms), %d replicas on rank 0 (%d in all), MOD=%d, %s PMC 0.125 / VMC '
                                   '0.125 / HMC 0.75 x 8 steps, outputs off' % (desc, el, sz, natoms, ns, ns_total, mod,
                                                                               'iterative' if args.iterative else 'bulk'),
                       'preset': args.config if not custom else None, 'replicas_per_gpu': ns, 'replicas_total': ns_total,
                       'sweeps_per_step': ns_total * mod,
                       'parallelism': ('rows/gpu' if not split else 'slots/gpu, split-row exchange over %s' % backend),
                       'world_size_seen_by_backend': world_seen, 'backend': backend},
            # The binding roofline: the working set is LDS-resident, so the ceiling is fp64 vector issue.
            # achieved = ALGORITHMIC flops per launch (7.0 evaluations per sweep x measured interacting pairs x 40 flop) / the
            # kernel's HIP-event time over the timed (sustained) region; executed = what the kernel actually evaluated (it keeps forces
            # across moves: ~6.26 evaluations per sweep); traffic = HBM bytes per launch from the committed rocprofv3 PMC passes.
            'roofline': {'bound': 'fp64-valu', 'achieved': sus['achieved'], 'peak': FP64_VEC_PEAK_TF, 'unit': 'TFLOP/s',
                         'frac': sus['frac'], 'traffic': prof.get('traffic'), 'traffic_range': prof.get('traffic_range'),
                         'kernel': 'nm_cycles_kernel' if sus['cycles_per_launch'] > 1 else 'nm_block_kernel', 'kernel_avg_ms': sus['kernel_avg_ms'],
                         'launches': sus['launches'], 'cycles_per_launch': sus['cycles_per_launch'],   # a timed region is ONE launch of nm_cycles_kernel where nm_run_cycles fuses it
                         'kernel_ms_per_cycle': sus['kernel_avg_ms'] / sus['cycles_per_launch'],
                         'launches_that_did_no_work': heals_s,   # blocks re-issued at fewer workgroups per replica inside the region (not in kernel_avg_ms)
                         'algorithmic_flop_per_launch': sus['algorithmic_flop_per_launch'], 'algorithmic_evals_per_sweep': evals_alg,
                         'executed': sus['executed'], 'frac_executed': sus['frac_executed'], 'evals_per_sweep': sus['evals_per_sweep'],
                         'mean_pairs_per_eval': sus['mean_pairs_per_eval'], 'flop_per_pair': FLOP_PER_PAIR,
                         'list_rebuilds_per_sweep': sus['list_rebuilds_per_sweep'],
                         'valu_active_share': prof.get('valu_active_share'), 'profile': prof.get('source'),
                         'profile_commit': prof.get('commit'), 'clusters_on_one_xcd': sus['clusters_on_one_xcd'],
                         'cus_per_replica': eng.cus_per_replica, 'cus_occupied': min(ns * eng.cus_per_replica, 256), 'cus_total': 256},
            # the HBM line the contract asks for: algorithmic bytes (48 N + 24 N PHMC per sweep) / kernel time, << 1 % by design
            'roofline_hbm': {'bound': 'hbm', 'achieved': sus['hbm_gbs_algorithmic'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': sus['hbm_gbs_algorithmic'] / HBM_PEAK_GBS, 'traffic': prof.get('traffic'),
                             'algorithmic_bytes_per_launch': sus['algorithmic_bytes_per_launch']},
        }
        if rec is not None:
            on = ns_total * mod * args.steps / dt_r
            out['record'] = dict(rec, outputs_off=sus['value'], outputs_on=on, io_share=1.0 - on / sus['value'],
                                 ms_per_step_on=dt_r / args.steps * 1e3, kernel_avg_ms_on=kms_r / max(launches_r, 1), launches_on=launches_r,
                                 slot_block_ms_mean_on=float((st_r[:, 4] / np.maximum(st_r[:, 6], 1.0) * 1e-5).mean()),
                                 slot_block_ms_max_on=float((st_r[:, 4] / np.maximum(st_r[:, 6], 1.0) * 1e-5).max()),
                                 list_rebuilds_per_sweep_on=float(st_r[:, 1].sum() / (ns * mod * args.steps)),
                                 note='rank 0; every cycle: thermo rows + positions snapshotted behind the block (nm_snapshot: device copy, D2H on a side stream), '
                                      'fetched one cycle later; 17-column .thrm row and N+1-line .traj frame per replica appended by nm_append_outputs '
                                      '(remcmc:235-256) on a helper thread while the next block runs')
            out['value'], out['ms_per_step'] = on, dt_r / args.steps * 1e3
            out['metric'] = metric + ' (outputs on)'
            out['config']['workload'] = out['config']['workload'].replace('outputs off', 'outputs ON every cycle (-sc 0)')
        out['replicas'] = {'note': 'rank 0, block after the timed region, slot k = i*NT + j (pressure i, temperature j)',
                           'accept_pmc': [round(float(a), 3) for a in last[:, 14]],
                           'accept_vmc': [round(float(a), 3) for a in last[:, 15]],
                           'accept_hmc': [round(float(a), 3) for a in last[:, 16]],
                           'pe_per_atom': [round(float(a) / natoms, 4) for a in last[:, 1]],
                           'vol_per_atom': [round(float(a) / natoms, 4) for a in last[:, 4]]}
        if with_cpu:
            out['cpu_baseline'] = cpu_baseline(eng, natoms, el, mod, T, tn, k0, args.cpu_seconds, bulk=not args.iterative)
    if dist is not None:
        dist.barrier()
    eng.close()
    return out


class Recorder:
    """write_outputs (remcmc:259-286) for the bench's --record leg: per cycle the thermo rows and the positions of this rank's replicas go
    D2H and one .thrm row + one .traj frame per replica are appended (nm_append_outputs, threaded C formatter) on a helper thread while
    the GPU runs the next block — what neuralmelting_amd/remcmc.py's main loop does with -sc 0"""

    def __init__(self, rdir, k0, ns, natoms):
        import ctypes as C
        from neuralmelting_amd import _lib as B
        self.B, self.C, self.ns, self.natoms = B, C, ns, natoms
        self.thrm = (C.c_char_p * ns)(*[os.path.join(rdir, 'bench.%04d.thrm' % (k0 + k)).encode() for k in range(ns)])
        self.traj = (C.c_char_p * ns)(*[os.path.join(rdir, 'bench.%04d.traj' % (k0 + k)).encode() for k in range(ns)])
        self.snaps, self.eng, self.thread, self.err = 0, None, None, None
        self.d2h_s = self.write_s = 0.0
        self.bytes = 0

    def _write(self, rows, x, box):
        t0 = time.perf_counter()
        if os.environ.get('NM_BENCH_NO_WRITE'):   # dev: isolate the cost of the copies from that of formatting and files
            return
        B = self.B
        rc = B.load().nm_append_outputs(self.ns, self.natoms, self.thrm, self.traj, rows.ctypes.data_as(B.c_double_p),
                                        x.ctypes.data_as(B.c_double_p), box.ctypes.data_as(B.c_double_p), 8)
        if rc != 0:
            self.err = IOError('nm_append_outputs failed (%d)' % rc)
        self.write_s += time.perf_counter() - t0

    def _join(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
            if self.err is not None:
                raise self.err

    def cycle(self, eng):
        """called right after run_block was enqueued: this cycle's outputs are snapshotted behind the block (Engine.snapshot: the copy to the
        host runs beside the stream)"""
        eng.snapshot()
        self.snaps += 1
        self.eng = eng
        self.bytes += self.ns * (17 * 11 + 1 + (self.natoms + 1) * 34)   # ' %.4E' = 11 bytes per number

    def drain(self, eng):
        """called once the cycle is queued: the previous cycle's snapshot is fetched (it waits for that copy only) and handed to the writer thread"""
        import threading
        if self.snaps > 1:
            t0 = time.perf_counter()
            rows, xs, boxs = eng.snapshot_fetch()
            self.snaps -= 1
            self.d2h_s += time.perf_counter() - t0
            self._join()
            self.thread = threading.Thread(target=self._write, args=(np.ascontiguousarray(rows), np.ascontiguousarray(xs), np.ascontiguousarray(boxs)))
            self.thread.start()

    def flush(self):
        self._join()
        while self.snaps:
            t0 = time.perf_counter()
            rows, xs, boxs = self.eng.snapshot_fetch()
            self.snaps -= 1
            self.d2h_s += time.perf_counter() - t0
            self._write(np.ascontiguousarray(rows), np.ascontiguousarray(xs), np.ascontiguousarray(boxs))

    def summary(self, steps):
        return {'write_ms_per_step': self.write_s / steps * 1e3, 'fetch_ms_per_step': self.d2h_s / steps * 1e3,
                'text_bytes_per_step': self.bytes // max(steps, 1)}


def measured_profile(config, ns, mod, cycles_per_launch=1.0):
    """What the committed rocprofv3 PMC passes of this round say about nm_block_kernel on the preset's workload (profiles/,
    written by scripts/collect_pmc.py together with the commit they were taken at): HBM bytes per launch — FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950, KB units — and the share of SIMD cycles that issued VALU.  Empty for
    workloads without a committed profile.  "Per launch" follows the line's own launch: the profile's bytes per cycle (its launch held
    `_meta.cycles_per_launch` cycles) times the cycles one launch of THIS run held."""
    if config is None:
        return {}
    f = None
    for rnd in ('r04', 'r03', 'r02'):   # this round's passes; an older round's file is still labelled with the commit it belongs to
        g = os.path.join(ROOT, 'profiles', '%s_pmc_block_kernel_%s.json' % (rnd, config))
        if os.path.isfile(g):
            f = g
            break
    if f is None:
        return {}
    d = json.load(open(f))
    meta = d.get('_meta', {})
    if meta.get('replicas') != ns or meta.get('mod') != mod:
        return {}
    out = {'source': 'profiles/' + os.path.basename(f), 'commit': meta.get('commit'), 'kernel': meta.get('kernel')}
    scale = cycles_per_launch / float(meta.get('cycles_per_launch') or 1)
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        out['traffic'] = (2.0 * d['FETCH_SIZE']['mean'] + d['WRITE_SIZE']['mean']) * 1024.0 * scale
        # min - max over the profiled launches and over the round's other profile runs of this preset (the cluster kernels' write-back of
        # hand-over lines moves the figure by up to 2x from run to run with unchanged kernels: quote a range, not a point)
        lo = (2.0 * d['FETCH_SIZE']['min'] + d['WRITE_SIZE']['min']) * 1024.0 * scale
        hi = (2.0 * d['FETCH_SIZE']['max'] + d['WRITE_SIZE']['max']) * 1024.0 * scale
        for extra in meta.get('traffic_other_runs', []):          # recorded per cycle
            lo, hi = min(lo, extra * cycles_per_launch), max(hi, extra * cycles_per_launch)
        out['traffic_range'] = [lo, hi]
    if 'SQ_ACTIVE_INST_VALU' in d and 'GRBM_GUI_ACTIVE' in d:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = d['GRBM_GUI_ACTIVE']['mean'] / 8.0
        out['valu_active_share'] = 4.0 * d['SQ_ACTIVE_INST_VALU']['mean'] / (256 * 4 * cyc)
    return out


def cpu_baseline(eng, natoms, el, mod, T, tn, k0, seconds, bulk=True):
    """the oracle (C restatement of the same path, kind "port": the reference's own CPU path needs a LAMMPS build that is neither
    in its tree nor in this image) timed on the host cores on the engine's post-warm-up states: (i) ONE thread on a bounded
    sample of the replicas, (ii) OpenMP over replicas on all cores, each for about `seconds` of wall time"""
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
    kw = dict(units=1, mass=lattice.MASS['Al'], pot=1) if el == 'Al' else {}
    tq = np.tile(T.astype(np.float64), (ns + tn - 1) // tn + 1)[(k0 % tn):(k0 % tn) + ns]
    common = dict(natoms=natoms, nstps=8, bulk=bulk, ppos=0.125, pvol=0.125, lat=lattice.LAT[el][1], seed=256, **kw)

    def leg(sel, nthreads, mod_leg, budget):
        xs, vs, bs = x[sel].copy(), v[sel].copy(), box[sel].copy()
        n, t0, cyc = len(bs), time.perf_counter(), 0
        while True:
            o = O.run_blocks(xs, vs, bs, d[sel], tq[sel], et[sel], pf[sel], mod=mod_leg, slot0=k0 + int(sel[0]), step=1000 + cyc,
                             nthreads=nthreads, **common)
            xs, vs, bs = o['x'], o['v'], o['box']
            cyc += 1
            el_ = time.perf_counter() - t0
            if el_ >= budget or cyc >= 64:
                return n * mod_leg * cyc / el_, el_, cyc
    # (i) one thread: a spread of replicas across the grid (cold solid ... hot fluid), blocks short enough to fit the budget
    sel1 = np.unique(np.linspace(0, ns - 1, min(ns, 8)).astype(int))
    mod1 = max(8, min(mod, 32))
    r1, t1, c1 = leg(sel1, 1, mod1, seconds)
    # (ii) all cores, all replicas
    ra, ta, ca = leg(np.arange(ns), cores, mod, seconds)
    return {'value': ra, 'unit': 'sweeps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d cycles of the same %d-replica workload (MOD=%d) from the GPU run\'s post-warm-up states, OpenMP over '
                      'replicas on %d threads, %.1f s wall' % (ca, ns, mod, cores, ta),
            'single_thread': {'value': r1, 'unit': 'sweeps/s', 'cores': 1,
                              'sample': '%d blocks of %d moves on %d replicas spread over the grid, one thread, %.1f s wall'
                                        % (c1, mod1, len(sel1), t1)}}


if __name__ == '__main__':
    main()
