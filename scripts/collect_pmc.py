"""summarise rocprofv3 --pmc passes (counter_collection csv files under the given directories) into one JSON:
per counter the mean / min / max over the launches of the kernel whose name contains KERNEL, plus `_meta` = {commit, config, replicas, mod,
kernel, cycles_per_launch} so that bench.py can tell which build, workload and launch shape the numbers belong to.  KERNEL `auto` (default):
nm_cycles_kernel where the profiled bench made its timed region one launch of --cycles cycles (nm_run_cycles; every such launch is kept),
else nm_block_kernel, one cycle per launch, --skip / --take choosing the timed ones.

    python scripts/collect_pmc.py out.json --config C2 --replicas 64 --mod 128 --commit $(git rev-parse --short HEAD) DIR [DIR ...]
(the passes themselves: scripts/profile_round.sh — separate runs per counter group, no tracing flags together with --pmc)"""
import argparse, csv, glob, json, os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('dirs', nargs='+')
    ap.add_argument('--kernel', default='auto')
    ap.add_argument('--cycles', type=int, default=10, help='cycles per launch of nm_cycles_kernel (the bench\'s --steps)')
    ap.add_argument('--config', default=None)
    ap.add_argument('--replicas', type=int, default=None)
    ap.add_argument('--mod', type=int, default=None)
    ap.add_argument('--commit', default=None)
    ap.add_argument('--skip', type=int, default=0, help='launches to drop from the front (warm-up)')
    ap.add_argument('--take', type=int, default=0, help='launches to keep after the skipped ones (0 = all)')
    ap.add_argument('--regime', default=None, help="what the kept launches are, e.g. 'equilibrated: cycles 30-39 after the lattice start'")
    a = ap.parse_args()
    per = defaultdict(lambda: defaultdict(float))           # counter -> dispatch -> value (summed over XCDs / instances)
    files = [f for d in a.dirs for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)]
    cycles = 1
    if a.kernel == 'auto':
        fused = any('nm_cycles_kernel' in row['Kernel_Name'] for f in files for row in csv.DictReader(open(f)))
        a.kernel = 'nm_cycles_kernel' if fused else 'nm_block_kernel'
        if fused:
            a.skip, a.take, cycles = 0, 0, a.cycles
    elif 'nm_cycles_kernel' in a.kernel:
        cycles = a.cycles
    for f in files:
        if True:
            for row in csv.DictReader(open(f)):
                if a.kernel in row['Kernel_Name']:
                    per[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
    res = {}
    for c, v in sorted(per.items()):
        vals = [v[k] for k in sorted(v)][a.skip:]
        if a.take:
            vals = vals[:a.take]
        res[c] = {'launches': len(vals), 'mean': sum(vals) / len(vals), 'min': min(vals), 'max': max(vals)}
    res['_meta'] = {'commit': a.commit, 'config': a.config, 'replicas': a.replicas, 'mod': a.mod, 'kernel': a.kernel, 'cycles_per_launch': cycles, 'regime': a.regime,
                    'launches_skipped': a.skip}
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
