"""summarise rocprofv3 --pmc passes (counter_collection csv files under the given directories) into one JSON:
per counter the mean / min / max over the launches of the kernel whose name contains KERNEL (default nm_block_kernel), plus
`_meta` = {commit, config, replicas, mod} so that bench.py can tell which build and workload the numbers belong to.

    python scripts/collect_pmc.py out.json --config C2 --replicas 64 --mod 128 --commit $(git rev-parse --short HEAD) DIR [DIR ...]
(the passes themselves: scripts/profile_round.sh — separate runs per counter group, no tracing flags together with --pmc)"""
import argparse, csv, glob, json, os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('dirs', nargs='+')
    ap.add_argument('--kernel', default='nm_block_kernel')
    ap.add_argument('--config', default=None)
    ap.add_argument('--replicas', type=int, default=None)
    ap.add_argument('--mod', type=int, default=None)
    ap.add_argument('--commit', default=None)
    ap.add_argument('--skip', type=int, default=0, help='launches to drop from the front (warm-up)')
    ap.add_argument('--take', type=int, default=0, help='launches to keep after the skipped ones (0 = all)')
    ap.add_argument('--regime', default=None, help="what the kept launches are, e.g. 'equilibrated: cycles 30-39 after the lattice start'")
    a = ap.parse_args()
    per = defaultdict(lambda: defaultdict(float))           # counter -> dispatch -> value (summed over XCDs / instances)
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                if a.kernel in row['Kernel_Name']:
                    per[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
    res = {}
    for c, v in sorted(per.items()):
        vals = [v[k] for k in sorted(v)][a.skip:]
        if a.take:
            vals = vals[:a.take]
        res[c] = {'launches': len(vals), 'mean': sum(vals) / len(vals), 'min': min(vals), 'max': max(vals)}
    res['_meta'] = {'commit': a.commit, 'config': a.config, 'replicas': a.replicas, 'mod': a.mod, 'kernel': a.kernel, 'regime': a.regime,
                    'launches_skipped': a.skip}
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
