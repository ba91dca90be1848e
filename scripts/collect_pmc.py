"""summarise rocprofv3 --pmc passes (counter_collection csv files under the given directories) into one JSON:
per counter the mean / min / max over the launches of the kernel whose name contains KERNEL (default nm_block_kernel).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 2 --no-cpu
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS \\
              SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py ...
    python scripts/collect_pmc.py out.json gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
(separate passes, no tracing flags together with --pmc)"""
import csv, glob, json, os, sys
from collections import defaultdict


def main(out, dirs, kernel='nm_block_kernel'):
    per = defaultdict(lambda: defaultdict(float))           # counter -> dispatch -> value (summed over XCDs / instances)
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row['Kernel_Name']:
                    per[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
    res = {c: {'launches': len(v), 'mean': sum(v.values()) / len(v), 'min': min(v.values()), 'max': max(v.values())}
           for c, v in sorted(per.items())}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
