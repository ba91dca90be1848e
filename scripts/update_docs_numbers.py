"""dev helper: print the numbers of gpurun_out/final_<round>/bench_*.json and profiles/<round>_pmc_* in the layout of the tables of DESIGN.md §7 /
BASELINE.md (the tables themselves are edited by hand).   python scripts/update_docs_numbers.py [r04]"""
import csv, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 else 'r04'
D = {}
for f in ['C2', 'driver', 'C3', 'C4', 'C5', 'C5x2', 'runsh', 'C1', 'C2_iter', 'C2_rows4', 'C2_rows2', 'C2_rows1', 'C2_record', 'C5_record']:
    g = 'gpurun_out/final_%s/bench_%s.json' % (RND, f)
    if os.path.isfile(g):
        D[f] = json.load(open(g))
for f, d in D.items():
    cb = d.get('cpu_baseline') or {}
    r = d['roofline']
    print('%-9s sustained %9.0f (%.2f ms) window %9.0f (%.2f ms) frac %.4f / exec %.4f / window %.4f  %.2f TF  alg %.1f GB/s  rb %.2f  cpu %s / %s  x%s  Q=%d prof %s'
          % (f, d['value'], r['kernel_avg_ms'], d['window']['value'], d['window']['kernel_avg_ms'], r['frac'], r['frac_executed'], d['window']['frac'], r['achieved'],
             (d.get('roofline_hbm') or {}).get('achieved', 0), r['list_rebuilds_per_sweep'], '%.0f' % cb['value'] if cb else '-',
             '%.0f' % cb['single_thread']['value'] if cb else '-', '%.0f' % (d['value'] / cb['value']) if cb else '-', r['cus_per_replica'], r.get('profile_commit')))
for t in ['C2', 'C3', 'C4', 'C5', 'runsh']:
    d = json.load(open('profiles/%s_pmc_block_kernel_%s.json' % (RND, t)))
    tr = (2 * d['FETCH_SIZE']['mean'] + d['WRITE_SIZE']['mean']) * 1024
    cyc = d['GRBM_GUI_ACTIVE']['mean'] / 8
    r = list(csv.reader(open('profiles/%s_kernel_stats_%s.csv' % (RND, t))))
    print('%-6s fetch %.1f MB write %.1f MB traffic %.3f GB  VALU %.3f  conflicts / LDS issue %.2f, / CU cycles %.3f  wait %.2f | all launches %.2f ms | %s'
          % (t, d['FETCH_SIZE']['mean'] / 1024, d['WRITE_SIZE']['mean'] / 1024, tr / 1e9, 4 * d['SQ_ACTIVE_INST_VALU']['mean'] / (256 * 4 * cyc),
             d['SQ_LDS_BANK_CONFLICT']['mean'] / d['SQ_ACTIVE_INST_LDS']['mean'], d['SQ_LDS_BANK_CONFLICT']['mean'] / (256 * cyc),
             d['SQ_WAIT_ANY']['mean'] / d['SQ_WAVE_CYCLES']['mean'], float(r[1][3]) / 1e6, d['_meta']['commit']))
for f in ('C2_record', 'C5_record'):
    if f in D:
        r = D[f]['record']
        print('%-10s outputs off %9.0f on %9.0f  io share %.3f  write %.2f ms/step  text %.0f KB/step' % (f, r['outputs_off'], r['outputs_on'], r['io_share'], r['write_ms_per_step'], r['text_bytes_per_step'] / 1e3))
