"""One number per quantity: the tables of DESIGN.md §7.1 and BASELINE.md §5 are GENERATED from the bench lines of gpurun_out/final_<round>/ and the
PMC summaries of profiles/<round>_pmc_* (scripts/evidence.sh writes both).

    python scripts/update_docs_numbers.py [r04]            print the numbers
    python scripts/update_docs_numbers.py r04 --write      regenerate the tables between the <!-- numbers:... --> markers of both files
"""
import csv, json, os, sys
RND = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'r04'
WRITE = '--write' in sys.argv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
def kpc(o):
    """kernel ms per CYCLE: a launch of nm_cycles_kernel holds `cycles_per_launch` of them"""
    return o['kernel_avg_ms'] / (o.get('cycles_per_launch') or 1)


D = {}
for f in ['C2', 'driver', 'C3', 'C4', 'C5', 'C5x2', 'runsh', 'C1', 'C2_iter', 'C2_rows4', 'C2_rows2', 'C2_rows1', 'C2_record', 'C5_record']:
    g = 'gpurun_out/final_%s/bench_%s.json' % (RND, f)
    if os.path.isfile(g):
        D[f] = json.load(open(g))
PM = {}
for t in ['C2', 'C3', 'C4', 'C5', 'runsh']:
    d = json.load(open('profiles/%s_pmc_block_kernel_%s.json' % (RND, t)))
    cyc = d['GRBM_GUI_ACTIVE']['mean'] / 8
    kname = d['_meta'].get('kernel') or 'nm_block_kernel'
    cpl = d['_meta'].get('cycles_per_launch') or 1          # everything below is per CYCLE (one block of MOD moves per replica)
    r = [x for x in csv.reader(open('profiles/%s_kernel_stats_%s.csv' % (RND, t))) if x and kname in x[0]] or [[kname, '0', '0', '0']]
    r = [None] + r
    PM[t] = dict(traffic=(2 * d['FETCH_SIZE']['mean'] + d['WRITE_SIZE']['mean']) * 1024 / cpl, valu=4 * d['SQ_ACTIVE_INST_VALU']['mean'] / (256 * 4 * cyc),
                 conf=d['SQ_LDS_BANK_CONFLICT']['mean'] / d['SQ_ACTIVE_INST_LDS']['mean'], wait=d['SQ_WAIT_ANY']['mean'] / d['SQ_WAVE_CYCLES']['mean'],
                 all_ms=float(r[1][3]) / 1e6 / cpl, kernel=kname, commit=d['_meta']['commit'], other=d['_meta'].get('traffic_other_runs', []),
                 fetch=d['FETCH_SIZE']['mean'] / 1024 / cpl, write=d['WRITE_SIZE']['mean'] / 1024 / cpl)

if not WRITE:
    for f, d in D.items():
        cb = d.get('cpu_baseline') or {}
        r = d['roofline']
        print('%-9s sustained %9.0f (%.2f ms) window %9.0f (%.2f ms) frac %.4f / exec %.4f / window %.4f  %.2f TF  alg %.1f GB/s  rb %.2f  cpu %s / %s  x%s  Q=%d prof %s'
              % (f, d['value'], kpc(r), d['window']['value'], kpc(d['window']), r['frac'], r['frac_executed'], d['window']['frac'], r['achieved'],
                 (d.get('roofline_hbm') or {}).get('achieved', 0), r['list_rebuilds_per_sweep'], '%.0f' % cb['value'] if cb else '-',
                 '%.0f' % cb['single_thread']['value'] if cb else '-', '%.0f' % (d['value'] / cb['value']) if cb else '-', r['cus_per_replica'], r.get('profile_commit')))
    for t, m in PM.items():
        print('%-6s fetch %.1f MB write %.1f MB traffic %.3f GB  VALU %.3f  conflicts / LDS issue %.2f  wait %.2f | all launches %.2f ms | %s'
              % (t, m['fetch'], m['write'], m['traffic'] / 1e9, m['valu'], m['conf'], m['wait'], m['all_ms'], m['commit']))
    for f in ('C2_record', 'C5_record'):
        if f in D:
            r = D[f]['record']
            print('%-10s outputs off %9.0f on %9.0f  io share %.3f  write %.2f ms/step  text %.0f KB/step' % (f, r['outputs_off'], r['outputs_on'], r['io_share'], r['write_ms_per_step'], r['text_bytes_per_step'] / 1e3))
    sys.exit(0)


def rate(v):
    return '%.3f M' % (v / 1e6) if v >= 1e6 else '%.1f k' % (v / 1e3) if v < 2e5 and v >= 1e5 else '%.0f k' % (v / 1e3) if v >= 1e4 else '%.0f' % v


def ks(v):
    return '%.1f k' % (v / 1e3) if v >= 1e3 else '%.0f' % v


R3 = {'C2': '1.395 M', 'driver': '1.393 M', 'C3': '352 k', 'C4': '714 k', 'C5': '214 k', 'C5x2': '228 k', 'runsh': '1.275 M', 'C1': '109.6 k', 'C2_iter': '1.494 M'}
ALG = {'C2': 138.4e6, 'C3': 234e6, 'C4': 138.4e6, 'C5': 2.21e9, 'runsh': 4.33e9}   # algorithmic bytes per launch (48 N + 18 N per sweep x NS x MOD)


def traffic_cell(t):
    m = PM[t]
    a = ALG[t]
    tr = m['traffic']
    s = ('%.0f MB' % (tr / 1e6) if tr < 1e9 else '%.2f GB' % (tr / 1e9) if tr < 1e10 else '%.1f GB' % (tr / 1e9))
    if m['other']:
        lo, hi = min(m['other'] + [tr]), max(m['other'] + [tr])
        s += ' in this run, %.0f-%.0f MB over the rounds\' runs of the same kernel (write-back cadence of the hand-over lines, H-7.3)' % (lo / 1e6, hi / 1e6)
    s += ' / %s' % ('%.0f MB' % (a / 1e6) if a < 1e9 else '%.2f GB' % (a / 1e9))
    if tr / a >= 5:
        s += ' = %.0fx' % (tr / a)
    return s


def design_rows():
    rows = []
    def row(label, key, pm=None, extra_frac=True):
        d = D[key]; r = d['roofline']; cb = d.get('cpu_baseline') or {}
        cells = [label, '**%s** (%.2f)' % (rate(d['value']), kpc(r)) if pm else '%s (%.2f)' % (rate(d['value']), kpc(r)),
                 '%s (%.2f)' % (rate(d['window']['value']), kpc(d['window'])),
                 '%.2f %% / %.2f %% (%.2f %%)' % (100 * r['frac'], 100 * r['frac_executed'], 100 * d['window']['frac']),
                 '%.2f' % r['list_rebuilds_per_sweep'], '%.0f %%' % (100 * PM[pm]['valu']) if pm else '', traffic_cell(pm) if pm else '',
                 '%s (%d thr) / %s' % (ks(cb['value']), cb['cores'], ks(cb['single_thread']['value'])) if cb else '',
                 '%.0fx' % (d['value'] / cb['value']) if cb else '', R3.get(key, '')]
        rows.append('| ' + ' | '.join(cells) + ' |')
    row('C2 (64 x 256, Q = 4)', 'C2', 'C2')
    row("C2, the driver's flags (warm-up 5, 20 steps)", 'driver')
    if 'C2_record' in D:
        rec = D['C2_record']['record']
        rows.append('| C2 with outputs on (`--record`, §7.5) | %s (%.2f); %s with outputs off in the same run: I/O share %.1f %% | | | | | | | | — |'
                    % (rate(rec['outputs_on']), rec['kernel_avg_ms_on'], rate(rec['outputs_off']), 100 * rec['io_share']))
    row('C3 share (32 x 864, Q = 8)', 'C3', 'C3')
    row('C4 Al EAM (64 x 256, Q = 4)', 'C4', 'C4')
    row('C5 share (128 x 2048, Q = 2)', 'C5', 'C5')
    row('C5 share as a grid of twice the chip (`NM_OVERSUBSCRIBE=1`, Q = 4)', 'C5x2')
    row('run.sh setting (1024 x 500, Q = 1, half lists; 78.2 GB = 18x with full lists)', 'runsh', 'runsh')
    row('C1 (2 x 2 x 256, Q = 8: 32 of 256 CUs)', 'C1')
    d = D['C2_iter']; cb = d['cpu_baseline']
    rows.append('| C2, iterative position moves (`--iterative`: cycles 3-7, window only) | — | %s (%.2f) | %.2f %% | %.2f | | | %s / %s | %.0fx | %s |'
                % (rate(d['window']['value']), kpc(d['window']), 100 * d['window']['frac'], d['window']['list_rebuilds_per_sweep'],
                   ks(cb['value']), ks(cb['single_thread']['value']), d['value'] / cb['value'], R3['C2_iter']))
    rows.append('| strong-scaling legs of the 8 x 8 grid: 32 / 16 / 8 replicas on one GPU (Q = 8) | %s / %s / %s (%.2f / %.2f / %.2f) | | | | | | | | 813 k / 416 k / 212 k |'
                % (rate(D['C2_rows4']['value']), rate(D['C2_rows2']['value']), rate(D['C2_rows1']['value']), kpc(D['C2_rows4']['roofline']),
                   kpc(D['C2_rows2']['roofline']), kpc(D['C2_rows1']['roofline'])))
    head = ['| preset | sustained sweeps/s (kernel ms per cycle) | window (kernel ms per cycle) | fp64 frac, sustained: algorithmic / executed (window) | rebuilds per sweep | VALU issuing (PMC) | HBM-side traffic per cycle / algorithmic | CPU all cores / one thread | x all-core CPU | round 3 sustained |',
            '|---|---|---|---|---|---|---|---|---|---|']
    return '\n'.join(head + rows)


def baseline_rows():
    cols = ['C1', 'C2', 'C3', 'C4', 'C5', 'runsh']
    def cell(fn):
        return ' | '.join(fn(c) for c in cols)
    def cpu1(c): return '{:,.0f}'.format(D[c]['cpu_baseline']['single_thread']['value']).replace(',', ' ')
    def cpua(c): cb = D[c]['cpu_baseline']; return '{:,.0f}'.format(cb['value']).replace(',', ' ') + ' (%d threads)' % cb['cores']
    def sus(c):
        s = '{:,.0f}'.format(D[c]['value']).replace(',', ' ')
        if c == 'C2': s = '**%s** (driver\'s flags, warm-up 5 / 20 steps: %s)' % (s, '{:,.0f}'.format(D['driver']['value']).replace(',', ' '))
        if c == 'C5': s = '**%s** (%s as a grid of twice the chip)' % (s, '{:,.0f}'.format(D['C5x2']['value']).replace(',', ' '))
        if c == 'runsh': s = '**%s**' % s
        return s
    def win(c):
        s = '{:,.0f}'.format(D[c]['window']['value']).replace(',', ' ')
        if c == 'C2': s += ' (driver\'s flags: %s)' % '{:,.0f}'.format(D['driver']['window']['value']).replace(',', ' ')
        return s
    r3 = {'C1': '109 614', 'C2': '1 394 520', 'C3': '352 236', 'C4': '713 500', 'C5': '214 180', 'runsh': '1 275 299'}
    def sp(c): return ('**%.0fx** (target >= 10x)' if c == 'C2' else '%.0fx') % (D[c]['value'] / D[c]['cpu_baseline']['value'])
    def kms(c): return '%.2f' % kpc(D[c]['roofline'])
    def fr(c):
        r = D[c]['roofline']
        if c == 'C1': return '%.2f %% (32 CUs)' % (100 * r['frac'])
        if c == 'C2': return '%.2f TF (%.2f %%; executed %.2f %%; window %.2f %%)' % (r['achieved'], 100 * r['frac'], 100 * r['frac_executed'], 100 * D[c]['window']['frac'])
        return '%.2f TF (%.2f %%)' % (r['achieved'], 100 * r['frac'])
    def hb(c):
        if c == 'C1': return '—'
        h = D[c]['roofline_hbm']; m = PM[c]
        s = '%.1f (%.2f %%); ' % (h['achieved'], 100 * h['frac'])
        return s + traffic_cell(c).split(' / ')[0] + ((' (%.0fx)' % (m['traffic'] / ALG[c])) if m['traffic'] / ALG[c] >= 5 else '')
    def va(c): return '—' if c == 'C1' else '%.0f %%' % (100 * PM[c]['valu'])
    def rb(c): return '%.2f' % D[c]['roofline']['list_rebuilds_per_sweep']
    rows = ['| Quantity | C1 (2x2, 4^3) | C2 (1 GPU) | C3 share (32 x 864) | C4 (Al EAM, 64 x 256) | C5 share (128 x 2048) | run.sh setting (1024 x 500) |', '|---|---|---|---|---|---|---|',
            '| CPU restatement, 1 core, sweeps/s | ' + cell(cpu1) + ' |', '| CPU restatement, all cores, sweeps/s | ' + cell(cpua) + ' |',
            '| MI355X sweeps/s, sustained (= `value`) | ' + cell(sus) + ' |', '| MI355X sweeps/s, window | ' + cell(win) + ' |',
            '| round 3, sustained | ' + cell(lambda c: r3[c]) + ' |', '| speed-up vs all-core CPU (sustained) | ' + cell(sp) + ' |',
            '| kernel ms per cycle (one block of MOD moves per replica), sustained | ' + cell(kms) + ' |', '| fp64 FLOP/s, algorithmic, sustained (fraction of 78.6 TF) | ' + cell(fr) + ' |',
            '| algorithmic HBM GB/s (fraction of 8 TB/s); measured HBM-side traffic per cycle (run.sh setting: 18x with full lists) | ' + cell(hb) + ' |',
            '| VALU issuing, share of SIMD cycles (PMC, equilibrated launches) | ' + cell(va) + ' |', '| list rebuilds per sweep, sustained | ' + cell(rb) + ' |']
    return '\n'.join(rows)


def put(path, tag, text):
    s = open(path).read()
    a, b = '<!-- numbers:%s:begin -->' % tag, '<!-- numbers:%s:end -->' % tag
    if a not in s:
        raise SystemExit('%s has no marker %s' % (path, a))
    i, j = s.index(a) + len(a), s.index(b)
    open(path, 'w').write(s[:i] + '\n' + text + '\n' + s[j:])


commit = PM['C2']['commit']
put('DESIGN.md', 'design-7.1', design_rows())
put('BASELINE.md', 'baseline-5', baseline_rows())
for path in ('DESIGN.md', 'BASELINE.md', 'profiles/README.md'):
    s = open(path).read()
    a, b = '<!-- numbers:commit:begin -->', '<!-- numbers:commit:end -->'
    while a in s:
        i = s.index(a); j = s.index(b, i)
        s = s[:i] + '\x00' + '`%s`' % commit + '\x01' + s[j + len(b):]
    s = s.replace('\x00', a).replace('\x01', b)
    open(path, 'w').write(s)
print('tables regenerated for commit', commit)
