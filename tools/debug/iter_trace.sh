# Per-dispatch kernel trace of a short bench run; prints per-iteration kernel durations of the last step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/itrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/itrace -o t -- python3 $R/bench.py --steps 2 --warmup 1 --resident --no-cpu-baseline --traffic none > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/itrace/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = lambda r: r['Kernel_Name']
# keep the last 16 solve kernels' worth of dispatches
idx = [i for i, r in enumerate(rows) if 'gn_solve' in names(r)]
last = idx[-16:]
start = idx[-17] + 1
prev_end = int(rows[start - 1]['End_Timestamp'])
it = 0
cur = {}
t_iter0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    n = names(r)
    key = 'walk' if 'walk_kernel' in n else 'deep' if 'walk_list' in n else 'redo' if 'redo' in n else 'accum' if 'accum' in n else 'solve' if 'gn_solve' in n else 'other'
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev_end) / 1e3
    prev_end = int(r['End_Timestamp'])
    cur[key] = cur.get(key, 0) + d
    cur['gaps'] = cur.get('gaps', 0) + max(gap, 0)
    if key == 'solve':
        tot = (int(r['End_Timestamp']) - t_iter0) / 1e3
        print('iter %2d: %s total %.0f us' % (it, ' '.join('%s %.0f' % (k, v) for k, v in cur.items()), tot))
        it += 1; cur = {}; t_iter0 = int(r['End_Timestamp'])
PY
