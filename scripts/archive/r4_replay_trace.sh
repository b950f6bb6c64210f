#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4rt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/r4_replay_one.py 8 3 1024 > $OUT/log.txt 2>&1; echo "rc=$?"
cd $R
f=$(ls $OUT/t/*/*kernel_trace.csv | head -1)
python - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys, collections
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# the last evaluation: from the last gram_kernel launch group; find last 'logp'/'rows_dot' end
ends = [i for i, r in enumerate(tr) if 'rows_dot_ss' in r['Kernel_Name']]
# take kernels between the 2nd-last and last rows_dot (the final replayed step)
lo = ends[-1 - len([1 for _ in range(1)])] if len(ends) > 8 else 0
# simpler: last 1/4 of the trace after the final replay_panel kernels begin
rep = [i for i, r in enumerate(tr) if 'replay_panel' in r['Kernel_Name']]
n_per = 31
start = rep[-n_per] - 40 if len(rep) >= n_per else 0
tr2 = tr[max(start, 0):]
t0 = int(tr2[0]['Start_Timestamp'])
print('span %.2f ms' % ((max(int(r['End_Timestamp']) for r in tr2) - t0) / 1e6))
byq = collections.defaultdict(lambda: [0, 0.0])
for r in tr2:
    k = (r['Queue_Id'], r['Kernel_Name'].split('(')[0][:50])
    byq[k][0] += 1; byq[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for k, v in sorted(byq.items(), key=lambda kv: -kv[1][1])[:25]:
    print('q%s %-52s %4d launches %8.2f ms' % (k[0], k[1], v[0], v[1]))
print('--- timeline (start ms, dur us, queue, grid, kernel)')
for r in tr2[:260]:
    print('%8.3f %8.1f q%s %6d %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'],
          int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Kernel_Name'].split('(')[0][:60]))
PY
