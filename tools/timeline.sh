#!/bin/bash
# per-launch kernel durations of the last benchmark frame (rocprofv3 kernel trace)
out=gpurun_out/timeline_$1; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-latency-leg --no-pmc ${BENCH_ARGS:---steps 128 --warmup 128} > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
out=sys.argv[1]
f=glob.glob(out+'/kt/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last frame = from last k_init on
gkey='Grid_Size_X' if 'Grid_Size_X' in rows[0] else 'Grid_Size'
inits=[i for i,r in enumerate(rows) if 'k_init' in r['Kernel_Name']]
big=max(int(rows[i][gkey]) for i in inits)
idx=[i for i in inits if int(rows[i][gkey])==big]     # the largest passes (all frames in flight)
nxt=[i for i in inits if i>idx[-1]]
last=rows[idx[-1]:(nxt[0] if nxt else len(rows))]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    n=r['Kernel_Name'].split('(')[0].replace('void mcrt::','').replace('mcrt::','')
    print("%-28s start %9.1f us  dur %8.1f us  grid %s wg %s vgpr %s lds %s" % (n, (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X',r.get('Grid_Size','')), r.get('Workgroup_Size_X',r.get('Workgroup_Size','')), r.get('VGPR_Count',''), r.get('LDS_Block_Size','')))
PY
