export TMPDIR=/tmp MCRT_TUNING=1 MCRT_PACKET_BOUNCES=2
out=gpurun_out/pmc_packet; rm -rf $out; mkdir -p $out
B="python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-latency-leg --no-pmc"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d $out/q1 -- $B > $out/q1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $out/q2 -- $B > $out/q2.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/q*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0].replace('void mcrt::', '').replace('mcrt::','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ('k_trace_packet', 'k_trace_lane<false>'):
    v = agg.get(k)
    if not v: continue
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, 'launches', len(v['SQ_INSTS_VALU']))
    for c in sorted(a): print('   %-26s %14.1f' % (c, a[c]))
PY
tail -3 $out/q2.log
