set -u
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/s60
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the driver's own command under rocprofv3 (kernel trace + stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_shape_under_rocprof.json 2> $OUT/trace.log
python3 - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows:
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in d.values())
with open(out + '/rocprofv3_summary_driver_shape.txt', 'w') as o:
    o.write('== kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5) ==\n')
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
        o.write(f"{k[:100]:100s} calls={len(v)} avg_ns={sum(v)/len(v):.1f} min_ns={min(v)} max_ns={max(v)} pct={100*sum(v)/tot:.2f}\n")
    # the 25 launches of the step kernel in launch order (5 warm-up + 20 timed)
    st = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'step_kernel_pipe' in r['Kernel_Name']]
    st.sort()
    o.write('step kernel launches in order: duration ns / gap to previous end ns\n')
    for i, (s, e) in enumerate(st):
        o.write(f"  {i:2d} {e - s:6d} {(s - st[i-1][1]) if i else 0:9d}\n")
print(open(out + '/rocprofv3_summary_driver_shape.txt').read())
P
rm -rf $OUT/trace
cd $GRAFT_REPO_ROOT
for w in cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - $w $OUT/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done
