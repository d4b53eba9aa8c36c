set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s6; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
for w in cstr cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  timeout 600 python bench.py --workload $w $( [ $w = cstr ] || echo --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err
  python - $w $O/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) copy {r.get('copy_ceiling_GBps')} sane {d['config']['sane']}")
    if 'cpu_baseline' in d:
        c=d['cpu_baseline']; print('   cpu: 1 thr %.3e, %d thr %.3e, phys %d: %.3e, all %d: %.3e quota %s pinned %s' % (c['one_thread_env_steps_per_s'], c['cores'], c['value'], c['physical_cores']['cores'], c['physical_cores']['value'], c['all_host_cpus']['cores'], c['all_host_cpus']['value'], c['cgroup_cpu_quota_cpus'], c['threads_pinned_first_touch_parallel']))
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_shape_$i.json 2>/dev/null
  PCG_BENCH_NACT=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver_shape_nact8_$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5s6/driver_shape*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], '%.4e'%d['value'], 'ms/step %.3f us'%(d['ms_per_step']*1e3), 'kernel %.2f'%d['roofline']['kernel_avg_us'])
P
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $O/valu_counters.txt; wc -l $O/valu_counters.txt; head -40 $O/valu_counters.txt | tr '\n' ' '
