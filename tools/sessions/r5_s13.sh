set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s13; mkdir -p $O
for rep in 1 2; do for cap in 1200 1500; do for cw in 1 2; do
  PCG_Q_W1CAP=$cap PCG_Q_COOPW=$cw timeout 600 python bench.py --workload mixed --no-cpu-baseline > $O/mixed.json 2> $O/mixed.err
  python - "mixed w1cap $cap coopw $cw" $O/mixed.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:28s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us ME kernel {r['kernel_avg_us']:9.2f} us sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done; done; done
export PMC_ROUND=r5 PCG_Q_W1CAP=1500
bash tools/prof_all.sh mixed 2>&1 | tail -4
python - <<'P'
import json
d=json.load(open('gpurun_out/pmc.json'))
for s,e in d['mixed']['segments'].items(): print(s, e['kernel'][:90], 'traffic MB', e['traffic_bytes_per_launch']/1e6, 'us', e.get('rocprof_avg_us'))
P
