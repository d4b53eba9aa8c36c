set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s12; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_seulex.py tests/test_gpu_rodas4.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_erk.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
for lean in 1 0; do for rep in 1 2; do
  if [ $lean = 0 ]; then export PCG_Q_NOLEAN=1; else unset PCG_Q_NOLEAN; fi
  timeout 600 python bench.py --workload mixed --no-cpu-baseline > $O/mixed_lean$lean.json 2> $O/mixed.err
  python - "mixed lean $lean" $O/mixed_lean$lean.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:14s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us ME kernel {r['kernel_avg_us']:9.2f} us sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done; done
unset PCG_Q_NOLEAN
export PMC_ROUND=r5
bash tools/prof_all.sh mixed 2>&1 | tail -8
python - <<'P'
import json
d=json.load(open('gpurun_out/pmc.json'))
for s,e in d['mixed']['segments'].items(): print(s, 'traffic MB', e['traffic_bytes_per_launch']/1e6, 'us', e.get('rocprof_avg_us'))
P
