set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s21
# the accuracy leg of the cpu_baseline for the workloads whose plain relative figure was questioned
for w in me20 me10 four_tank cryst_cv8; do
  python bench.py --workload $w > gpurun_out/s21/bench_${w}_with_cpu.json 2>/dev/null
  python - $w gpurun_out/s21/bench_${w}_with_cpu.json <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); c=d['cpu_baseline']
print(sys.argv[1], 'rel', c['step_vs_tight_max_rel_err'], 'in reference tolerances', c['step_vs_tight_max_err_in_reference_tolerances'], 'cpu', c['value'], c['all_host_cpus']['value'])
P
done
# eight ranks on this one device (gloo): the launch path of `python bench.py --gpus 8` and the host cost per launch
PCG_BENCH_BACKEND=gloo python bench.py --gpus 8 --batch 131072 --steps 590 --warmup 59 --no-cpu-baseline > gpurun_out/s21/bench_8ranks_one_device_cstr.json 2>gpurun_out/s21/8r.err
PCG_BENCH_BACKEND=gloo python bench.py --gpus 8 --workload mixed --batch 131072 --steps 59 --warmup 6 --no-cpu-baseline > gpurun_out/s21/bench_8ranks_one_device_mixed.json 2>>gpurun_out/s21/8r.err
PCG_BENCH_BACKEND=gloo python bench.py --gpus 2 --batch 131072 --steps 590 --warmup 59 --no-cpu-baseline > gpurun_out/s21/bench_2ranks_one_device_cstr.json 2>>gpurun_out/s21/8r.err
python bench.py --batch 131072 --steps 590 --warmup 59 --no-cpu-baseline > gpurun_out/s21/bench_1rank_b2p17_cstr.json 2>>gpurun_out/s21/8r.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s21/bench_*rank*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['config']
        print(f.split('/')[-1], 'n_gpus', d['n_gpus'], 'ranks_seen', c['ranks_seen'], 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'host us/launch (max over ranks)', c['host_launch_loop_us_per_step_max_over_ranks'], 'first envs', c['rank_first_env'])
    except Exception as e: print(f,'FAILED',e)
P
