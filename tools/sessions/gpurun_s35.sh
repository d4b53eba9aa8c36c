# me20 with the tile's state in LDS (PCG_Q_TILE=512: two sub-tiles per workgroup): the HBM counters of that shape
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s35
PCG_Q_TILE=512 PROF_PMC_STEPS=118 PROF_PMC_WARMUP=12 bash tools/prof.sh me20_t512 --workload me20 > /dev/null 2>&1
python - <<'P' | tee gpurun_out/s35/me20_t512_xlds_traffic.txt
import csv,glob
def mean(pat,name):
    v=[]
    for f in glob.glob('gpurun_out/prof_me20_t512/'+pat+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'step_kernel_queue' in r['Kernel_Name'] and r['Counter_Name']==name: v.append(float(r['Counter_Value']))
    return sum(v)/len(v), len(v)
f,n=mean('pmc_FETCH_SIZE','FETCH_SIZE'); w,_=mean('pmc_WRITE_SIZE','WRITE_SIZE')
rd=f*1024*2; wr=w*1024
alg=434*(1<<18)
print('me20, PCG_Q_TILE=512 (state parked in LDS, two sub-tiles per workgroup): launches %d' % n)
print('read %.1f MB  write %.1f MB  total %.1f MB per launch = %.2f x the algorithmic %.1f MB' % (rd/1e6, wr/1e6, (rd+wr)/1e6, (rd+wr)/alg, alg/1e6))
P
grep step_kernel gpurun_out/prof_me20_t512/summary.txt | head -2 | tee -a gpurun_out/s35/me20_t512_xlds_traffic.txt
