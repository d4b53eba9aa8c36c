set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s14; mkdir -p $O
X=$PWD/pc-gym_amd/libpcgym_hip_exp5.so
# (a) the experiment build (PCG_INT_RODAS4 plans run the fifth-order pair) against the oracle's rodas5, bit for bit
PCGYM_HIP_LIB=$X python - <<'P' 2>&1 | tail -5
import sys, numpy as np, torch, ctypes as C
sys.path[:0]=['tests','tests/golden']
import helpers as H
from oracle import oracle as O
from test_gpu_parity import _plan_for
from test_oracle_golden import _spec_for_integration
l=O.lib(); l.orc_set_ros_pair.restype=None; l.orc_set_ros_pair(5)
for fix,kw in (("multistage_extraction",dict(integrator="rodas4",rtol=1e-6,atol=1e-8)),("multistage_extraction_d",dict(integrator="rodas4",rtol=6e-8,atol=6e-8))):
    g=H.gold("tight_"+fix)
    spec=_spec_for_integration("multistage_extraction",float(g["dt"]),g["u"].shape[1],cooperative=False,**kw)
    lib,plan=_plan_for(spec,torch)
    xs,us=g["x"].T.copy(),g["u"].T.copy()
    x=torch.tensor(xs,device="cuda"); u=torch.tensor(us,device="cuda")
    ns=torch.zeros((2,x.shape[1]),dtype=torch.int32,device="cuda")
    assert lib.pcg_integrate(plan,x.shape[1],x.data_ptr(),u.data_ptr(),ns.data_ptr(),None)==0
    torch.cuda.synchronize()
    want,ns_o=O.integrate(spec,xs,us)
    got=x.cpu().numpy()
    print(fix,"steps identical",bool(np.all(ns.cpu().numpy()==ns_o)),"max rel diff",float(np.max(np.abs(got-want)/np.abs(want))),"bitwise",bool(np.array_equal(got,want)),"attempts mean",ns_o.sum(0).mean(), "err vs tight", float(np.max(np.abs(got-g["xf"].T)/np.abs(g["xf"].T))))
P
run() { # label, env..., -- args
  local label=$1; shift
  ( "$@" > $O/b.json 2> $O/b.err ) ; python - "$label" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:44s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us sane {d['config'].get('sane')} steps {d['config'].get('attempts_per_env_step', d['config'].get('mean_attempts'))}")
except Exception as e: print(sys.argv[1],'FAILED',e, open(sys.argv[2].replace('.json','.err')).read()[-600:])
P
}
for rep in 1 2; do
run "me10_ros4 pair4 default"          timeout 600 python bench.py --workload me10_ros4 --no-cpu-baseline
run "me10_ros4 pair4 coop off"         timeout 600 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr 0
run "me10_ros4 pair5 3e-8 coop off"    env PCGYM_HIP_LIB=$X timeout 600 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr 0
run "me10_ros4 pair5 6e-8 coop off"    env PCGYM_HIP_LIB=$X PCG_BENCH_ME_TOL=6e-8 timeout 600 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr 0
run "mixed pair4 default"              timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed pair5 6e-8 coop off"        env PCGYM_HIP_LIB=$X PCG_BENCH_ME_TOL=6e-8 timeout 600 python bench.py --workload mixed --no-cpu-baseline --coop-thr 0
done
