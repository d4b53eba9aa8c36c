import copy, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests'); sys.path.insert(0, ROOT+'/tests/golden')
import torch
import scenarios as SC
from tools.user_model_probe import run
S=SC.scenarios()
for name in ["heat_exchanger_sp","biofilm_sp","me_reactive","me_canonical","cstr_canonical"]:
    p=copy.deepcopy(S[name]["env_params"]); p.pop("noise",None); p.pop("noise_percentage",None)
    p.update(integrator="tsit5", rtol=1e-8, atol=1e-8)
    t,_=run(p, 1<<18, steps=30, reps=3)
    print("%-20s tsit5 1e-8: %.1f us per step of 2^18 envs" % (name,t), flush=True)
