set -u
cd $GRAFT_REPO_ROOT
export PMC_ROUND=r5
bash tools/prof_all.sh 2>&1 | tail -60
ls gpurun_out/ | head -30
