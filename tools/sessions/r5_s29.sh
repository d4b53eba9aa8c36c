set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s29; mkdir -p $O
for args in "rk4 cv8 dopri5 tsit5" "rk4 cv8 --classic"; do
  tag=$(echo $args | tr ' ' '_' | tr -d '-')
  timeout 1500 python tools/integrator_sweep.py $args > $O/sweep_$tag.txt 2>&1; echo "rc $?" >> $O/sweep_$tag.txt
  echo "== $args: $(grep -c 'worst rel diff' $O/sweep_$tag.txt) combinations"; grep -E "BAD|bad combinations|rc |Error" $O/sweep_$tag.txt | head -12
done
