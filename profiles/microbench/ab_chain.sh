# usage (on the GPU box): bash profiles/microbench/ab_chain.sh lib1.so lib2.so ...   — chain5 / chain7 bench lines of library builds, alternating, 2 rounds
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2; do
  for l in "$@"; do
    if [ "$l" == "default" ]; then unset MPCRL_LIB_PATH; else export MPCRL_LIB_PATH=$PWD/$l; fi
    for n in 5 7; do
      steps="--steps 10 --warmup 3"; [ $n == 7 ] && steps="--steps 5 --warmup 2"
      python bench.py --workload chain$n $steps --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l chain$n', 'ms %.3f' % d['ms_per_step'], 'ipm', d['config']['ipm_iters_mean'], 'sqp', d['config']['sqp_iters_mean'], 'conv', d['config']['converged_fraction'])"
    done
  done
done
