# usage (on the GPU box): bash profiles/microbench/chain_probe.sh [tests] [n_mass ...]  — chain parity tests, bench line, phase profile
# (needs ab/prof.so built with -DMPCRL_PROFILE_PHASES for the phase part)
R=$GRAFT_REPO_ROOT; cd $R
if [ "$1" == "tests" ]; then shift
  timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_surface.py -m gpu -q -k "chain" --timeout 600 --timeout-method=thread 2>&1 | tail -15
fi
for n in "$@"; do
  steps="--steps 10 --warmup 3"; [ $n == 7 ] && steps="--steps 5 --warmup 2"
  python bench.py --workload chain$n $steps --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain$n', 'ms %.3f' % d['ms_per_step'], 'ipm', d['config']['ipm_iters_mean'], 'sqp', d['config']['sqp_iters_mean'], 'conv', d['config']['converged_fraction'])"
  [ -f ab/prof.so ] && MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/chain_phases.py $n 2>/dev/null
done
