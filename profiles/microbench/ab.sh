# usage (on the GPU box): bash profiles/microbench/ab.sh <bench args...> -- lib1.so lib2.so ...   (A/B of library builds, alternating, 2 rounds)
args=(); libs=(); seen=0
for a in "$@"; do if [ "$a" == "--" ]; then seen=1; elif [ $seen == 0 ]; then args+=("$a"); else libs+=("$a"); fi; done
for round in 1 2; do
  for l in "${libs[@]}"; do
    if [ "$l" == "default" ]; then unset MPCRL_LIB_PATH; else export MPCRL_LIB_PATH=$PWD/$l; fi
    python bench.py --no-cpu --no-secondary "${args[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'value %.4g' % d['value'], 'ms %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ipm', d['config'].get('ipm_iters_mean'), 'conv', d['config'].get('converged_fraction'))"
  done
done
