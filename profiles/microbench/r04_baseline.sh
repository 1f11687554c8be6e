# usage (on the GPU box): bash profiles/microbench/r04_baseline.sh <tag>  — GPU suite + one bench line per workload
set -x
T=${1:-r04a}
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread -x 2>&1 | tail -15 > gpurun_out/${T}_gputests.log
for w in cartpole linear chain5 chain7 td3; do
  extra=""; [ $w != cartpole ] && extra="--workload $w"
  steps="--steps 50 --warmup 10"; [ $w == chain5 ] && steps="--steps 10 --warmup 3"; [ $w == chain7 ] && steps="--steps 5 --warmup 2"; [ $w == td3 ] && steps="--steps 60 --warmup 10"
  timeout 300 python bench.py $extra $steps --no-cpu > gpurun_out/${T}_${w}_bench.json 2>/dev/null
done
cat gpurun_out/${T}_gputests.log
for w in cartpole linear chain5 chain7 td3; do python -c "import json,sys; d=json.loads(open('gpurun_out/${T}_${w}_bench.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'])"; done
