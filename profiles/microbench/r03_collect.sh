# usage (on the GPU box): bash profiles/microbench/r03_collect.sh   — everything the round's profiles/ files are made from
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 400 --timeout-method=thread 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > gpurun_out/r03_gputests.log
for w in cartpole linear chain5 chain7 td3; do
  extra=""; [ $w != cartpole ] && extra="--workload $w"
  steps="--steps 50 --warmup 10"; [ $w == chain5 ] && steps="--steps 10 --warmup 3"; [ $w == chain7 ] && steps="--steps 5 --warmup 2"; [ $w == td3 ] && steps="--steps 60 --warmup 10"
  python bench.py $extra $steps > gpurun_out/r03_final_${w}_bench.json 2>/dev/null
  # (rocprofv3 around the graph-replayed TD3 loop did not return on two boxes in round 3: its kernel stats are taken from an eager run,
  #  timeout 150 rocprofv3 --kernel-trace --stats ... -- python bench.py --workload td3 --no-graph --steps 20 --warmup 5)
  [ $w != td3 ] && timeout 300 bash profiles/microbench/kstats.sh r03_$w $extra $steps --no-cpu > /dev/null 2>&1
done
python bench.py --no-sens --no-cpu > gpurun_out/r03_final_cartpole_nosens_bench.json 2>/dev/null
python bench.py --rti --no-cpu > gpurun_out/r03_final_cartpole_rti_bench.json 2>/dev/null
for b in 1024 2048 3072 8192 32768; do python bench.py --no-cpu --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($b, d['value'], d['ms_per_step'])"; done > gpurun_out/r03_batch_sweep.txt
timeout 300 bash profiles/microbench/hbm_traffic.sh cartpole 3 1 > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh linear 3 1 --workload linear > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh chain5 2 1 --workload chain5 > /dev/null 2>&1
timeout 400 bash profiles/microbench/hbm_traffic.sh chain7 2 1 --workload chain7 > /dev/null 2>&1
ls -la gpurun_out | tail -30
