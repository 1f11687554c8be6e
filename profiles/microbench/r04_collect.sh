# usage (on the GPU box): bash profiles/microbench/r04_collect.sh   — everything the round's profiles/r04_* files are made from
set -x
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --timeout-method=thread 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > gpurun_out/r04_gputests.log
for w in cartpole linear chain5 chain7 td3; do
  extra=""; [ $w != cartpole ] && extra="--workload $w"
  steps="--steps 50 --warmup 10"; [ $w == chain5 ] && steps="--steps 10 --warmup 3"; [ $w == chain7 ] && steps="--steps 5 --warmup 2"; [ $w == td3 ] && steps="--steps 60 --warmup 10"
  python bench.py $extra $steps > gpurun_out/r04_final_${w}_bench.json 2>/dev/null
  # (rocprofv3 around the graph-replayed TD3 loop did not return on two boxes in round 3: its kernel stats come from an eager run)
  [ $w != td3 ] && timeout 300 bash profiles/microbench/kstats.sh r04_$w $extra $steps --no-cpu > /dev/null 2>&1
done
timeout 200 bash profiles/microbench/kstats.sh r04_td3 --workload td3 --no-graph --steps 20 --warmup 5 --no-cpu > /dev/null 2>&1
python bench.py --no-sens --no-cpu > gpurun_out/r04_final_cartpole_nosens_bench.json 2>/dev/null
python bench.py --rti --no-cpu > gpurun_out/r04_final_cartpole_rti_bench.json 2>/dev/null
timeout 300 bash profiles/microbench/hbm_traffic.sh cartpole 3 1 > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh linear 3 1 --workload linear > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh chain5 2 1 --workload chain5 > /dev/null 2>&1
timeout 400 bash profiles/microbench/hbm_traffic.sh chain7 2 1 --workload chain7 > /dev/null 2>&1
if [ -f ab/prof.so ]; then
  (MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/chain_phases.py 5; MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/chain_phases.py 7) > gpurun_out/r04_chain_phases.txt 2>/dev/null
fi
ls -la gpurun_out | tail -40
