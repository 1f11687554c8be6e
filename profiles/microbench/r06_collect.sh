# usage (on the GPU box): bash profiles/microbench/r06_collect.sh [quick]   — everything the round's profiles/r05_* files are made from
set -x
R=$GRAFT_REPO_ROOT; cd $R
T=r06
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > gpurun_out/${T}_gputests.log
fi
# the driver's line: headline + the secondary configurations in one run
python bench.py > gpurun_out/${T}_final_bench.json 2>/dev/null
for w in cartpole linear chain5 chain7 td3; do
  extra=""; [ $w != cartpole ] && extra="--workload $w"
  steps="--steps 50 --warmup 10"; [ $w == chain5 ] && steps="--steps 10 --warmup 3"; [ $w == chain7 ] && steps="--steps 5 --warmup 2"; [ $w == td3 ] && steps="--steps 60 --warmup 10"
  [ $w != cartpole ] && python bench.py $extra $steps > gpurun_out/${T}_final_${w}_bench.json 2>/dev/null
  # (rocprofv3 around the graph-replayed TD3 loop did not return on two boxes in round 3: its kernel stats come from an eager run)
  [ $w != td3 ] && timeout 300 bash profiles/microbench/kstats.sh ${T}_$w $extra $steps --no-cpu > /dev/null 2>&1
done
timeout 200 bash profiles/microbench/kstats.sh ${T}_td3 --workload td3 --no-graph --steps 20 --warmup 5 --no-cpu > /dev/null 2>&1
python bench.py --no-sens --no-cpu --no-secondary > gpurun_out/${T}_final_cartpole_nosens_bench.json 2>/dev/null
python bench.py --rti --no-cpu --no-secondary > gpurun_out/${T}_final_cartpole_rti_bench.json 2>/dev/null
python bench.py --workload td3 --steps 60 --warmup 10 --cold-replay > gpurun_out/${T}_final_td3_cold_replay_bench.json 2>/dev/null
timeout 300 bash profiles/microbench/hbm_traffic.sh cartpole 3 1 > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh linear 3 1 --workload linear > /dev/null 2>&1
timeout 300 bash profiles/microbench/hbm_traffic.sh chain5 2 1 --workload chain5 > /dev/null 2>&1
timeout 400 bash profiles/microbench/hbm_traffic.sh chain7 2 1 --workload chain7 > /dev/null 2>&1
if [ -f ab/prof.so ]; then
  (MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/chain_phases.py 5; MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/chain_phases.py 7) > gpurun_out/${T}_chain_phases.txt 2>/dev/null
fi
(timeout 300 bash profiles/microbench/pmc.sh "chain_sqp_kernel" --workload chain5 --steps 2 --warmup 1 --no-cpu) > gpurun_out/${T}_pmc.txt 2>/dev/null
if [ -f ab/prof.so ]; then
  MPCRL_LIB_PATH=$PWD/ab/prof.so python profiles/microbench/lq_phases.py > gpurun_out/${T}_lq_phases.txt 2>/dev/null
fi
(echo "== lq_solve_kernel<3, 16>, bench.py --workload linear --steps 3 --warmup 1 (4 launches)"; timeout 300 bash profiles/microbench/pmc.sh "lq_solve_kernel" --workload linear --steps 3 --warmup 1 --no-cpu) > gpurun_out/${T}_lq_pmc.txt 2>/dev/null
MPCRL_LINEAR_SPL=1 python bench.py --workload linear --steps 50 --warmup 10 --no-cpu --no-secondary > gpurun_out/${T}_final_linear_spl1_bench.json 2>/dev/null
MPCRL_LINEAR_SPL=1 timeout 300 bash profiles/microbench/kstats.sh ${T}_linear_spl1 --workload linear --steps 50 --warmup 10 --no-cpu > /dev/null 2>&1
ls -la gpurun_out | tail -40
