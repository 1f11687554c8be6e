# usage (on the GPU box): bash profiles/microbench/r05_benchlines.sh — the round's bench lines once profiles/r05_hbm_traffic.json matches the sources
R=$GRAFT_REPO_ROOT; cd $R; T=r05
python bench.py > gpurun_out/${T}_final_bench.json 2>/dev/null
python bench.py --workload linear --steps 50 --warmup 10 > gpurun_out/${T}_final_linear_bench.json 2>/dev/null
MPCRL_LINEAR_SPL=1 python bench.py --workload linear --steps 50 --warmup 10 --no-cpu --no-secondary > gpurun_out/${T}_final_linear_spl1_bench.json 2>/dev/null
python bench.py --workload chain5 --steps 10 --warmup 3 > gpurun_out/${T}_final_chain5_bench.json 2>/dev/null
python bench.py --workload chain7 --steps 5 --warmup 2 > gpurun_out/${T}_final_chain7_bench.json 2>/dev/null
python bench.py --workload td3 --steps 60 --warmup 10 > gpurun_out/${T}_final_td3_bench.json 2>/dev/null
python bench.py --no-sens --no-cpu --no-secondary > gpurun_out/${T}_final_cartpole_nosens_bench.json 2>/dev/null
python bench.py --rti --no-cpu --no-secondary > gpurun_out/${T}_final_cartpole_rti_bench.json 2>/dev/null
grep -o '"traffic": [^,}]*' gpurun_out/${T}_final_*bench.json
