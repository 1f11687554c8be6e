set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03a_gputests.log
python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
python bench.py --workload linear --no-cpu > gpurun_out/r03a_linear.json 2>> gpurun_out/r03a_bench.err
python bench.py --workload chain5 --no-cpu --steps 10 --warmup 3 > gpurun_out/r03a_chain5.json 2>> gpurun_out/r03a_bench.err
python bench.py --workload chain7 --no-cpu --steps 5 --warmup 2 > gpurun_out/r03a_chain7.json 2>> gpurun_out/r03a_bench.err
python bench.py --workload td3 --steps 30 --warmup 10 > gpurun_out/r03a_td3.json 2>> gpurun_out/r03a_bench.err
cat gpurun_out/r03a_gputests.log
