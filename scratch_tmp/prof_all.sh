cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof -o run --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/r01_v4_bench.json 2>/dev/null
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_v4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm -o run --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
  cp $(find /tmp/pm -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc_$c.csv
done
bash $R/profiles/microbench/sq_counters.sh > $R/gpurun_out/r01_v4_sq_counters.txt 2>&1
python $R/bench.py > $R/gpurun_out/bench_full.json 2>/dev/null
for a in "--batch 3072" "--batch 32768" "--rti" "--no-sens"; do echo "$a"; python $R/bench.py --steps 10 --warmup 2 --no-cpu $a 2>&1 | tail -1 | cut -c50-200; done
head -4 $R/gpurun_out/r01_v4_kernel_stats.csv | cut -c1-150
