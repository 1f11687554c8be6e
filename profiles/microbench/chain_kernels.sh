# usage (on the GPU box): bash profiles/microbench/chain_kernels.sh [n_mass ...] — per-kernel average durations of one chain bench (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for n in "$@"; do
  steps="--steps 6 --warmup 2"; [ $n == 7 ] && steps="--steps 4 --warmup 1"
  timeout 300 bash profiles/microbench/kstats.sh probe_chain$n --workload chain$n $steps --no-cpu > /dev/null 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/probe_chain${n}_kernel_stats.csv")))
tot = 0.0
for r in rows:
    if "mpcrl" in r["Name"] and "cold_iterate" not in r["Name"]:
        us = float(r["AverageNs"]) / 1e3; tot += us
        print("chain$n %-28s %9.1f us" % (r["Name"].split("mpcrl::")[1].split("<")[0], us))
print("chain$n sum of kernels %.3f ms" % (tot / 1e3))
PY
done
