# usage (on the GPU box): bash profiles/microbench/hbm_traffic.sh <tag> <steps> <warmup> <bench.py args...>
# HBM traffic of one bench.py step from the TCC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (they do not fit
# one pass), --pmc only together with --kernel-trace (MI355X_MICROARCH.md, HBM / rocprofv3 sections).  Counter values are KB at
# the L2's memory-side interface; per kernel they are summed over all dispatches and divided by the number of steps run
# (steps + warmup).  Writes gpurun_out/<tag>_hbm.json.
tag=$1; steps=$2; warm=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_$c -o run --output-format csv -- python $R/bench.py --steps $steps --warmup $warm --no-cpu --no-secondary --settle 0 "$@" > /tmp/pm_$c.out 2>&1
done
python3 - "$tag" "$steps" "$warm" "$R" "$@" <<'PY'
import csv, sys, glob, json, collections
tag, steps, warm, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n = steps + warm
out = {"command": "bench.py --steps %d --warmup %d --no-cpu %s" % (steps, warm, " ".join(sys.argv[5:])), "steps_counted": n, "unit": "bytes per step (counter KB x 1024)", "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pm_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mpcrl::", "")
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in acc:
        out["kernels"].setdefault(k, {})[c] = acc[k] * 1024.0 / n
        out["kernels"][k]["dispatches_per_step"] = cnt[k] / n
tot_f = sum(v.get("FETCH_SIZE", 0.0) for v in out["kernels"].values())
tot_w = sum(v.get("WRITE_SIZE", 0.0) for v in out["kernels"].values())
out["fetch_bytes_per_step_raw"], out["write_bytes_per_step"] = tot_f, tot_w
json.dump(out, open("%s/gpurun_out/%s_hbm.json" % (R, tag), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
