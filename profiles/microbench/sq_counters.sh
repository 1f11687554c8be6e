cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -o run --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:40]
    if 'small_solve' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in acc:
    for c in acc[k]: print(k, c, "%.4g per launch" % (acc[k][c]/cnt[(k,c)]))
PY
done
