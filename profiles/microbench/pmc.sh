# usage (on the GPU box): bash profiles/microbench/pmc.sh <kernel-name-substring> <bench.py args...>
# SQ / TCC counter passes (separate rocprofv3 runs, --pmc only with --kernel-trace), summed per launch of the matching kernel.
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -o run --output-format csv -- python $R/bench.py --no-secondary "$@" > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
for c in acc: print(c, "total %.5g over %d dispatches" % (acc[c], cnt[c]))
PY
done
