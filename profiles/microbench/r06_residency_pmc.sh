# usage (on the GPU box): bash profiles/microbench/r06_residency_pmc.sh <workload> <kernel-substring> <out-file> <batch> [<batch> ...]
# Why does every memory phase of the chain SQP kernel run 30-70 % slower at 1024 resident wavefronts than at 256?
# Translation (UTCL1), L2 <-> fabric (TCC_EA*), latency and stall counters of ONE kernel at several batch sizes: separate rocprofv3
# passes (--pmc only with --kernel-trace), summed over the dispatches of the matching kernel and divided by their number.
wl=$1; pat=$2; out=$3; shift 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $R/gpurun_out/$out
for B in "$@"; do
 echo "== $wl batch $B, kernel *$pat*" >> $R/gpurun_out/$out
 for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
            "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCC_BUSY_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
            "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -o run --output-format csv -- python $R/bench.py --workload $wl --batch $B --steps 2 --warmup 1 --settle 0 --no-cpu --no-secondary > /dev/null 2>/tmp/pm.err
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "  ($set: no counter file: $(tail -1 /tmp/pm.err | cut -c1-160))" >> $R/gpurun_out/$out; continue; fi
  python3 - "$f" "$pat" >> $R/gpurun_out/$out <<'PY'
import csv, sys, collections
acc=collections.defaultdict(float); cnt=collections.Counter(); dur=[]
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
for c in acc: print("  %-48s %.6g per dispatch (%d dispatches)" % (c, acc[c]/cnt[c], cnt[c]))
PY
 done
 # kernel durations of the same configuration (no counters)
 rm -rf /tmp/ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks -o run --output-format csv -- python $R/bench.py --workload $wl --batch $B --steps 3 --warmup 1 --settle 0 --no-cpu --no-secondary > /dev/null 2>&1
 f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "$pat" "$f" | cut -d, -f1-4 | cut -c1-200 >> $R/gpurun_out/$out
done
cat $R/gpurun_out/$out
