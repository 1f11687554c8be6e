cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks_seq
timeout 200 rocprofv3 --kernel-trace -d /tmp/ks_seq -o run --output-format csv -- python $R/bench.py --workload td3 --steps 20 --warmup 5 --no-cpu > /tmp/ks_seq.out 2>/tmp/ks_seq.err
f=$(find /tmp/ks_seq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# steps: the roll-out solve is the plain small_solve_kernel<CartpoleDev> launch with grid of E instances; take the last 2 occurrences
idx = [i for i, n in enumerate(names) if "td3_cartpole_collect_kernel" in n]
m = len(idx) * 3 // 4
a, b = idx[m], idx[m + 2]      # two steps (policy / no policy), inside the timed region
t0 = int(rows[a]["Start_Timestamp"])
def short(n):
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:150]
for r in rows[a:b]:
    print("%8.1f us  +%6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, short(r["Kernel_Name"])))
PY
