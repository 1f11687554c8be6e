"""Builds profiles/<round>_hbm_traffic.json (round tag = argv[1], default r05) from the per-workload counter passes that profiles/microbench/hbm_traffic.sh left in
gpurun_out/<workload>_hbm.json, and stamps it with the hash of the kernel sources (bench.csrc_hash): bench.py reports
roofline.traffic from this file only while the sources still hash to the same value.
    python profiles/microbench/merge_traffic.py        (in the dev container, after the GPU run)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE around `python bench.py --steps S --warmup W "
                 "--no-cpu [--workload ...]` (profiles/microbench/hbm_traffic.sh; MI355X_MICROARCH.md HBM section: counter values are KB at the L2's "
                 "memory-side interface, Infinity-Cache hits included; FETCH_SIZE under-reports 16-B/lane streaming reads by 2x on gfx950 and is "
                 "uncalibrated for the 8-B/lane accesses that dominate here, so the raw values are reported and the total is a LOWER bound). Per step = "
                 "summed over every kernel of one bench step; the one-off cold_iterate_kernel of handle creation and the framework's own kernels "
                 "(random numbers, copies of the peak-bandwidth probe) are excluded.",
       "csrc_sha": bench.csrc_hash()}
batches = {"cartpole": 4096, "linear": 4096, "chain5": 1024, "chain7": 1024}
for w, B in batches.items():
    f = os.path.join(ROOT, "gpurun_out", f"{w}_hbm.json")
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    keep = {k: v for k, v in d["kernels"].items() if ("small_" in k or "lq_solve" in k or "chain_" in k or "order_kernel" in k or "rocclr_fill" in k)}
    fetch = sum(v.get("FETCH_SIZE", 0.0) for v in keep.values())
    write = sum(v.get("WRITE_SIZE", 0.0) for v in keep.values())
    out[w] = {"batch": B, "command": d["command"], "fetch_bytes_per_step_raw": fetch, "write_bytes_per_step": write,
              "traffic_bytes_per_step": fetch + write, "kernels": keep}
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
json.dump(out, open(os.path.join(ROOT, "profiles", f"{TAG}_hbm_traffic.json"), "w"), indent=1)
print({k: (v["traffic_bytes_per_step"] if isinstance(v, dict) else v) for k, v in out.items() if k != "method"})
