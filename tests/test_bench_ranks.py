"""bench.py's launcher and rank plumbing on CPU: `python bench.py --gpus 2` with no launcher in the environment must start
two ranks itself (torch.distributed.run, 127.0.0.1 rendezvous), time the region as the MAX over ranks and print ONE JSON line
from rank 0 with n_gpus = the real world size.  The GPU step itself is replaced by a no-op (MPCRL_BENCH_DRYRUN)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(gpus, extra_env=None, workload=None, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MPCRL_BENCH_DRYRUN"] = "1"
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "5", "--warmup", "1"]
                         + (["--workload", workload] if workload else []) + list(extra_args),
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_the_slowest():
    line = _run(2)
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["dryrun"] is True
    assert line["ms_per_step"] >= 2.0          # rank 1 sleeps 2 ms per step: MAX over ranks, not rank 0's 1 ms
    assert line["rccl_ranks"] == 2             # the collective's own count of the ranks that joined (all-reduce of ones)


def test_every_workload_is_rank_aware():
    """BASELINE config 4 is "chain_mass ... 1 -> 8 GPU scaling" and config 5 shards its environments over 8 GPUs: `--workload chain5 |
    chain7 | td3 | linear --gpus 2` must go through the same launcher, rendezvous, barrier-bracketed MAX-over-ranks region and print
    ONE line with n_gpus = 2 (round 3's chain bench pinned cuda:0 and printed one line per rank)."""
    for wl in ("chain5", "td3", "linear"):
        line = _run(2, workload=wl)
        assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["workload"] == wl, line
        assert line["ms_per_step"] >= 2.0
        assert ("chain_mass n_mass=5" in line["metric"]) == (wl == "chain5")


def test_gpus_1_stays_in_process():
    line = _run(1)
    assert line["n_gpus"] == 1 and "rccl_ranks" not in line      # no communicator, no claim


def test_headline_line_carries_the_secondary_configurations():
    """The driver only runs `bench.py --gpus 1`: the chain (BASELINE config 4), linear-system (config 1) and TD3 (config 5) figures ride on that ONE
    line as `secondary`, measured by the same timed_steps AFTER the headline's timed region (its ms_per_step must not contain them);
    `--no-secondary` and every non-headline workload leave the field out."""
    import bench
    line = _run(1)
    sec = line["secondary"]
    assert set(sec) == {wl for wl, _, _ in bench.SECONDARY} == {"chain5", "chain7", "linear", "td3"}
    for wl, steps, warmup in bench.SECONDARY:
        assert sec[wl]["steps"] == steps and sec[wl]["warmup"] == warmup and sec[wl]["ms_per_step"] >= 1.0
    assert line["ms_per_step"] < 5.0                 # 1 ms sleeps: the ~90 secondary steps are not in the headline's region
    assert "secondary" not in _run(1, workload="linear") and "secondary" not in _run(2)
    assert "secondary" not in _run(1, extra_args=["--no-secondary"])


def test_traffic_figure_is_bound_to_the_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed counter pass: it must vanish when the kernels it was measured on have changed."""
    sys.path.insert(0, ROOT)
    import bench
    doc = {"csrc_sha": bench.csrc_hash(), "cartpole": {"batch": 4096, "traffic_bytes_per_step": 123.0}}
    f = tmp_path / "t.json"
    f.write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    assert bench.measured_traffic("cartpole", 4096, True, False)[0] == 123.0
    assert bench.measured_traffic("cartpole", 2048, True, False)[0] is None
    doc["csrc_sha"] = "0" * 16
    f.write_text(json.dumps(doc))
    assert bench.measured_traffic("cartpole", 4096, True, False)[0] is None


def test_spawn_command_is_the_drivers_launch_line():
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    cmd = bench.spawn_command(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "3"], 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_mismatched_launcher_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "launcher started 2" in (out.stderr + out.stdout)
