"""The driver's own bench command must produce its line (round 2's died of a GPU page fault inside a secondary leg and lost every number
of the round).  Runs bench.py in a child process exactly as the driver does and checks the contract fields."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=900):
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    return r, lines


def test_the_drivers_bench_command_prints_its_line(gpu_required):
    r, lines = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["steps"] == 20 and out["warmup"] == 5 and out["n_gpus"] == 1 and out["unit"] == "frames/s" and out["value"] > 0
    assert abs(out["ms_per_step"] * out["value"] - 1e3) < 1e-6 * 1e3
    for leg in ("host_buffers", "pose_latency_unpipelined", "multi_stream", "update_at_load", "batched_streams", "batched_filter"):
        assert leg in out and "error" not in out[leg], (leg, out.get(leg))
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["achieved"] > 0 and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    hot = [o for o in out["roofline_other"] if "section 8(f)" not in o["kernel"]]
    assert "traffic" in rf and rf["avg_us"] > 0
    # the identity: the hot-path kernel with the largest average in the committed kernel trace of this configuration (round 6: the same kernel in every invocation) —
    # or, for a configuration without a committed trace, the slowest hot-path kernel timed live
    if "trace_avg_us" in rf:
        assert all("trace_avg_us" in o for o in hot) and rf["trace_avg_us"] >= max(o["trace_avg_us"] for o in hot)
        assert "klt_kernel3" in rf["kernel"]                                      # (what profiles/r06_kernel_stats_stream.md says for the stock configuration)
    else:
        assert rf["avg_us"] >= max(o["avg_us"] for o in hot)
    cb = out["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    par = out["parity"]
    assert par["frames"] >= 141 and par["max_state_delta"] <= 1e-6 and par["counters_equal_every_frame"] and par["rank_truncation_nrank_equal"]
    assert out["pose_latency_unpipelined"]["frames"] >= 8 and out["pose_latency_unpipelined"]["updated_last_frame"] == 1
    assert out["last_frame"]["device_error"] == 0


@pytest.mark.parametrize("steps,warmup", [(1, 0), (60, 2)])
def test_short_and_odd_step_counts_do_not_kill_the_line(gpu_required, steps, warmup):
    """(steps, warmup) the pose-latency plan cannot serve from the main sequence: the leg moves to the parity sequence instead of dying"""
    r, lines = _run(["--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu", "--batch", "", "--batch-streams", "", "--no-streams"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(lines[-1])
    assert out["steps"] == steps and "error" not in out["pose_latency_unpipelined"], out["pose_latency_unpipelined"]


def test_forced_sharded_line_equals_the_single_gpu_updater(gpu_required):
    """bench.py --force-sharded: the sharded frame (rvio_hip_frame_sharded_dev: update_local -> the REAL ncclAllGather on the filter stream ->
    update_global, one C-ABI call per frame) at world 1 against the same frames through the unsharded updater: block sums in rank order, so
    the two agree to rounding — asserted, not just printed (it was null in round 2)."""
    r, lines = _run(["--gpus", "1", "--steps", "60", "--warmup", "5", "--force-sharded", "--no-cpu", "--batch", "", "--batch-streams", "", "--no-streams"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(lines[-1])
    assert "ncclAllGather" in out["config"]["parallelism"], out["config"]["parallelism"]
    assert out["last_frame"]["updated"] == 1 and out["last_frame"]["device_error"] == 0
    assert out["max_state_delta_sharded_vs_single_gpu"] is not None and out["max_state_delta_sharded_vs_single_gpu"] <= 1e-9, out["max_state_delta_sharded_vs_single_gpu"]
