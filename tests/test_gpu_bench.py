"""bench.py's launch contract: `python bench.py --gpus N` alone must run N ranks (or fail loudly), at any N, with the workloads
BASELINE.json names.  Run as subprocesses with NO launcher environment, the way the driver starts it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")


def run_bench(args, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS and k != "VPT_BENCH_DEVICE"}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


QUICK = ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra-workloads", "--no-latency", "--no-live-pmc", "--frames-in-flight", "2"]


def test_gpus_2_without_a_launcher_runs_two_ranks():
    """Two ranks on device 0 through the one-device hook: the script launches itself under torch.distributed.run."""
    r, line = run_bench(["--gpus", "2"] + QUICK, {"VPT_BENCH_DEVICE": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["value"] > 0
    assert line["rccl"]["nranks"] == 2
    assert line["config"]["workload"] == "cornell_1080p_d8" and line["config"]["frames_per_step_per_gpu"] == 2
    assert line["roofline"]["frac"] > 0


def test_gpus_2_with_one_visible_device_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    r, line = run_bench(["--gpus", "2"] + QUICK)
    assert r.returncode != 0 and line is None
    assert "device" in r.stderr


def test_strong_scaling_and_workloads_block_at_n_2():
    """The extra workloads run sharded on every rank (weak and strong) and the line keeps the block at N > 1."""
    r, line = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--frames-in-flight", "2", "--scaling", "strong", "--extra-steps", "1"],
                        {"VPT_BENCH_DEVICE": "0"}, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    wl = line["workloads"]
    assert set(wl) == {"atrium_1080p_d8", "atrium_4k_d8", "glass_bust_1080p_d32"}
    for name, w in wl.items():
        assert w["value"] > 0 and w["strong"]["value"] > 0 and w["rccl"]["nranks"] == 2, name
    assert wl["atrium_4k_d8"]["width"] == 3840 and wl["atrium_4k_d8"]["height"] == 2160
    assert wl["glass_bust_1080p_d32"]["post_ms_in_timed_region"] > 0


def test_the_4k_and_post_workloads_as_headline():
    r, line = run_bench(["--workload", "atrium_4k_d8"] + QUICK)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["metric"] == "Msamples/s at 3840x2160" and line["config"]["width"] == 3840 and line["n_gpus"] == 1
    assert line["config"]["pipeline"] == "staged (streams)" and line["set_scene"]["bvh_build_ms"] > 0
    assert_physical_kernel_table(line)
    r, line = run_bench(["--workload", "glass_bust_1080p_d32"] + QUICK)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["config"]["post_in_timed_region"] is True and line["post_ms_in_timed_region"] > 0
    # 2-frame batches at depth 32: the streams hand the rest of a batch to the one-launch finisher (k_finish, timed under "bounce"), which must be priced
    # with ITS units — paths taken over, path-bounces run — not with the fused per-bounce kernels' (round 5: 138 x the HBM peak in this row)
    assert "bounce" in line["roofline"]["kernels"] and line["roofline"]["kernels"]["bounce"]["units_per_launch"] > 0
    assert_physical_kernel_table(line)


def assert_physical_kernel_table(line):
    """No row of the per-kernel table may claim more bytes across HBM than HBM can move: records_GBs counts only what MUST cross (path records, queue words, frame sums)."""
    for name, k in line["roofline"]["kernels"].items():
        assert 0 < k["records_GBs"] <= 8000.0, (name, k)


def test_headline_counters_are_measured_by_the_run_itself():
    """roofline.traffic / valu_busy at N = 1: bench.py spawns its own rocprofv3 --pmc passes (live_pmc) instead of copying profiles/traffic.json;
    the committed file's figures stay beside them."""
    import shutil
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    r, line = run_bench(["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra-workloads", "--no-latency", "--frames-in-flight", "16"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    roof = line["roofline"]
    assert roof["pmc"]["source"].startswith("live"), roof["pmc"]
    assert roof["kernel"] == "primary" and any("k_whole" in k for k in roof["pmc"]["kernel_names"])
    assert roof["traffic"] > 0 and 0.0 < roof["valu_busy"] <= 1.0
    # the whole-path launch writes a 16-byte frame sum per sample and parks a share of the later hits: within a factor of the records it must move
    assert 0.5 < roof["traffic"] / roof["record_bytes_per_launch"] < 8.0, (roof["traffic"], roof["record_bytes_per_launch"])
    assert 32.0 < roof["valu"]["active_lanes_of_64"] <= 64.0
    assert "committed_file" in roof["pmc"]
