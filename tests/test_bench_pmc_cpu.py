"""bench.py's live counter passes, without a GPU: the kernel-name -> stage mapping, and live_pmc against a stand-in `rocprofv3` that writes the
counter CSVs a real pass writes (one row per dispatch and counter) — the averaging per launch, the gfx950 corrections, the fall-back reasons."""
import importlib.util
import os
import stat
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_stage_of_kernel(bench):
    s = bench.stage_of_kernel
    assert s("void vpt::k_whole<false, false, true>(vpt::DeviceScene, vpt::RenderParams)") == "primary"
    assert s("void vpt::k_whole<true, false, true>(vpt::DeviceScene)") is None            # the counting instantiation of kernel_profile's traversal pass
    assert s("void vpt::k_bounce<true, false, true, false, false, true>(x)") == "primary"  # FIRST
    assert s("void vpt::k_bounce<true, false, false, false, false, false>(x)") == "bounce"
    assert s("void vpt::k_bounce<true, true, false, false, false, false>(x)") is None      # COUNT
    assert s("void vpt::k_trace_vote<false, false, false, true, false, false, false, true>(a)") == "extend"
    assert s("void vpt::k_trace_shadow<true, false, true, false, true>(a)") == "shadow"
    assert s("void vpt::k_shade_stream<-1>(a)") == "shade" and s("vpt::k_join(a)") == "join" and s("vpt::k_resolve(a)") == "resolve"
    assert s("void vpt::k_finish<true>(a)") == "bounce" and s("vpt::k_refill_stream(a)") == "primary"
    assert s("vpt::k_bloom_down_chain(vpt::DownChain)") is None


FAKE = textwrap.dedent('''\
    #!/usr/bin/env python3
    import os, sys
    a = sys.argv[1:]
    d, tag = a[a.index("-d") + 1], a[a.index("-o") + 1]
    counters = a[a.index("--pmc") + 1:a.index("--output-format")]
    if os.environ.get("FAKE_FAIL") == tag:
        sys.exit(3)
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    val = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "SQ_ACTIVE_INST_VALU": 2.0e9, "SQ_THREAD_CYCLES_VALU": 96.0e9, "GRBM_GUI_ACTIVE": 8.0e7}
    with open(os.path.join(d, "host", tag + "_counter_collection.csv"), "w") as f:
        f.write("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\\n")
        for i in range(3):   # three launches of the headline kernel, one of another kernel and one counting instantiation: only the first kind counts
            for c in counters:
                f.write('%d,"void vpt::k_whole<false, false, true>(vpt::DeviceScene)",%s,%f\\n' % (i, c, val[c] * (1.0 + 0.1 * (i - 1))))
        for c in counters:
            f.write('7,"vpt::k_resolve(vpt::RenderParams)",%s,%f\\n' % (c, 7.0))
            f.write('8,"void vpt::k_whole<true, false, true>(vpt::DeviceScene)",%s,%f\\n' % (c, 9.0e12))
    ''')


@pytest.fixture()
def fake_rocprof(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in list(os.environ):
        if k.startswith("ROCPROF") or k.startswith("ROCP_"):
            monkeypatch.delenv(k)
    return exe


def test_live_pmc_averages_per_launch_and_applies_the_corrections(bench, fake_rocprof):
    entry, info = bench.live_pmc("cornell_1080p_d8", "primary", 0, 0)
    assert entry is not None, info
    assert entry["kernel_names"] == ["void vpt::k_whole<false, false, true>"] and info["launches_per_pass"] == 3
    # mean over the three launches: FETCH 1000 KiB, WRITE 500 KiB -> (2 x 1000 + 500) x 1024 bytes
    assert abs(entry["hbm_bytes_per_launch"] - 2500.0 * 1024.0) < 1.0 and abs(entry["fetch_bytes_raw"] - 1000.0 * 1024.0) < 1.0
    # VALU busy = 2e9 x 4 / (1024 x 8e7 / 8) = 0.78; lanes = 96e9 / 2e9 = 48
    assert entry["valu_busy"] == pytest.approx(0.781, abs=2e-3) and entry["valu_busy_raw"] == entry["valu_busy"] and entry["lanes_per_valu_instr"] == 48.0
    assert info["source"].startswith("live") and "corrections" in info


def test_live_pmc_reports_why_it_fell_back(bench, fake_rocprof, monkeypatch):
    monkeypatch.setenv("FAKE_FAIL", "write")
    entry, info = bench.live_pmc("cornell_1080p_d8", "primary", 0, 0)
    assert entry is None and "pass write" in info["error"]
    monkeypatch.delenv("FAKE_FAIL")
    entry, info = bench.live_pmc("atrium_1080p_d8", "shade", 0, 0)      # after a failed pass the other workloads do not try again (the line must go out within minutes)
    assert entry is None and "skipped after an earlier failure" in info["error"]
    monkeypatch.setattr(bench, "_LIVE_PMC_OFF", "")
    entry, info = bench.live_pmc("cornell_1080p_d8", "shade", 0, 0)     # no launch of that stage in the passes
    assert entry is None and "no shade launches" in info["error"]
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")                    # already under a profiler: never nest
    entry, info = bench.live_pmc("cornell_1080p_d8", "primary", 0, 0)
    assert entry is None and "profiler" in info["error"]


def test_roofline_prefers_the_live_entry_and_keeps_the_files_figures(bench):
    prof = {"kernels": {"primary": {"share": 0.98, "avg_ms": 218.0, "records_GBs": 272.8, "record_bytes_per_unit": 31.7, "units_per_launch": 1.0e9, "algorithmic_bytes_per_unit": 1662.0,
                                    "algorithmic_GBs": 14000.0, "algorithmic_frac_of_hbm_peak": 1.75}, "resolve": {"share": 0.02, "avg_ms": 4.8, "records_GBs": 1.0}},
            "bvh": {"node_bytes": 128}, "traversal": {}, "pipeline": "whole paths"}
    live = ({"hbm_bytes_per_launch": 4.8e10, "fetch_bytes_raw": 7e9, "write_bytes": 3.4e10, "valu_busy": 1.0, "valu_busy_raw": 1.2, "lanes_per_valu_instr": 47.3, "kernel_names": ["k"]},
            {"source": "live: test", "launches_per_pass": 5, "seconds": 1.0, "corrections": "x"})
    r = bench.roofline_for("cornell_1080p_d8", prof, live)
    assert r["traffic"] == 4.8e10 and r["valu_busy"] == 1.0 and r["bound"] == "valu" and r["pmc"]["source"] == "live: test" and "committed_file" in r["pmc"]
    assert r["valu"]["active_lanes_of_64"] == 47.3 and 0 < r["valu"]["frac"] <= 1.0
    failed = (None, {"error": "rocprofv3 not found"})
    r2 = bench.roofline_for("cornell_1080p_d8", prof, failed)
    assert r2["pmc"]["source"] == "profiles/traffic.json" and r2["pmc"]["live_failed"]["error"] == "rocprofv3 not found"
