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


def test_kernel_table_prices_the_finisher_with_its_own_units(bench):
    """A streams batch whose tail ran in the one-launch finisher (k_finish, timed under "bounce"): round 5's table took the finisher's launches for fused
    bounces and priced the raygen launch and the finisher with the fused kernels' units — 138 x the HBM peak on the driver's glass-bust block.  The finisher's
    unit is a path-bounce it ran, its records cross HBM once per path taken over, and the stream stages count only the rays they traced."""
    K = ("primary", "extend", "shade", "connect", "bounce", "resolve", "bloom", "tonemap", "shadow", "join")
    launches = dict.fromkeys(K, 0); ms = dict.fromkeys(K, 0.0)
    launches.update(primary=25, extend=24, shade=24, shadow=48, join=24, bounce=1, resolve=4)
    ms.update(primary=20.0, extend=140.0, shade=150.0, shadow=60.0, join=40.0, bounce=1.75, resolve=4.0)
    st = {"samples": 1_874_534_400, "closest_rays": 6_085_799_731, "shadow_rays": 2_031_925_852, "connect_paths": 1_900_000_000, "primary_hits": 0, "primary_survivors": 0, "primary_shadow_rays": 0,
          "finish_paths": 41_893, "finish_closest_rays": 294_015, "finish_shadow_rays": 98_000, "bvh_node_bytes": 64, "frames_in_flight": 904, "kernel_launches": launches, "kernel_ms": ms}
    tc = {"nodes_per_closest_ray": 9.0, "tris_per_closest_ray": 2.0, "nodes_per_shadow_ray": 10.0, "tris_per_shadow_ray": 3.0}
    t = bench.kernel_table(st, tc)
    assert set(t) == {"primary", "extend", "shade", "shadow", "join", "bounce", "resolve"}
    assert t["bounce"]["units_per_launch"] == 294_015 and t["bounce"]["record_bytes_per_unit"] < 100.0
    assert t["primary"]["record_bytes_per_unit"] == 68.0                      # the raygen launch of the streams, not the fused bounce 0
    assert abs(t["extend"]["units_per_launch"] - (6_085_799_731 - 294_015) / 24) < 1.0
    assert all(0 < k["records_GBs"] <= 8000.0 for k in t.values()), {n: k["records_GBs"] for n, k in t.items()}
