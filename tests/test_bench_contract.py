"""The bench line's contract, checked on bench.py's LIVE output: the test runs `python bench.py` in a child process (one small step
of configs[2]'s scene: same code path, reduced tessellation / resolution so that it takes seconds) and asserts on the one JSON line
it prints — keys and types of the driver's contract, BASELINE.json's metric string, value = rays / time, stage times that sum to
the step, a `roofline` object whose `frac` is achieved / peak against the HBM peak and whose other fractions each name their
denominator (`ceiling_source` / `denominator`), counters that are either derived from the build that is running or null and
flagged stale, and the `cpu_baseline` object.  (Round 3's version of this file read committed JSON files: it stayed green when
bench.py broke.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, timeout=600):
    env = dict(os.environ)
    env.pop("MIPT_LIB_OVERRIDE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line, got %d" % len(lines)
    return json.loads(lines[0])


SMALL = ("--steps", "2", "--warmup", "1", "--grid", "160", "--width", "480", "--height", "270", "--spp-per-step", "32")


@pytest.fixture(scope="module")
def line():
    return run_bench(*SMALL)


@pytest.mark.gpu
def test_live_bench_line_keeps_the_driver_contract(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["metric"] == base["metric"] and line["unit"] == "Mrays/s" and line["higher_is_better"] is True
    assert line["vs_baseline"] is None                         # the reference publishes no number for this metric
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["finite"] is True
    assert line["scaling"] in ("weak", "strong")
    ls = line["launch_stats"]
    rays = ls["rays_closest"] + ls["rays_shadow"]
    assert rays > 0 and rays / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6 == pytest.approx(line["value"], rel=1e-6)
    st = line["stage_ms_per_step"]
    assert set(st) == {"extend", "shadow", "generate+shade", "resolve"}
    assert 0.0 < sum(st.values()) <= line["ms_per_step"] * 1.001    # nothing is counted twice (at this size the host side of a step outweighs its kernels)


@pytest.mark.gpu
def test_live_roofline_names_every_denominator(line):
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # the headline fraction's definition is frozen: the text below is the contract
    assert r["frac_definition"] == ("SURVEY 8(d): algorithmic bytes per launch (24 n_box + 8 n_node + 64 n_tri from the oracle's counters of the reference's ordered traversal, "
                                    "x the rays of one launch) / mean launch time (HIP events on the render stream) / HBM peak 8 TB/s (MI355X_MICROARCH.md)")
    assert "cache" in r["frac_note"]
    # achieved = algorithmic bytes per launch / launch time, and the dominant kernel's launches fit inside the step
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e9, rel=1e-9)
    assert r["ms_per_launch"] * r["launches"] / line["steps"] == pytest.approx(line["stage_ms_per_step"]["extend"], rel=0.02)
    assert r["algorithmic_bytes_per_launch"] == pytest.approx(r["bytes_per_closest_ray"] * r["rays_per_launch"], rel=1e-6)
    # one account of the kernel's time: no second or third fraction beside issue_model; what the run measures on the device is raw rates
    assert "ceilings_measured_in_this_run" not in r and "latency_model" not in r and "frac_l1_lookups" not in r
    dr = r["device_rates_measured_in_this_run"]
    assert dr["source"].startswith("measured in this run by mipt_measure_") and not any("frac" in k for k in dr)
    # what comes from the committed PMC run is either of THIS build, or absent and flagged
    d = r.get("derived_from_pmc_run")
    if d is None:
        assert "pmc_note" in r and r["traffic"] is None
    elif d["same_library_build"]:
        assert d["stale"] is False and r["traffic"] > 0 and 0.0 < r["frac_hbm_measured"] <= 1.0
        assert r["frac_hbm_measured"] == pytest.approx(r["traffic"] / (r["ms_per_launch"] * 1e-3) / 8e12, rel=1e-9)
        assert "denominator" in r["issue_model"] and 0.0 < r["issue_model"]["vmem_issue_busy"] <= 1.2
    else:
        assert d["stale"] is True and r["traffic"] is None and r["frac_hbm_measured"] is None
        assert "issue_model" not in r
    assert "sha256" in (d or {"keyed_on": "sha256"})["keyed_on"]
    sh = line["roofline_shade_kernel"]
    assert sh["frac"] == pytest.approx(sh["achieved"] / sh["peak"], rel=1e-9) and "frac_definition" in sh
    # no fraction that claims to be MEASURED HBM traffic may exceed what this device streams (the line once said 1.16)
    ceiling = (r["peak_measured_stream_read"] or 8000.0) / 8000.0
    for obj in (r, line.get("roofline_shadow_kernel") or {}, sh):
        f = obj.get("frac_hbm_measured")
        assert f is None or 0.0 < f <= ceiling, (obj.get("kernel"), f, ceiling)


def test_committed_counters_charge_every_shade_build_with_its_own_launches():
    """profiles/pmc_counters.json: the depth-0 builds of the shade tiers are named for what they are and launch once per pass; the stage's bytes
    per vertex follow from per-build launches (2 passes per step at 1080p: 2 + 2 x (nb_bounces - 1) launches of a tier)."""
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    for wl, depth in (("c2", 4), ("c1", 4), ("c3", 12), ("c4", 4)):
        ks = j[wl]["kernels"]
        assert not any("[quad]" in k for k in ks), "the second template argument is INITIAL (depth 0), not a quad build"
        assert ks["k_wf_shade<1>[depth0]"]["launches_per_step"] == 2 and ks["k_wf_generate"]["launches_per_step"] == 2
        assert ks["k_wf_shade<1>"]["launches_per_step"] == 2 * (depth - 1)
        st = j[wl]["stage_generate_shade"]
        total = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for k, v in ks.items() if k.startswith(("k_wf_shade", "k_wf_generate", "k_wf_merl_eval")))
        assert st["hbm_bytes_per_step"] == pytest.approx(total, rel=1e-12)
        assert 100.0 < st["hbm_bytes_per_vertex"] < 1500.0


@pytest.mark.gpu
def test_live_cpu_baseline_object(line):
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["unit"] == "Mrays/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert line["value"] > cb["value"]                    # context, not credit: the CPU path timed beside the GPU one


@pytest.mark.gpu
def test_two_gpus_in_a_child_process_where_two_exist():
    """bench.py --gpus 2 from the plain command line: ONE process, mipt_create(ids, 2), RCCL ncclReduce or a non-zero exit."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    out = run_bench("--gpus", "2", "--no-cpu-baseline", *SMALL)
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_reduce"] == 2
    assert out["config"]["reduce"].startswith("RCCL"), out["config"]["reduce"]


def test_bench_refuses_to_run_without_a_gpu_or_with_fewer_than_asked():
    """No CPU fallback: without a device (this container) or with fewer devices than --gpus, bench.py exits non-zero."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to measure fewer GPUs" in (r.stderr + r.stdout)
    # a run that fails still leaves ONE JSON line: nothing measured (value null), where it stopped, how the framebuffers would have been reduced
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["stage"] == "devices" and "refusing" in d["error"] and d["n_gpus"] == n + 1 and "stages_in_order" in d


def test_source_hash_is_stable_and_sees_the_flags():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    h = ge.source_hash()
    assert h == ge.source_hash() and len(h) == 16
    saved = list(ge.HIPCC_FLAGS)
    try:
        ge.HIPCC_FLAGS.append("-DSOMETHING")
        assert ge.source_hash() != h
    finally:
        ge.HIPCC_FLAGS[:] = saved
