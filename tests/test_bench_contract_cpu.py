"""The shape of bench.py's one JSON line, checked on the line the final build printed on the GPU box
(profiles/r5_bench.json): the driver's contract fields, the roofline and cpu_baseline objects, and
the arithmetic that ties them together.  (bench.py itself needs a GPU: tests/test_abi.py checks that
it refuses to run without one.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r5_bench.json")) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints ONE line on stdout"
    return json.loads(lines[0])


def test_driver_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frames of all ranks per second of the timed region
    frames = d["n_gpus"] * d["config"]["frames_per_gpu"] * d["steps"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the kernel's average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["algorithmic_bytes_per_launch"] == (d["config"]["frames_per_gpu"] - 1) * 1080 * 1920
    # measured traffic (committed counter pass) is what the kernel needs and no more
    assert r["traffic"] is None or 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert "traffic_source" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["agrees_with_gpu"] is True and c["single_core"]["cores"] == 1


def test_both_halves_of_the_metric_are_in_the_objects_the_driver_keeps():
    """roofline.kernels carries the luma-SAD kernel and both lookups (full corpus, 1/8 shard) with what a
    reader needs to recompute their fractions; cpu_baseline.matcher the two CPU restatements of find_duplicates; the
    headline fraction uses the same time base as ms_per_step."""
    d = _line()
    r = d["roofline"]
    assert abs(r["avg_launch_ms"] - d["ms_per_step"]) < 1e-12
    ks = r["kernels"]
    names = [k["name"] for k in ks]
    assert names[0] == "luma_sad_flat_kernel" and names.count("ts_match_index_topk_kernel") == 2 and "ts_match_wq_topk_kernel" in names
    for k in ks:
        if "algorithmic_bytes" in k:
            assert abs(k["frac"] - k["algorithmic_bytes"] / (k["avg_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-9, k["name"]
        if k["name"].startswith("ts_match"):
            assert k["pairs_per_s"] > 0 and k["frac_bound"]["resource"] in ("valu_issue", "lds", "fabric_random_lines")
            assert 0 < k["frac_bound"]["frac"] < 1 and set(k["frac_bound"]["bounds"]) == {"valu_issue", "lds", "fabric_random_lines"}
    m = d["cpu_baseline"]["matcher"]
    assert m["python_restatement"]["cores"] == 1 and m["c_sorted_binary_search"]["cores"] >= 1
    assert m["python_restatement"]["unit"] == m["c_sorted_binary_search"]["unit"] == "pairs/s"
    assert d["match"]["rccl_ranks"] == d["n_gpus"] == 1


def test_e2e_counts_scored_frames_and_stays_under_the_link():
    """e2e.value = frames the scene kernels scored (an upload stops at its duplicate verdict,
    inspector/app.py:249-255), never uploads x frames; and what was scored crossed the link: GBps_luma <= h2d.GBps x the
    copies in flight."""
    d = _line()
    for e in [d["e2e"]] + d["e2e"].get("other_shapes", []):
        assert e["frames_scored"] <= e["frames_submitted"] == e["uploads"] * e["frames_per_upload"]
        assert abs(e["value"] - e["frames_scored"] * e["submitted_frames_per_s"] / e["frames_submitted"]) / e["value"] < 1e-6
        assert e["GBps_luma"] <= d["h2d"]["GBps"] * d["e2e"]["h2d_bound_check"]["copies_in_flight"] * 1.02
    assert d["e2e"]["h2d_bound_check"]["ok"] is True
    assert [e["uploads"] for e in d["e2e"]["other_shapes"]] == [64]                 # configs[4]'s shape


def test_secondary_objects():
    d = _line()
    m = d["match"]
    assert m["unit"] == "pairs/s" and m["corpus_videos"] == 100000 and m["queries_per_batch"] == 4096
    assert abs(m["value"] - m["corpus_videos"] * m["queries_per_batch"] / (m["ms_per_batch"] * 1e-3)) / m["value"] < 1e-6
    assert m["roofline"]["bound"] == "hbm" and 0 < m["roofline"]["frac"] < 1 and m["queries_with_overflowed_shard_lists"] == 0
    assert "shard8_roofline" in m and "config2" in m
    assert d["config0"]["gpu"]["cuts_equal_cpu"] is True
    assert d["e2e"]["all_done"] is True and d["h2d"]["GBps"] > 0


# ---- the self-launcher and the watchdog's return code (no GPU: `--launch-stub` ranks join a gloo group) ----
import subprocess
import sys

BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_without_a_launcher_starts_n_fresh_rank_processes():
    """`python bench.py --gpus N` as the driver runs `--gpus 1`: the parent touches no GPU,
    starts N rank processes of bench.py itself and hands them the rendezvous in the environment."""
    p = _run("--gpus", "2", "--steps", "3", "--dry-launch")
    assert p.returncode == 0, p.stderr
    d = json.loads(p.stdout)
    assert d["ranks"] == 2 and d["launch"][1] == BENCH and d["launch"][2:] == ["--gpus", "2", "--steps", "3"]
    assert d["env"]["WORLD_SIZE"] == "2" and d["env"]["MASTER_ADDR"] == "127.0.0.1" and int(d["env"]["MASTER_PORT"]) > 0


def test_launcher_relays_rank0_line_and_the_worst_return_code():
    p = _run("--gpus", "2", "--launch-stub")
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"stub": True, "n_gpus": 2, "sum_of_rank_numbers": 3.0}
    p = _run("--gpus", "2", "--launch-stub", env={"TVZ_BENCH_STUB_FAIL_RANK": "1"})
    assert p.returncode == 7, (p.returncode, p.stderr)
    assert p.stdout.strip() == ""                       # no line from a failed run


def test_a_mismatched_world_size_is_refused():
    p = _run("--gpus", "2", "--launch-stub", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


def test_an_abandoned_leg_exits_non_zero_and_names_the_rank():
    """bench.py:run_under_watchdog: the line is printed with e2e.error and the
    ranks that never finished; every rank leaves with return code 3."""
    p = _run("--gpus", "2", "--launch-stub", "--stub-hang-e2e", "--e2e-timeout", "1")
    assert p.returncode == 3, (p.returncode, p.stderr)
    d = json.loads(p.stdout)
    assert "did not finish" in d["e2e"]["error"] and d["e2e"]["ranks_that_never_finished"] == [1]
    p = _run("--gpus", "1", "--launch-stub", "--stub-hang-e2e", "--e2e-timeout", "0")
    assert p.returncode == 3 and "error" in json.loads(p.stdout)["e2e"]
