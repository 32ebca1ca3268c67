"""bench.py's output contract, checked on the CPU: the helper that sizes the CPU-baseline sample,
the PMC-traffic lookup, and the schema of the last committed bench line (profiles/)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_cpu_sample_sizing_fills_the_budget_within_the_caps():
    calls = []

    def run(nt):
        calls.append(nt)
        return 0.001 * nt                      # 1 ms per step

    nt, reps, el = bench.sized_cpu_sample(run, 100, 2000, 15.0, bytes_per_step=1e6, mem_cap=5e8)
    assert nt == 500 and calls[0] == 100       # memory cap: 5e8 / 1e6 steps
    assert 25 <= reps <= 35 and abs(el - reps * 0.5) < 1e-9
    nt, reps, el = bench.sized_cpu_sample(run, 100, 300, 0.05, bytes_per_step=1.0)
    assert nt == 100 and reps == 1             # calibration already exceeds the budget


def _fake_tree(tmp_path, monkeypatch, body=b"kernel v1"):
    """A throw-away repo root with one kernel source, so that the fingerprint logic can be driven."""
    (tmp_path / "physicsbasedfwi2_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "physicsbasedfwi2_amd" / "csrc" / "k.hip").write_bytes(body)
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    return tmp_path


def test_profile_summaries_are_quoted_only_for_the_kernels_they_describe(tmp_path, monkeypatch):
    """roofline.traffic and the latency floor come from committed rocprofv3 / ablation summaries that carry the
    fingerprint of the kernel sources; once a kernel file changes they are dropped, never quoted stale."""
    root = _fake_tree(tmp_path, monkeypatch)
    sha = bench.csrc_sha16()
    doc = {"csrc_sha16": sha, "commit": "abc1234",
           "elastic_100x300": {"adjoint+imaging": {"bytes_per_cell_step": 42.5}}}
    (root / "profiles" / (bench.PROFILE_ROUND + "_pmc_traffic.json")).write_text(json.dumps(doc))
    floor = {"csrc_sha16": sha, "commit": "abc1234",
             "elastic_100x300": {"adjoint+imaging": {"floor_s_per_step": 9.3e-6}}}
    (root / "profiles" / (bench.PROFILE_ROUND + "_latency_floor.json")).write_text(json.dumps(floor))
    assert bench.measured_traffic("elastic_100x300", "adjoint+imaging") == (42.5, "csrc " + sha)
    assert bench.latency_floor("elastic_100x300", "adjoint+imaging") == (9.3e-6, "csrc " + sha)
    assert bench.measured_traffic("no_such_workload", "x")[0] is None
    (root / "physicsbasedfwi2_amd" / "csrc" / "k.hip").write_bytes(b"kernel v2")
    assert bench.csrc_sha16() != sha
    val, why = bench.measured_traffic("elastic_100x300", "adjoint+imaging")
    assert val is None and "other kernels" in why
    assert bench.latency_floor("elastic_100x300", "adjoint+imaging")[0] is None


def test_roofline_objects_use_one_cell_convention_and_never_exceed_one(tmp_path, monkeypatch):
    root = _fake_tree(tmp_path, monkeypatch)
    sha = bench.csrc_sha16()
    (root / "profiles" / (bench.PROFILE_ROUND + "_pmc_traffic.json")).write_text(json.dumps(
        {"csrc_sha16": sha, "commit": "c0ffee0", "elastic_100x300": {
            "adjoint+imaging": {"bytes_per_cell_step": 42.5}, "forward+save": {"bytes_per_cell_step": 36.5}}}))
    (root / "profiles" / (bench.PROFILE_ROUND + "_latency_floor.json")).write_text(json.dumps(
        {"csrc_sha16": sha, "commit": "c0ffee0", "elastic_100x300": {
            "adjoint+imaging": {"floor_s_per_step": 9.3e-6}, "forward+save": {"floor_s_per_step": 6.2e-6}}}))
    interior = 100 * 300 * 32
    kern = {"forward+save": bench.kernel_report("elastic_100x300", "forward+save", 7.6e-6, interior, 60.0, 20.0, True),
            "adjoint+imaging": bench.kernel_report("elastic_100x300", "adjoint+imaging", 11.7e-6, interior, 80.0, 20.0,
                                                   True)}
    r = bench.roofline_of(kern, "adjoint+imaging", interior)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["cells"] == "interior"
    assert abs(r["achieved"] - 20.0 * interior / 11.7e-6 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["frac"] < 1.0 and r["traffic"] == 42.5 * interior and r["traffic_profiled_at"] == "csrc " + sha
    assert abs(r["streaming_equivalent_GBs"] - 80.0 * interior / 11.7e-6 / 1e9) < 1e-6
    lat = r["latency"]
    assert lat["bound"] == "latency" and abs(lat["frac"] - 9.3 / 11.7) < 1e-9 and lat["floor_profiled_at"] == "csrc " + sha
    # per-step streaming family: SURVEY 8d bytes, no latency object
    k2 = {"adjoint+imaging": bench.kernel_report("elastic_350x1700", "adjoint+imaging", 323e-6, 350 * 1700 * 32, 80.0,
                                                 20.0, False)}
    r2 = bench.roofline_of(k2, "adjoint+imaging", 350 * 1700 * 32)
    assert "latency" not in r2 and r2["traffic"] is None and 0.55 < r2["frac"] < 0.62
    assert k2["adjoint+imaging"]["traffic_note"].startswith("workload not in")


def _committed(suffix):
    """Newest committed profiles/rNN_<suffix> (this round's once it has been measured, else the round before)."""
    rounds = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_" + suffix)), reverse=True)
    assert rounds, suffix
    return os.path.join(ROOT, "profiles", rounds[0])


def _committed_full_report():
    """The full report of the last measured default run: rNN_bench_default_detail.json (round 4 on), or round 3's
    rNN_bench_default_output.json, which had the same layout (it was printed whole, and the driver lost it)."""
    det = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_default_detail.json"))
    path = os.path.join(ROOT, "profiles", det[-1]) if det else os.path.join(ROOT, "profiles", "r03_bench_default_output.json")
    with open(path) as fh:
        return json.load(fh)


def test_final_stdout_line_is_one_bounded_json_object():
    """Round 3's line grew to 23 KB and the driver parsed nothing.  The line bench.py prints LAST is formatted here from
    a committed full report: one JSON object, <= 6000 bytes, carrying the contract fields with roofline.frac and
    cpu_baseline.value - whatever the secondary workloads add."""
    d = _committed_full_report()
    also = d.pop("also", [])
    for a in also:                                  # the full report keeps these per entry already
        assert "config" in a and "kernels" in a and "roofline" in a
    text = bench.final_line(d, also)
    assert "\n" not in text and len(text.encode()) <= bench.MAX_LINE_BYTES <= 6000
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "check", "ranks"):
        assert k in line, k
    assert line["config"]["workload"] == bench.ElasticMarmousi.name and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
    assert "traffic" in r and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert abs(line["value"] - d["value"]) <= 1e-3 * d["value"]
    assert len(line.get("also", [])) == len(also)
    for a in line.get("also", []):
        assert a["roofline"]["frac"] > 0 and a["value"] > 0 and set(a["kernels"]) == {"forward+save", "adjoint+imaging"}
    # a report that would not fit loses its extras, never the headline
    fat = [dict(a, config=dict(a["config"], workload="w" * 1500)) for a in also] * 2
    text = bench.final_line(d, fat)
    assert len(text.encode()) <= bench.MAX_LINE_BYTES
    assert "also" not in json.loads(text) and json.loads(text)["roofline"]["frac"] == r["frac"]


def test_committed_compact_line_is_what_the_driver_can_parse():
    """profiles/rNN_bench_default_output.json from round 4 on is the very line the run printed."""
    path = _committed("bench_default_output.json")
    raw = open(path).read().strip()
    d = json.loads(raw)
    if "also" in d and d["also"] and "config" in d["also"][0]:
        return                                      # round 3's full-format file: covered by the test above
    assert len(raw.encode()) <= bench.MAX_LINE_BYTES and "\n" not in raw
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["check"]["verified"] is True


def test_committed_bench_line_has_the_contract_fields():
    d = _committed_full_report()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "check"):
        assert k in d, k
    assert d["config"]["workload"] == bench.ElasticMarmousi.name and "model" not in d["config"]
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    entries = [d] + d.get("also", [])
    assert entries[1]["config"]["workload"].startswith(bench.AcousticMarmousi.name)
    assert any(e["config"]["grid"] == [1000, 3000] for e in entries[1:])          # the SEAM-sized sample (C5's grid)
    for e in entries:
        r = e["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert 0 < r["frac"] <= 1.0 and r["cells"] == "interior"
        assert r["traffic"] is None or (r["traffic"] > 0 and r["traffic_profiled_at"])
        assert e["check"]["bitwise_repeatable"] is True and e["check"]["verified"] is True
        assert e["check"].get("fallbacks", 0) == 0           # no single-launch time loop gave up inside the timed region
        assert e["check"]["loss"] > 1e-8 and e["check"]["grad_abs_sum"] > 1e-8
        if "cpu_baseline" in e:
            c = e["cpu_baseline"]
            assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "C oracle" in c["sample"]
        for k in e["kernels"].values():
            if k["lds_resident"]:
                # diagnostic quoted for an LDS-resident loop: issue cycles of the shipped kernel over the step's cycles
                assert k["issue"]["bound"] == "issue"
                assert k["issue"]["frac_simd_valu"] is None or 0.05 < k["issue"]["frac_simd_valu"] <= 1.0
                assert k["latency"]["bound"] == "latency"
    assert "cpu_baseline" in entries[0] and "cpu_baseline" in entries[1]
    # whole-job throughput = interior cells*steps of all shots / wall time
    wl = bench.ElasticMarmousi
    units = wl.nz * wl.nx * wl.nt * wl.shots_per_gpu * d["steps"]
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) <= 1e-6 * d["value"]


def test_pmc_tool_knows_the_kernels_the_bench_runs():
    """tools/pmc_traffic.py maps kernel names to bench labels by regular expression: every time-loop kernel of
    this round's committed kernel-trace summaries must fall under exactly one label of its physics, set-up
    kernels under none."""
    import csv
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    seen = {}
    last_round = os.path.basename(_committed("bench_default_kernel_stats.csv"))[:3]
    for fname in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if fname.startswith(last_round) and fname.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fname))):
                seen[pmc_traffic.clean(r["Name"])] = fname
    assert any(n.startswith("el_cluster_adj<") for n in seen) and any(n.startswith("ac_cluster<2,") for n in seen)
    assert any(n.startswith(("el_fwd_fused<", "el_step_v<")) for n in seen) and any(n.startswith("el_adj_s<") for n in seen)
    loop_re = (r"(ac_cluster<[12],|ac_step<\d+, \d+, (true, false|false, true)>|el_cluster_fwd<true|el_cluster_adj<|"
               r"el_step_[vs]<\d+, \d+, [12]>|el_adj_s<|el_adj_v$|el_fwd_fused<[12]>|el_adj_fused<|el_adj_walk<|el_inject_adjsrc$)")
    for n in seen:
        phys = "acoustic" if n.startswith("ac_") else "elastic" if n.startswith("el_") else None
        if phys is None:
            continue
        hits = [lab for lab, pats in pmc_traffic.LABELS[phys].items() if any(re.search(p, n) for p in pats)]
        assert len(hits) == (1 if re.match(loop_re, n) else 0), (n, hits)


def test_gpus_without_a_launcher_starts_the_ranks_and_fails_loudly_when_one_fails():
    """`bench.py --gpus 2` with WORLD_SIZE unset spawns two ranks before touching the GPU; here (no GPU) every rank
    exits with an error and the parent must report it with a non-zero exit code instead of printing a line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--device-index", "0", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                         text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        assert res.returncode == 0 and json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 2
    else:
        assert res.returncode != 0 and "exited with code" in res.stderr and "n_gpus" not in res.stdout
    # a launcher-provided world that disagrees with --gpus is an error, not a silent one-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in res.stderr


def test_xcd_tile_order_is_a_bijection_with_contiguous_runs():
    """The per-step kernels remap blockIdx so that neighbouring tiles share an XCD's L2 (csrc: xcd_tile; launch
    order modulo 8 = XCD).  Same arithmetic here: a permutation of the launch grid for every shape.  Mode 2: z slices in
    full sets of eight go WHOLE to one XCD each, in row-major tile order; in the remaining slices (and in mode 1) each
    XCD walks one contiguous run of tiles.  Mode 3 (default): the sequence tile-major / slice-minor is cut into eight
    contiguous pieces - every XCD takes all slices of a tile back to back, tile after tile."""
    def remap(mode, gx, gy, gz, x, y, z):
        n2, z8 = gx * gy, gz & ~7
        if mode == 3:
            total = n2 * gz
            L = x + gx * y + n2 * z
            c, idx, q, r = L & 7, L >> 3, total >> 3, total & 7
            e = c * q + min(c, r) + idx
            T, bz = e // gz, e % gz
        elif mode == 2 and z < z8:
            L = x + gx * y + n2 * z
            c, idx = L & 7, L >> 3
            zl = idx // n2
            T, bz = idx - zl * n2, c + 8 * zl
        else:
            L = x + gx * y
            c, idx, q, r = L & 7, L >> 3, n2 >> 3, n2 & 7
            T, bz = c * q + min(c, r) + idx, z
        return T % gx, T // gx, bz, c
    for mode in (1, 2, 3):
        for gx, gy, gz in ((1, 1, 1), (3, 7, 1), (12, 2, 3), (47, 63, 1), (27, 22, 8), (27, 23, 17), (5, 3, 16)):
            seen, per_xcd, seq = set(), {}, {}
            for z in range(gz):
                for y in range(gy):
                    for x in range(gx):
                        bx, by, bz, c = remap(mode, gx, gy, gz, x, y, z)
                        assert 0 <= bx < gx and 0 <= by < gy and 0 <= bz < gz
                        seen.add((bx, by, bz))
                        per_xcd.setdefault((c, mode == 2 and z >= (gz & ~7), bz), []).append(by * gx + bx)
                        seq.setdefault(c, []).append((by * gx + bx) * gz + bz)
            assert len(seen) == gx * gy * gz
            if mode == 3:
                for c, es in seq.items():
                    assert es == list(range(es[0], es[0] + len(es)))       # one contiguous piece of the tile-major sequence
                continue
            for (c, tail, bz), ts in per_xcd.items():
                if mode == 2 and not tail and bz < (gz & ~7):
                    assert ts == list(range(ts[0], ts[0] + len(ts)))
                    assert len(ts) == gx * gy and bz % 8 == c              # the whole slice on one XCD
                elif mode == 1 or tail:
                    assert ts == list(range(ts[0], ts[0] + len(ts)))      # row-major, contiguous
