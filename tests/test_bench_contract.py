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


def test_traffic_lookup_reads_the_committed_pmc_summary():
    cells = 960000
    t = bench.measured_traffic(bench.ElasticMarmousi.name, "adjoint+imaging", cells)
    with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
        ref = json.load(fh)[bench.ElasticMarmousi.name]["adjoint+imaging"]["bytes_per_cell_step"]
    assert t == ref * cells and bench.measured_traffic("no_such_workload", "x", 1) is None


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_default_output.json")) as fh:
        d = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] == bench.AcousticMarmousi.name and "model" not in d["config"]
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "C oracle" in c["sample"]
    # whole-job throughput = cells*steps of all shots / wall time
    wl = bench.AcousticMarmousi
    units = wl.nz * wl.nx * wl.nt * wl.shots_per_gpu * d["steps"]
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) <= 1e-6 * d["value"]


def test_pmc_tool_knows_the_kernels_the_bench_runs():
    """tools/pmc_traffic.py maps kernel-name prefixes to bench labels: every prefix must match a kernel
    of the committed kernel-trace summary, and the traffic file must cover both workloads."""
    import csv
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    names = [r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
             for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_bench_default_kernel_stats.csv")))]
    for pre in pmc_traffic.KERNELS:
        assert any(n.startswith(pre) for n in names), pre
    with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
        t = json.load(fh)
    for wl in (bench.AcousticMarmousi, bench.ElasticMarmousi):
        assert set(t[wl.name]) == {"forward+save", "adjoint+imaging"}
        for rec in t[wl.name].values():
            assert 0 < rec["bytes_per_cell_step"] < 80.0      # below the algorithmic figures of SURVEY 8d


def test_xcd_tile_order_is_a_bijection_with_contiguous_runs():
    """The per-step kernels remap blockIdx so that each XCD (launch order modulo 8) walks one contiguous
    run of tiles (csrc: xcd_tile).  Same arithmetic here: a permutation for every grid shape, and the
    tiles of one XCD are consecutive."""
    for gx in (1, 3, 12, 47):
        for gy in (1, 2, 7, 63, 250):
            n2 = gx * gy
            q, r = n2 >> 3, n2 & 7
            seen, runs = set(), {}
            for L in range(n2):
                c, idx = L & 7, L >> 3
                T = c * q + min(c, r) + idx
                assert 0 <= T < n2
                seen.add(T)
                runs.setdefault(c, []).append(T)
            assert len(seen) == n2
            for c, ts in runs.items():
                assert ts == list(range(ts[0], ts[0] + len(ts)))
