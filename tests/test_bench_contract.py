"""The JSON line bench.py prints (its latest committed capture under profiles/) carries every key of the driver's contract."""
import glob
import json
import os
import re

from tests.conftest import ROOT


def _latest():
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_bench_v*.json"))
    key = lambda p: tuple(int(v) for v in re.findall(r"r(\d+)_bench_v(\d+)", os.path.basename(p))[0])
    return max((f for f in files if re.search(r"_bench_v\d+\.json$", f)), key=key)


def test_committed_bench_line_has_the_contract_keys():
    d = json.load(open(_latest()))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"] == "corner_dams_256" and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 0.01                 # steps/s and ms/step describe the same run
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"]          # PMC traffic can only exceed the algorithmic bytes
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["avg_us"] * 1e-6) / 1e9) < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1


def test_committed_multi_gpu_lines_carry_the_cut_planes_and_the_secondary():
    """`bench.py --gpus N` lines captured on the GPU box (N processes on its one GPU, tools/multiproc_direct_bench.sh): the contract keys, the cut planes and
    FLUID bricks per rank (round-4 review, item 1a), the recovery counter and -- strong scaling of the default scene -- the corner_dams_512 secondary (1d)."""
    path = os.path.join(ROOT, "profiles", "r05_multiproc_direct.jsonl")
    lines = [json.loads(l) for l in open(path) if l.startswith("{")]
    assert len(lines) >= 6
    with_secondary = [d for d in lines if d.get("secondary") and "value" in d["secondary"]]
    assert with_secondary, "no captured line carries a completed secondary run"
    for d in lines:
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "fluid_bricks_per_rank", "transport", "transport_ops_per_step"):
            assert k in d, k
        assert d["n_gpus"] >= 2 and d["scaling"] == "strong" and len(d["fluid_bricks_per_rank"]) == d["n_gpus"]
        cuts = d["config"]["slab_cuts"]
        assert cuts[0] == 0 and cuts[-1] == d["config"]["grid"][2] and len(cuts) == d["n_gpus"] + 1 and all(c % 4 == 0 for c in cuts)
        if d["config"]["slab_cuts_mode"] != "uniform":
            assert min(d["fluid_bricks_per_rank"]) > 0                      # weighted cuts: nobody starts without fluid
    s2 = with_secondary[-1]["secondary"]
    assert s2["grid"] == [512, 512, 512] and s2["particles"] == 8065008 and s2["value"] > 0 and len(s2["fluid_bricks_per_rank"]) == with_secondary[-1]["n_gpus"]
