"""The JSON line bench.py prints is a contract with the driver (metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling /
dtype / data / config + roofline + cpu_baseline).  CPU-only check of the line the last evidence call of the round committed under profiles/:
the keys the contract names are there, the numbers are consistent with each other, and the blocks round 5 added (PCIe-inclusive rate, the
32-image per-GPU load of the 8-GPU run) carry what DESIGN.md quotes.  (bench.py itself needs a GPU: the product path has no CPU fallback.)"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_bench_bf16.log")))
    assert logs, "no committed bench line under profiles/"
    with open(logs[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1]), os.path.basename(logs[-1])


def test_committed_bench_line_keeps_the_contract():
    d, name = _line()
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert baseline["metric"].startswith(d["metric"]), (d["metric"], baseline["metric"])
    for k, t in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float), ("higher_is_better", bool),
                 ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (name, k, type(d[k]))
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    gb = d["config"]["global_batch"]
    assert abs(d["value"] - gb / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"], "value = whole-job images per second"
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    # the dominant family's time fits inside the step it was measured in (serialized instrumented step: a little longer than the timed one)
    fam_ms = sum(v["ms"] for v in r["by_kernel"].values())
    assert fam_ms < 1.2 * d["ms_per_step"], (fam_ms, d["ms_per_step"])
    h = d["roofline_hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and 0.0 < h["frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 100 * c["value"], "the GPU line next to the CPU baseline of the same run"


def test_round5_blocks_of_the_default_line():
    d, name = _line()
    if not name.startswith("r05") and "h2d_inclusive" not in d:
        return
    h = d["h2d_inclusive"]
    assert h["ms_per_step"] >= 0.98 * d["ms_per_step"] and h["h2d_mb_per_step"] > 100, "the PCIe-inclusive step is not faster than the resident one"
    p = d["per_gpu_32"]
    assert p["per_gpu_batch"] == 32 and abs(p["no_comm_projection_8gpu"] - 8 * p["value_one_gpu"]) < 1.0
    assert p["ms_per_step"] < d["ms_per_step"] and 0.0 < p["roofline"]["frac"] < d["roofline"]["frac"]
    also = d["also"]
    assert also["dtype"] == "f32" and also["roofline"]["peak"] < d["roofline"]["peak"]
