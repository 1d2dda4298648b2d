"""The JSON line bench.py / bench_diar.py print (driver contract): checked on the committed logs of the last GPU run
(profiles/), so that a change of the output format shows up in the CPU suite."""
import json
import os

import pytest

from conftest import ROOT

REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict,
            "roofline": dict, "cpu_baseline": dict}


@pytest.mark.parametrize("log", ["r01_bench_r640_1h_bf16.json.log", "r01_bench_r268_1h_bf16.json.log", "r01_bench_diar_1h_bf16.json.log"])
def test_committed_bench_line_has_the_contract_fields(log):
    lines = [l for l in open(os.path.join(ROOT, "profiles", log)).read().splitlines() if l.strip()]
    d = json.loads(lines[-1])                      # the JSON line is the LAST line of stdout
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None         # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_scripts_parse_and_default_to_one_gpu():
    import ast
    for name in ("bench.py", "bench_diar.py"):
        src = open(os.path.join(ROOT, name)).read()
        ast.parse(src)
        assert '"--gpus", type=int, default=1' in src and '"--steps"' in src and '"--warmup"' in src
