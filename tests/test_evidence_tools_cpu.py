"""The committed evidence set and the tool that prints DESIGN.md's table from it stay in step (CPU only)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_tag():
    tags = sorted({os.path.basename(p)[:5] for p in glob.glob(os.path.join(ROOT, "profiles", "r??_?_bench.json"))})
    assert tags, "no committed evidence set"
    return tags[-1]


def test_design_numbers_prints_the_table_from_the_latest_evidence_set():
    tag = _latest_tag()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), tag], capture_output=True, text=True,
                         cwd=ROOT, check=True).stdout
    assert "| what | ms | against |" in out and "two frames in flight" in out and "Content sweep" in out


def test_bench_line_of_the_latest_evidence_set_has_the_contract_fields():
    tag = _latest_tag()
    line = open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")).read().strip().splitlines()[-1]
    b = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    assert b["cpu_baseline"]["kind"] in ("port", "reference") and b["cpu_baseline"]["cores"] >= 1
    assert "workload" in b["config"] and b["vs_baseline"] is None
    # the traffic files the line's roofline reads are tagged with their workload
    for name in ("dense", "slots", "modular"):
        t = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{name}_traffic.json")))
        assert "_workload" in t
