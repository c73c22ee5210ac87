"""The committed bench line (profiles/r2_bench_v*.json, produced by `python bench.py` on the GPU box) carries every
field the measurement contract asks for; and bench.py refuses to run without a GPU instead of falling back."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r2_bench_v*.json")) if re.search(r"r2_bench_v\d+\.json$", f)]
    files.sort(key=lambda f: int(re.search(r"_v(\d+)\.json$", f).group(1)))
    assert files, "no bench line under profiles/"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "MH proposals/sec" and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # traffic from the PMC passes equals the algorithmic state load + store of the launch
    chains, coords = d["config"]["chains_per_gpu"], d["config"]["coords"]
    assert abs(r["traffic"] - 2 * 4 * chains * coords) < 0.01 * r["traffic"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # value = proposals of the timed steps / time
    per_step = d["n_gpus"] * chains * coords * d["config"]["sweeps_per_step"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_needs_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    import torch
    if torch.cuda.is_available():
        return  # on the GPU box this process sees the card through other means; the refusal is for CPU-only hosts
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
