"""bench.py prints ONE JSON line with the fields the driver reads (runs the real script on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", *extra], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("gemm", [None, "bf16x3", "fp32"])
def test_bench_line(gemm):
    d = _run(*(["--gemm", gemm] if gemm else []))
    if gemm is None:
        gemm = "fp16x2"  # the default arithmetic
        assert d["gemm"] == "fp16x2"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "samples/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 64000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0 and r["traffic"] > 0
    if gemm == "fp16x2":
        # three products per multiply-add: the GEMMs' HBM floor is above their matrix-pipe floor
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        # (3C + 4H) elements per frame over the three launches of a block (C = 512, H = 256, T = 3999, 32 x fp32)
        assert r["algorithmic_bytes_per_launch"] == 32 * 3999 * 4.0 * (3 * 512 + 4 * 256) / 3
        assert abs(r["step_algorithmic_bytes"] - 38.54e9) < 0.01e9
        assert abs(r["step_frac"] - r["step_algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 8e12) < 1e-9
        assert r["mfma_side"]["peak_TFLOPs"] == 2500.0 / 3 and r["mfma_side"]["floor_ms"] < r["algorithmic_bytes_per_launch"] / 8e12 * 1e3
    else:
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
        assert r["peak"] == (157.3 if gemm == "fp32" else 2500.0 / 6)
    if gemm != "fp32":
        other = "bf16x3_path" if gemm == "fp16x2" else "fp16x2_path"
        assert d["fp32_mfma_path"]["ms_per_step"] > d["ms_per_step"]  # the exact-fp32 MFMA path, timed beside it
        assert d[other]["ms_per_step"] > 0
        dev = d["deviation_from_fp32_mfma_path"]
        assert 0 <= dev["max_abs"] < 1e-4 and 0 <= dev["l2_rel"] < 1e-4
    assert "cpu_baseline" not in d  # (--no-cpu-baseline)
    if gemm == "fp16x2":  # the default run carries bounded timings of the other BASELINE configurations
        oc = d["other_configs"]
        assert oc["cfg3"]["fp16x2"]["ms"] > 0 and oc["cfg3"]["bf16"]["l2_rel_vs_fp32"] < 3e-2
        assert oc["cfg4"]["ms"] > 0 and oc["cfg4"]["us_per_serial_step"] > 0 and oc["cfg4"]["serial_steps"] == 1320
        assert oc["cfg5"]["chunks"] >= 300 and 0 < oc["cfg5"]["p50_ms"] <= oc["cfg5"]["p90_ms"] <= oc["cfg5"]["max_ms"]
    else:
        assert "other_configs" not in d
    assert "profiles/" in r["traffic_note"] and r["kernel"].startswith("ps::conv1x1")
    assert d["config"]["hip_streams_per_gpu"] == 1  # timed path, events and a kernel trace describe the same launches
    assert d["distributed"] == {"world_size": 1, "backend": None, "collective": None,
                                "ms_per_step_by_rank": [d["ms_per_step"]]}
    for k in ("dwconv", "free_encode", "free_decode"):
        assert 0.05 < r["hbm_bound_kernels"][k]["frac_of_8TBps"] < 1.0

