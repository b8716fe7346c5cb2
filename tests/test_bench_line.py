"""The one JSON line of the default `python bench.py` run, as the driver records it (it keeps the last 8 KB of stdout)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_default_bench_line_fits_the_drivers_tail_and_names_every_baseline_config():
    sys.path.insert(0, ROOT)
    import bench
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    # the default invocation with short legs: same code path, same fields (traffic passes off: they are two more child runs)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--min-seconds", "0.3", "--traffic", "off",
                        "--cpu-scale", "0.03"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.strip().count("\n") == 0, p.stdout[-2000:]
    assert len(lines[0]) < 8000, len(lines[0])
    d = json.loads(lines[0])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and "aai_axis_kernel" in d["roofline"]["kernel"]
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert [c["name"] for c in d["configs"]] == [n for n, _ in bench.CONFIG_SET]
    for c in d["configs"]:
        assert c["ms"] > 0 and c["mpix_s"] > 0 and 0 < c["frac"] < 1 and c["kernel"].startswith("aai_"), c
        if "cpu" in c:
            assert c["cpu"][0] > 0 and c["gpu_over_cpu"] > 1, c
    by = {c["name"]: c for c in d["configs"]}
    assert "aai_cell_kernel" in by["cfg3"]["kernel"] and "aai_cell_kernel" in by["cfg5"]["kernel"]
    assert by["cfg4"]["images"] == 64 and by["cfg2"]["dst"] == [2048, 2048] and by["cfg5"]["dst"] == [23170, 23170]
