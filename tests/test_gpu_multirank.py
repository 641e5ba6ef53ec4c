"""GPU: the multi-process path of bench.py under the driver's `pytest -m gpu`.

`python bench.py --gpus 2` starts its two ranks itself; with HVQM4_BENCH_SHARE_GPU=1 both use the one GPU of the box.  The
parent imports neither torch nor the HIP library; it runs here as a CHILD process of the test (never an exec of this
process, which may hold the GPU already).  Checked: two ranks rendezvous (gloo: no RCCL on a path that has no collective),
every rank decodes its own streams bit-exactly (the bench's own oracle check), the end-to-end leg runs on both ranks, the
whole-job line counts both ranks' pixels, and every rank reports its core slice and copy threads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*extra, env=None):
    e = dict(os.environ, HVQM4_BENCH_SHARE_GPU="1")
    e.update(env or {})
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "8", "--distinct", "2", "--width", "320",
           "--height", "240", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-sdk"] + list(extra)
    p = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_gpu_c5():
    d = _bench()
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["streams_per_gpu"] == 8 and d["config"]["pictures_per_step"] == 8 * 16
    assert d["verified_pictures"] == d["verified_pictures_expected"] > 0
    e2e = d["end_to_end_gpu_parse"]
    assert e2e["ranks"] == 2 and e2e["value"] > 0 and e2e["streaming_value_min_rank"] > 0
    assert e2e["affinity"]["ranks_on_host"] == 2 and e2e["affinity"]["pinned"] and 1 <= e2e["host_copy_threads"] <= 8
    # whole-job value = both ranks' pixels over the slower rank's time
    assert abs(d["value"] - 2 * 8 * 16 * 320 * 240 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 0.02 * d["value"]


def test_two_ranks_on_one_gpu_c4_shards_the_clips():
    d = _bench("--workload", "c4", "--no-gpu-parse")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["streams_per_gpu"] == 32                  # 64 clips, clip i -> rank i mod 2
    assert d["verified_pictures"] == d["verified_pictures_expected"] > 0
