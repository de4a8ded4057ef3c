"""bench.py's output contract, on small workloads: one JSON line with the required keys at N = 1, and the N > 1 code
path (process group, cyclic sharding, all-gather of score records, merge check) rehearsed with 2 gloo ranks sharing
the one GPU of the box (RCCL itself needs one GPU per rank; the driver runs that at round end)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def last_json(stdout):
    lines = [l for l in stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_single_gpu_contract():
    r = subprocess.run([sys.executable, "bench.py", "--frames", "120", "--desc", "500", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "distances/s" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] == "hbm"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["cores"] >= 1
    assert cb["gpu_vs_cpu_sample_mismatches"] == 0
    assert d["config"]["pairs_per_step"] == (120 - 30) * (120 - 29) // 2
    assert d["value"] > 0 and d["roofline_valu"]["bound"] == "valu"


def test_two_rank_rehearsal_gloo():
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--frames", "150",
                        "--desc", "500", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--cpu-seconds", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == "cyclic by frame"
    assert d["config"]["pairs_per_step"] == (150 - 30) * (150 - 29) // 2       # both shards together = the whole search
    assert d["value"] > 0
    assert d["merged_shards_vs_oracle_sample_mismatches"] == 0      # gathered + merged records are right BY VALUE


def test_stream_mode_contract():
    """--mode stream (configs[4] shape) must carry roofline and cpu_baseline too, and its records must equal the oracle's."""
    r = subprocess.run([sys.executable, "bench.py", "--mode", "stream", "--frames", "150", "--desc", "600", "--steps", "1",
                        "--warmup", "1", "--cpu-seconds", "1", "--stream-batch", "8"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["config"]["stream_batch"] == 8 and d["config"]["pairs_per_step"] == (150 - 30) * (150 - 29) // 2
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches"] >= 150 // 8 and d["roofline_valu"]["achieved"] > 0 and 0 < d["device_busy_frac"] <= 1.0
    assert d["cpu_baseline"]["gpu_vs_cpu_sample_mismatches"] == 0 and d["cpu_baseline"]["kind"] == "port"


def test_three_rank_stream_rehearsal_gloo():
    """The N > 1 form of --mode stream (configs[4] shape: every rank sees every frame, appends the ones it owns, scores
    are gathered once per step) with 3 gloo ranks sharing the box's GPU; the merged records are checked against the
    pair count of the whole search."""
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "3", "--mode", "stream",
                        "--frames", "130", "--desc", "400", "--steps", "1", "--warmup", "1", "--backend", "gloo",
                        "--cpu-seconds", "0", "--stream-batch", "8"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 3 and d["config"]["sharding"] == "cyclic by frame" and d["config"]["stream_batch"] == 8
    assert d["config"]["pairs_per_step"] == (130 - 30) * (130 - 29) // 2
    assert d["value"] > 0 and d["roofline"]["launches"] > 0
