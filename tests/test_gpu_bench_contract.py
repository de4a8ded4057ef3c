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


# ---- N devices in ONE process: `python3 bench.py --gpus N` as typed (no launcher), through lcm_group_* ---------------
def run_bench(*argv, timeout=900):
    r = subprocess.run([sys.executable, "bench.py", *argv], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    return r


def test_force_group_reproduces_the_plain_line():
    """--gpus 1 --force-group: the N = 1 point of a scaling curve taken through the group path must be the plain
    line's number: same workload, byte-identical records and index checksums, 0 mismatches against the CPU sample."""
    common = ["--frames", "200", "--desc", "2000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--no-extras"]
    a = run_bench(*common)
    b = run_bench("--gpus", "1", "--force-group", *common)
    assert a.returncode == 0, a.stderr[-2000:]
    assert b.returncode == 0, b.stderr[-2000:]
    da, db = last_json(a.stdout), last_json(b.stdout)
    assert REQUIRED <= set(db) and db["n_gpus"] == 1 and db["scaling"] == "strong"        # (--frames overrides the workload)
    assert db["config"]["api"] == "lcm_group_all_vs_all_argmin" and da["config"]["api"] == "lcm_all_vs_all_argmin"
    assert db["config"]["pairs_per_step"] == da["config"]["pairs_per_step"] == (200 - 30) * (200 - 29) // 2
    assert db["config"]["distances_per_step"] == da["config"]["distances_per_step"]
    assert db["equals_single_handle"] == {"records": True, "index_checksums": True}
    assert db["cpu_baseline"]["gpu_vs_cpu_sample_mismatches"] == 0 and db["cpu_baseline"]["gpu_vs_cpu_index_checksum_mismatches"] == 0
    g = db["group"]
    assert g["n_devices"] == 1 and g["rccl_ranks"] == 1 and g["transport"] == "rccl" and not g["loopback"] and g["arena_allgather"]["timed_steps_skipped_it"]
    assert g["arena_allgather"]["first_search_bytes_per_device"] == 200 * 2000 * 32
    assert len(g["kernel_ms_per_device"]) == 1 and g["kernel_ms_max"] > 0
    # a 17 K-pair search is a few ms: launch-bound, so only the order of magnitude is comparable here; the 1 % agreement
    # is checked at full size by bench.py's own `extra.group_of_one` block (cfg2)
    assert 0.5 < db["value"] / da["value"] < 2.0
    rf = db["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches_per_step"] >= 1


@pytest.mark.parametrize("world", [2, 8])
def test_n_devices_in_one_process_as_typed_loopback(world):
    """`python3 bench.py --gpus N` with WORLD_SIZE unset takes the in-process route (lcm_group_create(N) ...); on this
    one-GPU box the same command line with --loopback rehearses it (N shards on the one device) — weak-scaled cfg2 shape,
    here with small frames."""
    r = run_bench("--gpus", str(world), "--loopback", "--desc", "300", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extras")
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and "LOOPBACK" in d["note"] and d["scaling"] == "weak" and d["cpu_baseline"] is None
    g = d["group"]
    assert g["n_devices"] == world and g["loopback"] and g["rccl_ranks"] == 0 and len(g["kernel_ms_per_device"]) == world
    assert sum(g["pairs_per_device"]) == d["config"]["pairs_per_step"] >= world * 470935
    assert d["merged_vs_oracle_sample"]["record_mismatches"] == 0 and d["merged_vs_oracle_sample"]["index_checksum_mismatches"] == 0
    assert d["config"]["sharding"] == "cyclic by frame" and d["value"] > 0


def test_n_devices_without_enough_gpus_says_so():
    import ctypes
    n = ctypes.CDLL(os.path.join(ROOT, "slam-loop-closing_amd", "lib", "liblcm_hip.so")).lcm_device_count()
    if n >= 8:
        pytest.skip("this box really has 8 devices")
    r = run_bench("--gpus", "8", "--desc", "300", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-extras")
    assert r.returncode != 0 and "--loopback" in (r.stderr + r.stdout)


def test_group_stream_mode_loopback():
    """--mode stream --gpus N in one process: asynchronous group tickets (lcm_group_query_submit_batch / _collect_batch)."""
    r = run_bench("--gpus", "3", "--loopback", "--mode", "stream", "--frames", "140", "--desc", "500", "--steps", "1", "--warmup", "1",
                  "--cpu-seconds", "0", "--stream-batch", "8", "--stream-depth", "3")
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["config"]["batches_in_flight"] == 3
    assert d["config"]["pairs_per_step"] == (140 - 30) * (140 - 29) // 2 and "lcm_group" in d["config"]["form"]
    assert d["merged_vs_oracle_sample"]["record_mismatches"] == 0 and d["roofline"]["launches"] > 0 and d["value"] > 0
