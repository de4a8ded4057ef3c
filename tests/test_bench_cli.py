"""bench.py without a GPU (this container): it must refuse loudly — there is no CPU fallback to time — whatever launch
form is asked for, and `--gpus N` as typed must not demand a launcher any more (round 2 did)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="a HIP device is present: the GPU contract tests cover bench.py")
@pytest.mark.parametrize("argv", [[], ["--gpus", "8"], ["--gpus", "8", "--loopback"], ["--gpus", "1", "--force-group"], ["--mode", "stream"]])
def test_bench_refuses_to_run_without_a_device(argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", *argv, "--steps", "1", "--warmup", "0"], cwd=ROOT, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0
    msg = r.stderr + r.stdout
    assert "no CPU fallback" in msg
    assert "torch.distributed.run" not in msg            # `--gpus N` needs no launcher
    assert not any(line.startswith("{") for line in r.stdout.splitlines())      # and no JSON line is faked
