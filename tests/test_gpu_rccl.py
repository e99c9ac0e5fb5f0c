"""-m gpu: the collective library the N > 1 path uses (torch.distributed backend "nccl" = RCCL) is loaded and runs a collective on
this box — one rank, one GPU.  Not a scaling measurement: the shards of the table pool have no data-path collective, bench.py ends
with ONE gather of episode returns (reference: the in-process sum of one_vs_three.rs:55-60); this only makes sure that the first
time librccl is touched is not the driver's 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_launch_check():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["launch_check"] is True and line["ranks"] == 1
    assert line["rccl_world1"] is True and line["rccl_loaded"] is True, line
