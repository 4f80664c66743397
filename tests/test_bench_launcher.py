"""bench.py's launch contract: `python bench.py --gpus N` starts its own N ranks, WORLD_SIZE must equal --gpus, and the
whole N > 1 code path (gloo barriers, RCCL group after the timed region, counter all_gather, the c3 output gather, the
data-parallel c5 step) runs under the launcher at world size 1 on a single-GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_flag_without_devices_is_refused():
    """No silent single-rank run: asking for more GPUs than are visible is an error (here: none are visible)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the spawn would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_multi_rank_code_path_at_world_size_one():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29731", BENCH, "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=580, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 0
    ex = out["extra"]
    assert ex["rccl_ranks"] == 1
    assert ex["c3_gather"]["own_rows_match_on_every_rank"] is True and ex["c3_gather"]["scenes"] == 32
    assert ex["c5_train_step_fp32_data_parallel"]["ms_per_step"] > 0 and ex["c5_train_step_bf16_data_parallel"]["ms_per_step"] > 0
    assert out["roofline"]["kernel"] in out["roofline"]["stages"]
