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
    assert len(line) < 6144                                  # the driver parses ONE line; 31 KB came back as parsed = null in round 5
    assert out["rccl_ranks"] == 1 and len(out["per_rank_seconds"]) == 1
    assert out["c3_gather"]["own_rows_match_on_every_rank"] is True and out["c3_gather"]["scenes"] == 32
    assert out["c5_train_step_fp32_data_parallel"]["ms_per_step"] > 0 and out["c5_train_step_bf16_data_parallel"]["ms_per_step"] > 0
    assert set(out["roofline"]) == {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "flops_per_launch"}
    full = json.load(open(os.path.join(ROOT, out["extra_file"])))          # everything else is in the side file
    assert full["roofline"]["kernel"] in full["roofline"]["stages"]
    assert full["extra"]["c3_gather"]["checksum_qual"] == full["extra"]["c3_gather"]["checksum_qual"]


def test_contract_line_is_small_and_round_trips(tmp_path):
    """bench.py's line assembly on canned numbers (the complete record of the round-5 evidence run, 31 KB): the printed line
    carries the contract keys only, stays under 6 KB and survives json.loads; the complete record goes to the side file."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05", "final", "bench.json")).read().strip().splitlines()[-1])
    full.pop("summary")
    full["summary"] = bench.summary_of(full)
    side = tmp_path / "bench_extra.json"
    text = bench.emit(full, path=str(side))
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 6144
    line = json.loads(text)
    for k in bench.CONTRACT_KEYS:
        assert k in line, k
    assert list(line)[-1] == "summary"
    assert set(line["roofline"]) == set(bench.ROOFLINE_KEYS) and all(not isinstance(v, (dict, list)) for v in line["roofline"].values())
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
    assert set(line["cpu_baseline"]) == set(bench.CPU_BASELINE_KEYS) and line["cpu_baseline"]["value"] > 0
    assert "extra" not in line and "stages" not in line["roofline"]
    assert line["checked_vs_oracle"]["tolerance"] == 1e-4
    kept = json.load(open(side))
    assert "extra" in kept and "stages" in kept["roofline"]                 # nothing measured is lost
    # N > 1 digest keys survive the slimming
    full["extra"].update({"rccl_ranks": 8, "per_rank_seconds": [0.1] * 8, "c3_gather": {"scenes": 256, "all_gather_ms": 1.0,
                                                                                            "own_rows_match_on_every_rank": True, "checksum_qual": 1.0}})
    line = json.loads(bench.emit(full, path=str(side)))
    assert line["rccl_ranks"] == 8 and len(line["per_rank_seconds"]) == 8 and line["c3_gather"]["scenes"] == 256
    # and a summary that could not fit is shed from its tail, never the contract keys
    full["summary"] = {f"k{i}": "x" * 100 for i in range(100)}
    text = bench.emit(full, path=str(side))
    assert len(text) < bench.LINE_LIMIT and all(k in json.loads(text) for k in bench.CONTRACT_KEYS)
