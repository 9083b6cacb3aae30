"""bench.py end to end on the one GPU of the test box: the single-rank line, and the N = 2 control flow with both
ranks sharing the device (gloo rehearsal backend: RCCL refuses two ranks on one GPU)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_single_rank_line_has_the_contract_fields():
    rec = _bench(["--workload", "tiny", "--steps", "12", "--warmup", "3", "--cpu-budget", "1"])
    assert rec["n_gpus"] == 1 and rec["steps"] == 12 and rec["scaling"] == "weak" and rec["dtype"] == "f64"
    # every sweep launch of the timed call is a sample (the launches time themselves: zrk_sweep_stamps)
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["samples"] >= 6 and rec["roofline"]["achieved"] > 0
    assert 0 < rec["roofline"]["first_wave_in_to_last_wave_out_us"] <= rec["roofline"]["avg_kernel_us"]
    assert rec["roofline"]["effective"]["achieved"] >= rec["roofline"]["achieved"]
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cpu_model"]
    assert rec["value"] > 0 and "tiny" in rec["config"]["workload"]
    # (a table this small takes the plain loop) ... and the line says when the timed call's sweeps ran
    assert rec["roofline"]["ticks_per_launch"] == 1 and rec["config"]["loop"] == "two launches per tick on one stream"
    assert rec["setup"]["sweeps_span_us"] > rec["roofline"]["first_wave_in_to_last_wave_out_us"] and rec["setup"]["sweep_gaps_us"] >= 0


def test_the_line_names_the_loop_of_the_timed_call():
    """C2 runs overlapped with two ticks per launch; the one-tick calibration calls behind the timed region take the plain loop
    and must not be what `config.loop` describes."""
    rec = _bench(["--workload", "C2", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"])
    assert rec["roofline"]["ticks_per_launch"] == 2 and rec["roofline"]["samples"] == 6
    assert rec["config"]["loop"].startswith("overlapped, two ticks per sweep launch")
    assert rec["setup"]["sweeps_span_us"] > 0 and rec["setup"]["sweep_gaps_us"] >= 0


@pytest.mark.parametrize("workload,scaling,wire", [("tiny", "weak", "union"), ("tiny4", "strong", "union"), ("tiny", "weak", "masks")])
def test_two_ranks_started_by_the_gpus_flag(workload, scaling, wire):
    rec = _bench(["--gpus", "2", "--workload", workload, "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--wire", wire],
                 env={"ZRK_BENCH_BACKEND": "gloo"})
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling
    assert rec["config"]["exchange_overflow"] is False
    assert ("masks" in rec["config"]["exchange_wire"]) == (wire == "masks")
    assert rec["value"] > 0


def test_ensemble_workload_line():
    """--workload tiny5: configs[4]'s mechanics (one batched table of independent scenarios, no exchange) at test size."""
    rec = _bench(["--workload", "tiny5", "--steps", "12", "--warmup", "4", "--cpu-budget", "1"])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "weak" and "tiny5" in rec["config"]["workload"]
    assert rec["config"]["parallelism"] == "replicas1" and rec["roofline"]["achieved"] > 0
    assert rec["roofline"]["samples"] >= 6 and rec["roofline"]["ticks_per_launch"] == 1      # (an ensemble's sweeps time themselves too)
    assert rec["cpu_baseline"]["kind"] == "port" and rec["value"] > 0


def test_exchange_control_flow_on_one_rank_with_real_rccl():
    """ZRK_BENCH_FORCE_EXCHANGE: everything bench.py does for N > 1 -- the library's own RCCL communicator, list sizing after
    warm-up, the per-tick all-gather from the C side, the overflow report -- with a one-rank communicator."""
    rec = _bench(["--workload", "tiny", "--steps", "16", "--warmup", "40", "--no-cpu-baseline"],
                 env={"ZRK_BENCH_FORCE_EXCHANGE": "1"})
    assert rec["n_gpus"] == 1 and rec["config"]["exchange"] == "rccl, C side"
    assert rec["config"]["exchange_overflow"] is False and rec["config"]["exchange_entries_per_rank"] > 0
    assert rec["config"]["rccl_ranks_seen"] in (1, -1) and rec["config"]["exchange_pattern"].startswith("ncclAllGather")
    assert rec["config"]["exchange_host_wait_us_per_tick"] >= 0 and "bitmap" in rec["config"]["exchange_wire"]
    assert rec["value"] > 0


def test_closed_loop_workload_line():
    """--workload tiny-battery: the C2-battery line's mechanics at test size -- the battery's closed loop as the step."""
    rec = _bench(["--workload", "tiny-battery", "--steps", "40", "--warmup", "8", "--cpu-budget", "1"])
    assert rec["n_gpus"] == 1 and "closed loop" in rec["config"]["workload"] and rec["config"]["ticks_run"] == 48
    assert rec["config"]["launch_solves"] > 0 and rec["config"]["launches"] > 0
    assert rec["roofline"]["samples"] == 2 and rec["roofline"]["achieved"] > 0 and rec["value"] > 0
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == 1
