"""The N > 1 leg of bench.py end to end on a one-GPU box: `python bench.py --gpus 2` typed by hand starts its own
ranks (torch.distributed.run children; the parent never touches the GPU), the block-cyclic HIP driver runs with
both ranks on cuda:0 and the collectives over gloo (SGPR_BENCH_ONE_CARD=1: a rehearsal, not a measurement), and
rank 0 prints the one JSON line with roofline, cpu_baseline and the bytes every rank received."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_card():
    env = dict(os.environ, SGPR_BENCH_ONE_CARD="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-pts", "4096", "--nb", "1024",
                        "--steps", "1", "--warmup", "1", "--cpu-sample", "512"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["grid"] == [2, 1] and d["config"]["order_n"] == 8192
    assert d["residual_Ky_alpha_minus_z"] < 1e-10
    assert d["roofline"]["launches"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port")
    # algorithmic flop only (elements on / below the GLOBAL diagonal): a fraction above 1 means the count is wrong
    assert 0.0 < d["roofline"]["frac"] <= 1.0, d["roofline"]
    lo, hi = d["roofline"]["achieved_min_max_over_ranks"]
    assert 0.0 < lo <= hi <= 78.6
    assert d["rccl_world_size"] == 2 and len(d["ranks"]) == 2 and {r["rank"] for r in d["ranks"]} == {0, 1}
    assert all(r["device_index"] == 0 for r in d["ranks"])          # the rehearsal: both ranks on cuda:0
    assert d["panel_bytes_received_per_step"]["sum_over_ranks"] > 0
    assert "REHEARSAL" in d["data"]


def test_bench_three_ranks_two_pairs_per_point_on_one_card():
    """d = 2 pairs per point on a 3 x 1 grid (the generalised block-cyclic Gram build of round 3: one selection group per
    coordinate) through the same N > 1 leg; `--d` typed by hand has to survive the launcher's own argument parser"""
    env = dict(os.environ, SGPR_BENCH_ONE_CARD="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--d", "2", "--family", "C", "--n-pts", "1536",
                        "--nb", "512", "--steps", "1", "--warmup", "1", "--cpu-sample", "0"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["config"]["grid"] == [3, 1] and d["config"]["order_n"] == 6144
    assert d["config"]["pairs_per_point"] == 2
    assert d["residual_Ky_alpha_minus_z"] < 1e-10
    assert 0.0 < d["roofline"]["frac"] <= 1.0
    assert d["rccl_world_size"] == 3 and sorted(x["rank"] for x in d["ranks"]) == [0, 1, 2]
    # the block of right-hand sides against the distributed factor: column 0 is z, so it reproduces alpha
    assert d["solve_rhs_distributed"]["nrhs"] == 16 and d["solve_rhs_distributed"]["column0_vs_alpha"] < 1e-11, d["solve_rhs_distributed"]
