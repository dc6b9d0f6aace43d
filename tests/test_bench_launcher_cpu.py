"""`python bench.py --gpus N` must start its N ranks itself (no launcher around it) and relay rank 0's JSON line + the worst exit code.
Runs here without a GPU through ADA_BENCH_STUB=1: the launcher, the gloo process group, the barrier / max-over-ranks timing and the
DepthGather exchange are the real code paths of bench.py; only the step is a stand-in (the HIP model is covered by the -m gpu suite)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    env = dict(os.environ, ADA_BENCH_STUB="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env, timeout=600)


def test_bench_gpus2_launches_its_own_ranks():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--size", "28", "--repeats", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["global_batch"] == 8 and line["scaling"] == "weak"
    assert line["data"].startswith("stub") and line["value"] > 0
    # per-rank times of the headline block: the value is computed from the slowest rank, the spread shows a straggler
    assert len(line["ms_per_step_ranks"]) == 2 and abs(max(line["ms_per_step_ranks"]) - line["ms_per_step"]) < 1e-6
    assert 0.0 <= line["rank_spread_pct"] < 100.0 and line["cpu_baseline"] is None and "N = 1" in line["cpu_baseline_note"]
    # every rank's count of re-run images and its largest r reach rank 0 through the group (round 6; the stub's stand-in values: rank, 0.25 + 0.01 rank)
    assert line["escalated_ranks"] == [0, 1] and [round(v, 2) for v in line["r_max_ranks"]] == [0.25, 0.26]
    assert len(r.stdout.strip().splitlines()) == 1, "only rank 0's JSON line may reach stdout"


def test_bench_launcher_reports_a_failing_rank():
    # a negative image size makes every rank raise while it builds its inputs: the parent must exit non-zero and print no JSON line
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--size", "-1", "--repeats", "0")
    assert r.returncode != 0
    assert not any(ln.startswith("{") and '"metric"' in ln for ln in r.stdout.splitlines())
