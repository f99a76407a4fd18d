"""bench.py's own multi-rank code path, on CPU: `--gpus 2 --backend gloo --dry` must start two ranks by itself
(no external torchrun, no WORLD_SIZE in the environment), rendezvous on 127.0.0.1, shard, all-gather and print ONE
JSON line from rank 0 (BASELINE configs[2] is the same path with 8 ranks on RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{") and '"metric"' in l]


def test_gpus2_spawns_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 2 and j["dry"] is True and j["config"]["global_batch"] == 2048
    assert j["members_per_step"] == 2048 and j["solved_per_step"] == 2048.0
    assert len(j["ms_per_step_by_rank"]) == 2 and j["gather_self_check"] is True
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["warmup"] == 1


def test_single_rank_dry_line():
    r = _run(["--dry", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--batch", "16"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_lines(r.stdout)[0]
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 16 and j["ms_per_step_by_rank"] is None


def test_mismatched_world_size_is_refused():
    r = _run(["--gpus", "4", "--dry", "--backend", "gloo"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_no_gpu_means_loud_failure_not_a_cpu_number():
    r = _run(["--steps", "1", "--warmup", "0", "--batch", "4"])
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
