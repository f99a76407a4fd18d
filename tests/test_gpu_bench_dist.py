"""The multi-GPU code path of bench.py on REAL RCCL, as far as a 1-GPU box allows: the driver's own command line
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`)
with N = 1 and `--force-dist`, which makes the single rank go through everything the ranks of an 8-GPU run go through --
process group on the nccl (= RCCL) backend, barrier + synchronize bracketing, the all-gather of x* / status into a
world * B buffer, the MAX / SUM reductions -- and print the one JSON line with the gather self-check (VERDICT r1
item 1 asked for the launcher on CPU/gloo, tests/test_bench_launcher_cpu.py; this is its GPU counterpart)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_driver_command_with_one_rank_over_rccl():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--force-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)     # a child process: this test process never execs after touching the GPU
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    j = lines[0]
    assert j["n_gpus"] == 1 and j["config"]["global_batch"] == 1024 and j["steps"] == 2
    assert j["gather_self_check"] is True and len(j["ms_per_step_by_rank"]) == 1
    assert j["solved_per_step"] >= 1023 and j["value"] > 0
