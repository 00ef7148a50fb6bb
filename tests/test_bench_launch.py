"""`python bench.py --gpus N` (N > 1) without a launcher: bench.py starts its ranks itself (child processes under
torch.distributed.run) and relays their exit code — the driver's plain command must never die in an argument check.
No GPU here: the children get as far as "no HIP device" and the launcher hands their failure back."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_multi_gpu_command_starts_its_own_ranks_and_relays_their_exit_code(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WORLD_SIZE"] = "1"                              # an inherited single-process "distributed" environment is no launcher either
    env["RANK"] = "0"
    env["HIP_VISIBLE_DEVICES"] = ""                      # (were a GPU present: this test is about the launcher only)
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert "starting -m torch.distributed.run --nnodes=1 --nproc-per-node=2" in p.stderr, p.stderr[-2000:]
    assert "must be launched with" not in p.stderr
    assert "no HIP device visible" in p.stderr            # both ranks ran main() with WORLD_SIZE = 2 and failed loudly: no CPU fallback
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
