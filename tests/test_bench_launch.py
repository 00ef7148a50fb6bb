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
    # ... and only then the second driver was tried, as another fresh child: ONE process for all GPUs (rpt_multi_*), which fails as loudly
    assert "second attempt with ONE process driving all 2 GPUs" in p.stderr and "the one-process driver exited with code" in p.stderr
    assert p.stderr.count("no HIP device visible") >= 3   # two ranks + the one-process driver
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_driver_ranks_makes_no_second_attempt(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--driver", "ranks"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p.returncode != 0 and "no second attempt" in p.stderr and "ONE process driving" not in p.stderr


def test_the_watchdog_kills_a_hung_child_and_everything_it_started(tmp_path):
    """A communicator bring-up that never returns (the first run on a real 8-GPU box may meet one) must not burn the driver's whole budget: the
    child runs in its own process group under a watchdog; on expiry the GROUP is killed — the child and the rank processes it started, nothing
    looked up by name —, the last lines of its stderr and the hints are printed, and the caller gets `timed out`."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pidfile = tmp_path / "pids"
    child = ("import os, subprocess, sys, time\n"
             "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)'])\n"
             f"open({str(pidfile)!r}, 'w').write(f'{{os.getpid()}} {{g.pid}}')\n"
             "print('rank 0: ncclCommInitRank ...', file=sys.stderr); sys.stderr.flush()\n"
             "print('{\"partial\": 1}'); sys.stdout.flush()\n"
             "time.sleep(300)\n")
    t0 = time.perf_counter()
    rc, out, tail, timed_out = bench.run_child([sys.executable, "-c", child], dict(os.environ), 3.0, "the sleeping child")
    assert timed_out and rc is None and time.perf_counter() - t0 < 30
    assert any("ncclCommInitRank" in line for line in tail) and out == ['{"partial": 1}\n']
    pids = [int(x) for x in pidfile.read_text().split()]
    time.sleep(0.5)
    for pid in pids:                                       # the child AND its own child are gone
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid
    # a child that finishes in time is simply relayed
    rc, out, tail, timed_out = bench.run_child([sys.executable, "-c", "import sys; print('{}'); print('bye', file=sys.stderr); sys.exit(7)"], dict(os.environ), 30.0, "quick")
    assert (rc, out, timed_out) == (7, ["{}\n"], False) and tail == ["bye\n"]
