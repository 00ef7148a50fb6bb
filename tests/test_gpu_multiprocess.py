"""The product's N-rank gather, executed: N PROCESSES (one per rank, as `bench.py --gpus N` starts them) run
rpt_comm_init -> rpt_render_async -> rpt_gather_async -> rpt_read_gathered of csrc/rpt_comm.hip unchanged.

RCCL refuses two ranks on one device and every GPU box this build reaches has one GPU, so the ten RCCL entry points the
library resolves with dlsym come from tests/fake_rccl/librccl_fake.so here (RPT_RCCL_LIBRARY; stream-ordered send / receive
with per-channel rendezvous over shared memory).  Everything else — partition, snapshot, second stream, grouped
point-to-point calls to rank 0, strides, un-tile map, overlap with the next batch — is the code an 8-GPU run executes.
Reference: one caller, one image (src/trace.rs:136-224); the image must not depend on how many ranks rendered it.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "librccl_fake.so")
WORKER = os.path.join(ROOT, "tests", "fake_rccl", "rank_worker.py")


def _env():
    env = dict(os.environ)
    env["RPT_RCCL_LIBRARY"] = FAKE
    env["RPT_FAKE_RCCL_TIMEOUT_S"] = "60"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _unique_id():
    """rpt_comm_unique_id in a process of its own that resolves the stand-in (this pytest process may hold real RCCL)."""
    code = ("import importlib,sys; sys.path.insert(0, %r); hip = importlib.import_module('rust-path-tracer_amd.hip'); "
            "print(hip.comm_unique_id().hex()); print(hip.comm_library())" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120, check=True).stdout.split()
    assert out[1] == FAKE
    return out[0]


def _run_ranks(tmp_path, world, extra=()):
    assert os.path.exists(FAKE), "tests/fake_rccl/librccl_fake.so is missing: make fake_rccl"
    uid = _unique_id()
    procs = []
    for rank in range(world):
        log = open(tmp_path / f"rank{rank}.log", "w")
        procs.append((subprocess.Popen([sys.executable, WORKER, "--uid", uid, "--rank", str(rank), "--world", str(world), "--out", str(tmp_path), *extra],
                                       env=_env(), stdout=log, stderr=subprocess.STDOUT), log))
    failed = []
    for rank, (p, log) in enumerate(procs):
        try:
            rc = p.wait(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        log.close()
        if rc != 0:
            failed.append((rank, rc, open(tmp_path / f"rank{rank}.log").read()[-2000:]))
    assert not failed, failed
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


def _single(hipmod, rpt, world, W, H, nee, batches):
    w = world("DarkCornell")
    cfg = rpt.default_config(W, H, nee=nee)
    r = hipmod.Renderer(0)
    r.upload_scene(w); r.set_config(cfg); r.reset(rpt.blue_noise_seeds(W, H))
    images = []
    for n in batches:
        r.render(n)
        images.append(r.read_accum()[0].copy())
    st = r.stats()
    r.close()
    return images, st


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_n_processes_gather_the_one_rank_image(hipmod, rpt, world, tmp_path, ranks):
    """2, 3 and 8 processes: the image rank 0 reads equals the 1-rank render bit for bit, the communicator has N ranks as the
    collective library itself reports, the ranks' pixel blocks partition the image and their ray counts add up."""
    W, H, batches = 200, 136, (4, 4, 2)
    ref, st_ref = _single(hipmod, rpt, world, W, H, 1, batches)
    infos = _run_ranks(tmp_path, ranks, ["--width", str(W), "--height", str(H), "--nee", "1", "--batches", ",".join(map(str, batches)), "--second-image"])
    for r, info in enumerate(infos):
        assert (info["rank"], info["world"]) == (r, ranks) and info["library"] == FAKE
        assert info["ring_mismatches"] == 0                # rpt_debug_comm_selftest as a ring over the N ranks (one rank on real RCCL: test_gpu_gather.py)
    assert sum(i["pixels"] for i in infos) == W * H
    assert infos[0]["gathered_samples"] == sum(batches)
    img = np.load(tmp_path / "image.npy")
    assert np.array_equal(img.view(np.uint32), ref[-1].view(np.uint32))
    assert sum(i["extension_rays"] for i in infos) == st_ref["extension_rays"]           # (counted before the second image)
    assert sum(i["shadow_rays"] for i in infos) == st_ref["shadow_rays"]
    # flush path on the same communicator: reset + one 3-sample batch
    ref2, _ = _single(hipmod, rpt, world, W, H, 1, (3,))
    assert infos[0]["second_samples"] == 3
    assert np.array_equal(np.load(tmp_path / "image2.npy").view(np.uint32), ref2[0].view(np.uint32))


def test_overlapped_reads_see_batch_k_while_batch_k_plus_1_renders(hipmod, rpt, world, tmp_path):
    """Rank 0 reads the gathered image after every batch while the next one renders on all ranks: each read is exactly the
    1-rank image after that many batches (the snapshot precedes the next batch; the send buffer is not reused early)."""
    W, H, batches = 264, 200, (3, 3, 3, 3)
    ref, st_ref = _single(hipmod, rpt, world, W, H, 0, batches)
    infos = _run_ranks(tmp_path, 4, ["--width", str(W), "--height", str(H), "--nee", "0", "--batches", ",".join(map(str, batches)), "--read-every-batch"])
    assert infos[0]["per_batch_samples"] == [3, 6, 9]
    for k in range(3):
        got = np.load(tmp_path / f"image_after_batch{k}.npy")
        assert np.array_equal(got.view(np.uint32), ref[k].view(np.uint32)), k
    assert np.array_equal(np.load(tmp_path / "image.npy").view(np.uint32), ref[-1].view(np.uint32))
    assert sum(i["extension_rays"] for i in infos) == st_ref["extension_rays"]
    assert sum(i["samples"] for i in infos) == W * H * sum(batches)


@pytest.mark.parametrize("ranks,launcher", [(2, "torchrun"), (8, "torchrun"), (2, "plain"), (4, "plain"), (2, "multi"), (4, "multi")])
def test_bench_dress_rehearsal_of_the_scaling_run(tmp_path, ranks, launcher):
    """`bench.py --gpus N` exactly as the driver's scaling run starts it — under torch.distributed.run (one process per rank) AND as
    the plain `python bench.py --gpus N` with no launcher in the environment, where bench.py starts its ranks itself as watchdog-guarded child
    processes — except that every rank sits on the one GPU of this box (--rehearsal: gloo for torch.distributed, the stand-in for
    RCCL inside the library).  Every line the 2 / 4 / 8-GPU run executes runs here: unique id broadcast, rpt_comm_init, render +
    gather per step, drain, barriers, statistics reduction, rank 0's JSON line — whose image (gathered from all ranks) must pass
    the bitwise parity check against the oracle, with the gather reported as the library's.  `multi`: the second driver (`--driver multi`:
    ONE child process, rpt_multi_* — what the launcher falls back to when the per-process run fails or hangs), same line, same parity.
    A batch is 32 x N samples (N x as many samples of a pixel in flight on 1 / N of the image).  The number itself is not a measurement and says so."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    bench_args = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--rehearsal"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + bench_args
    elif launcher == "multi":
        cmd = [sys.executable] + bench_args + ["--driver", "multi"]
    else:
        cmd = [sys.executable] + bench_args
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert "rehearsal" in out and out["n_gpus"] == ranks and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["spp_per_step"] == 32 * ranks
    assert out["config"]["rpt_comm_world"] == [0, ranks]
    assert out["parity_check"]["bitwise"] is True and out["parity_check"]["windows"] >= 2 and out["parity_check"]["image_spp"] == 96 * ranks
    assert out["rays"]["extension"] > 1024 * 1024 * 64 * ranks and out["roofline"]["kernel"] == "k_traverse"
    if launcher == "multi":
        assert out["config"]["driver"].startswith("multi") and out["config"]["fallback_from"] is None
        assert "starting -m torch.distributed.run" not in p.stderr
    else:
        assert out["config"]["driver"].startswith("ranks")
        assert out["config"]["gather"] == "rccl-c-abi" and out["config"]["collective_library"] == FAKE
    # counter-derived per-launch figures are the kept whole-image passes scaled to the slots of rank 0's launches (1 / N of the image x N x the samples:
    # the same launch as the whole image at 32), or absent — never N x too high or too low
    rf = out["roofline"]
    if rf["traffic"] is not None:
        assert abs(rf.get("traffic_scaled_by", 1.0) - 1.0) < 0.02 and 0.9 < rf["traffic_over_algorithmic"] < 1.6
        assert 0.9 < out["pipeline_roofline"]["traffic_over_algorithmic"] < 1.6
    if launcher == "plain":
        assert "starting -m torch.distributed.run" in p.stderr
    if launcher != "torchrun":
        return
    # and WITHOUT the rehearsal flag the stand-in is refused: a scaling number can only come from RCCL
    cmd_real = [c for c in cmd if c != "--rehearsal"]
    p2 = subprocess.run(cmd_real, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert p2.returncode != 0 and "RPT_RCCL_LIBRARY" in (p2.stderr + p2.stdout)


def test_a_failing_rank_run_falls_back_to_the_one_process_driver(tmp_path):
    """The launcher's whole chain on a GPU: the one-process-per-GPU child fails (every rank exits with code 5: --rehearsal-fail-ranks), the launcher says so and starts the
    one-process driver (rpt_multi_*) as a second fresh child, whose line — bitwise parity against the oracle included — is relayed with `config.fallback_from` naming
    what happened.  (That the first child is killed when it HANGS instead: tests/test_bench_launch.py, no GPU needed.)"""
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--rehearsal", "--rehearsal-fail-ranks"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    assert "failing on purpose" in p.stderr and "second attempt with ONE process driving all 2 GPUs" in p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["driver"].startswith("multi") and "exited with code" in out["config"]["fallback_from"]
    assert out["parity_check"]["bitwise"] is True and out["parity_check"]["image_spp"] == 2 * 64


def test_bench_line_carries_the_other_single_gpu_workloads(tmp_path):
    """`bench.py --gpus 1` as the driver runs it: after the headline's timed loop the other single-GPU BASELINE workloads run on fresh
    contexts and appear under "workloads", each with its own value, roofline (dominant kernel, HIP-event timed), pipeline_roofline and a
    bitwise parity_check; the headline keeps its contract and attributes k_generate_first / k_complete in stage_ms.  (Two of the four
    extra workloads and short loops here; the driver's run takes all four.)"""
    env = dict(os.environ)
    env.pop("RPT_RCCL_LIBRARY", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-readback",
           "--extra-workloads", "darkcornell_mis,veachmis", "--extra-steps", "1", "--extra-warmup", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["metric"] == "Mrays/s" and out["n_gpus"] == 1 and out["parity_check"]["bitwise"] is True
    assert out["rays"]["shadow"] == 0 and out["rays"]["shadow_elided"] == 0 and out["value_as_the_reference_counts"] == out["value"]      # nee = 0
    sm = out["roofline"]["stage_ms"]
    assert sm["generate"] > 0 and sm["complete"] > 0 and sm["traverse"] > sm["shade"] > 0
    assert set(out["workloads"]) == {"darkcornell_mis", "veachmis"}
    for name, w in out["workloads"].items():
        assert w["value"] > 100 and w["unit"] == "Mrays/s" and w["steps"] == 1 and w["ms_per_step"] > 0
        assert w["rays"]["shadow"] > 0 and w["config"]["workload"].startswith(("DarkCornell.glb 1024x1024", "VeachMIS.glb 1920x1080"))
        # Mrays/s counts the rays WALKED on the device; the NEE evaluations whose shadow ray decides nothing (about half) are reported beside them
        assert 0.2 * w["rays"]["shadow"] < w["rays"]["shadow_elided"] < 5 * w["rays"]["shadow"]
        assert w["value_as_the_reference_counts"] > w["value"] and w["rays"]["per_sample_as_the_reference_counts"] > w["rays"]["per_sample"]
        r = w["roofline"]
        assert r["bound"] == "hbm" and r["kernel"].startswith("k_") and 0 < r["frac"] < 1 and r["avg_launch_ms"] > 0
        assert abs(r["achieved"] - r["algorithmic_bytes_per_unit"] * r["units_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-2 * r["achieved"]
        assert 0 < w["pipeline_roofline"]["frac"] < 1
        # a stage cannot move fewer HBM bytes than its algorithmic bytes: the counter figure covers every kernel the stage's time covers
        if r["traffic"] is not None:
            assert r["traffic"] >= 0.98 * r["algorithmic_bytes_per_unit"] * r["units_per_launch"], (name, r)
            assert r["traffic_over_algorithmic"] >= 0.98
            if r["kernel"] == "k_shadow":            # (the streamed LDS walk has the dense k_shadow_resolve pass behind it, the global-memory walk resolves in its own kernel)
                assert r["traffic_kernels"] in (["k_shadow", "k_shadow_resolve"], ["k_shadow"])
        pc = w["parity_check"]
        assert pc["bitwise"] is True and pc["windows"] >= 2 and pc["image_spp"] == pc["spp"] == 32
