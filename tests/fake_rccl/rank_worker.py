"""One RANK of a multi-process render through the C ABI — the process shape `bench.py --gpus N` has (one process per rank):

    rpt_create -> rpt_comm_init(unique id, rank, world) -> rpt_upload_scene / rpt_set_config / rpt_reset
    per batch: rpt_render_async ; rpt_gather_async          (csrc/rpt_comm.hip: snapshot, ncclSend / ncclRecv, root un-tile)
    rank 0:    rpt_read_gathered -> <out>/image.npy

Started by tests/test_gpu_multiprocess.py with RPT_RCCL_LIBRARY pointing at tests/fake_rccl/librccl_fake.so, every rank on the
one GPU of the test box.  Writes <out>/rank<r>.json = what this rank saw (communicator size as the library reports it, ray
counts, pixels owned).  No torch: the ranks meet through the 128-byte unique id on the command line."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--uid", required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--scene", default="DarkCornell")
    ap.add_argument("--width", type=int, default=200)
    ap.add_argument("--height", type=int, default=136)
    ap.add_argument("--nee", type=int, default=1)
    ap.add_argument("--batches", default="4,4,2")
    ap.add_argument("--read-every-batch", action="store_true", help="rank 0 reads the gathered image after every batch (overlapped loop)")
    ap.add_argument("--second-image", action="store_true", help="afterwards: reset and render once more (flush path), saved as image2.npy")
    args = ap.parse_args()

    rpt = importlib.import_module("rust-path-tracer_amd")
    hip = importlib.import_module("rust-path-tracer_amd.hip")
    world = rpt.World.from_path(rpt.fixture(args.scene + ".glb"))
    cfg = rpt.default_config(args.width, args.height, nee=args.nee)
    seeds = rpt.blue_noise_seeds(args.width, args.height)
    batches = [int(b) for b in args.batches.split(",")]

    r = hip.Renderer(0, rank=args.rank, world_size=args.world)
    r.comm_init(bytes.fromhex(args.uid), args.rank, args.world)
    seen_rank, seen_world = r.comm_world()
    ring_bad = r.comm_selftest(70001) if args.world > 1 else 0     # rpt_debug_comm_selftest: a ring of grouped send / receive over all ranks
    r.upload_scene(world); r.set_config(cfg); r.reset(seeds)
    per_batch = []
    for k, n in enumerate(batches):
        r.render_async(n)
        if args.read_every_batch and args.rank == 0 and k > 0:
            img, s = r.read_gathered()                       # the image after batch k-1, while batch k renders
            per_batch.append((int(s), img.copy()))
        r.gather_async()
    r.gather_wait()
    r.wait()
    st = r.stats()
    info = {"ring_mismatches": int(ring_bad), "rank": seen_rank, "world": seen_world, "library": hip.comm_library(), "pixels": int(r.local_pixels()),
            "extension_rays": int(st["extension_rays"]), "shadow_rays": int(st["shadow_rays"]), "samples": int(st["samples"])}
    if args.rank == 0:
        img, s = r.read_gathered()
        info["gathered_samples"] = int(s)
        np.save(os.path.join(args.out, "image.npy"), img)
        for i, (s_k, img_k) in enumerate(per_batch):
            np.save(os.path.join(args.out, f"image_after_batch{i}.npy"), img_k)
        info["per_batch_samples"] = [s_k for s_k, _ in per_batch]
    if args.second_image:
        r.reset(seeds)
        r.render_async(3)
        r.gather_async()
        r.gather_wait()
        r.wait()
        if args.rank == 0:
            img, s = r.read_gathered()
            info["second_samples"] = int(s)
            np.save(os.path.join(args.out, "image2.npy"), img)
    with open(os.path.join(args.out, f"rank{args.rank}.json"), "w") as f:
        json.dump(info, f)
    r.close()


if __name__ == "__main__":
    main()
