"""Framebuffer tile partition + the single per-batch gather (one process per GPU).

The reference has no multi-device path; pixels are independent
(kernels/src/lib.rs:209-226 touch only output[i] / rng[i]), so the image is cut
into 64x64 tiles dealt round-robin to ranks and every rank renders its own
pixels with global coordinates.  There is NO data-path collective while
rendering; after a sample batch ONE gather moves each rank's contiguous
tile-major accumulator block to rank 0 (RCCL over xGMI via torch.distributed,
backend "nccl"; "gloo" in the CPU tests), and rank 0 un-tiles.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import hip


def block_sizes(width, height, world_size):
    return [len(hip.tile_order(width, height, r, world_size)) for r in range(world_size)]


def untile_host(blocks, width, height, world_size):
    """Host reference of rpt_untile: list of per-rank (n_r, 4) arrays -> (H, W, 4) image."""
    img = np.zeros((height, width, 4), np.float32)
    for r, blk in enumerate(blocks):
        xy = hip.tile_order(width, height, r, world_size)
        img[(xy >> 16).astype(np.int64), (xy & 0xFFFF).astype(np.int64)] = np.asarray(blk, np.float32).reshape(-1, 4)
    return img


def tile_block_from_image(image, rank, world_size):
    """Extract rank's tile-major block from a full (H, W, 4) image (what a rank would hold)."""
    h, w = image.shape[:2]
    xy = hip.tile_order(w, h, rank, world_size)
    return np.ascontiguousarray(image[(xy >> 16).astype(np.int64), (xy & 0xFFFF).astype(np.int64)], np.float32)


class Gatherer:
    """Pre-allocated form of gather_blocks for the per-batch hot loop: every rank's block lands in its own
    equal-sized slot of ONE contiguous tensor on rank 0 (slot stride = the largest block), which rpt_untile
    consumes directly (block_stride_pixels = stride) — no per-step allocation, padding copy or concatenation."""

    def __init__(self, width, height, device, group=None, stream=None):
        """stream: the torch stream (e.g. torch.cuda.ExternalStream(renderer.stream_ptr())) the staging copy and the
        collective are ordered on.  The renderer works on its own non-blocking HIP stream, so WITHOUT it the caller must
        either wrap begin()/end() in `with torch.cuda.stream(...)` itself (bench.py does) or call renderer.wait() first —
        otherwise the copy races with the batch that is still accumulating."""
        self.group = group
        self.stream = stream
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sizes = block_sizes(width, height, self.world)
        self.stride = max(self.sizes)
        self.send = torch.zeros((self.stride, 4), dtype=torch.float32, device=device)
        self.recv = torch.zeros((self.world, self.stride, 4), dtype=torch.float32, device=device) if self.rank == 0 else None
        self.recv_list = [self.recv[r] for r in range(self.world)] if self.rank == 0 else None

        self._work = None

    def gather(self, local_block):
        """local_block: (n_local, 4) tensor aliasing the renderer's accumulator block. Returns recv on rank 0."""
        self.begin(local_block)
        return self.end()

    def begin(self, local_block):
        """Start the gather of this batch and return at once: the block is first copied into a torch-owned staging
        buffer (the collective never sees memory it did not allocate; 2 MiB per rank at 1024^2; also the move to the
        host for the gloo tests), so the renderer may go on accumulating the next batch while the blocks travel."""
        assert self._work is None, "previous gather not finished (call end())"
        n = self.sizes[self.rank]
        import contextlib
        with (torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()):
            self.send[:n].copy_(local_block)
            self._work = dist.gather(self.send, gather_list=self.recv_list, dst=0, group=self.group, async_op=True)

    def end(self):
        """Wait for the gather started by begin(); returns recv on rank 0 (None elsewhere, or when nothing is pending)."""
        if self._work is None:
            return None
        self._work.wait()
        self._work = None
        return self.recv


def gather_blocks(local_block, width, height, group=None):
    """The one collective of the multi-GPU path: gather per-rank blocks to rank 0.

    local_block: torch tensor (n_local, 4) float32 on this rank's device (CPU for gloo).
    Returns on rank 0 the list of per-rank tensors, elsewhere None.  Blocks may
    differ in size by a few tiles, so they travel padded to the largest block.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = block_sizes(width, height, world)
    n_max = max(sizes)
    padded = torch.zeros((n_max, 4), dtype=torch.float32, device=local_block.device)
    padded[: local_block.shape[0]] = local_block
    if rank == 0:
        recv = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, gather_list=recv, dst=0, group=group)
        return [recv[r][: sizes[r]] for r in range(world)]
    dist.gather(padded, gather_list=None, dst=0, group=group)
    return None


def device_block_as_tensor(renderer, device):
    """Alias the renderer's tile-major accumulator block (device memory owned by librpt_hip) as a torch tensor.
    The alias is invalidated by anything that re-allocates the path state (set_config with a new size, set_partition,
    set_samples_in_flight): fetch it again afterwards."""
    n = renderer.local_pixels()
    ptr = renderer.local_block_device_ptr()

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(h, device=device)
