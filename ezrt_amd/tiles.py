"""Image-space sharding for the multi-GPU path.

Pixels are independent, so a frame is cut into tile_w x tile_h tiles dealt round-robin to the
ranks (tile_id % world == rank -- the same rule `EzrtRenderParams.shard_*` applies inside the
kernels); every rank traces all spp of its tiles into its own full-size frame buffer, and the
frame is closed by ONE gather of the packed tiles to rank 0 (RCCL over xGMI when the tensors are
on GPUs: 7 peers -> root on 7 distinct links).  No other collective touches the data path.

torch is plumbing here (device tensors + torch.distributed); the functions also run on CPU tensors
with the gloo backend, which is how tests/test_tiles_gloo.py covers world_size 2 without GPUs.
"""
import torch


class TilePlan:
    def __init__(self, width, height, tile_w, tile_h, world):
        self.width, self.height = int(width), int(height)
        self.tile_w, self.tile_h = int(tile_w), int(tile_h)
        self.world = int(world)
        self.tiles_x = (self.width + self.tile_w - 1) // self.tile_w
        self.tiles_y = (self.height + self.tile_h - 1) // self.tile_h
        self.n_tiles = self.tiles_x * self.tiles_y
        self.per_rank = (self.n_tiles + self.world - 1) // self.world
        self._ids = {}

    def owner(self, tile_id):
        return tile_id % self.world

    def tile_ids(self, rank, device=None):
        """Tile ids owned by `rank`, padded (by repeating the last id) to `per_rank` entries."""
        key = (rank, str(device))
        if key not in self._ids:
            ids = list(range(rank, self.n_tiles, self.world))
            n_real = len(ids)
            while len(ids) < self.per_rank:
                ids.append(ids[-1] if ids else 0)
            self._ids[key] = (torch.tensor(ids, dtype=torch.long, device=device), n_real)
        return self._ids[key]

    # image [H, W, C] <-> tiles [n_tiles, tile_h, tile_w, C]
    def to_tiles(self, img):
        H, W, C = img.shape
        ph, pw = self.tiles_y * self.tile_h - H, self.tiles_x * self.tile_w - W
        if ph or pw:
            img = torch.nn.functional.pad(img, (0, 0, 0, pw, 0, ph))
        t = img.reshape(self.tiles_y, self.tile_h, self.tiles_x, self.tile_w, C).permute(0, 2, 1, 3, 4)
        return t.reshape(self.n_tiles, self.tile_h, self.tile_w, C)

    def from_tiles(self, tiles):
        C = tiles.shape[-1]
        t = tiles.reshape(self.tiles_y, self.tiles_x, self.tile_h, self.tile_w, C).permute(0, 2, 1, 3, 4)
        img = t.reshape(self.tiles_y * self.tile_h, self.tiles_x * self.tile_w, C)
        return img[: self.height, : self.width]

    def pack(self, img, rank):
        ids, _ = self.tile_ids(rank, img.device)
        return self.to_tiles(img).index_select(0, ids).contiguous()

    def unpack(self, packed_per_rank):
        """packed_per_rank[r] = what rank r packed -> the full frame."""
        first = packed_per_rank[0]
        key = ("all", str(first.device))
        if key not in self._ids:  # (destination tile id, row of the concatenated payload) of every REAL entry
            dst, src = [], []
            for r in range(self.world):
                ids, n_real = self.tile_ids(r, first.device)
                dst.append(ids[:n_real])
                src.append(torch.arange(n_real, dtype=torch.long, device=first.device) + r * self.per_rank)
            self._ids[key] = (torch.cat(dst), torch.cat(src))
        dst, src = self._ids[key]
        tiles = torch.empty((self.n_tiles, self.tile_h, self.tile_w, first.shape[-1]), dtype=first.dtype,
                            device=first.device)
        tiles.index_copy_(0, dst, torch.cat(list(packed_per_rank)).index_select(0, src))
        return self.from_tiles(tiles)


def _gather_frame_native(accum, plan, rank, dist, dst, lib, via_cpu=False):
    """gather_frame with the library's own kernels around the collective (include/ezrt_mgpu.h: ezrt_tiles_pack_device
    on every rank, ezrt_tiles_unpack_device per peer on `dst`): the layout and kernels of the one-process
    ezrt_mgpu_gather, with torch.distributed (RCCL) as the transport.  Assembles IN PLACE: on `dst` the rank's own frame
    buffer already holds its tiles and receives the peers'.  CPU tensors work with the oracle build of the header."""
    assert accum.is_contiguous() and accum.dtype == torch.float32 and accum.shape == (plan.height, plan.width, 4)
    st = torch.cuda.current_stream(accum.device).cuda_stream if accum.is_cuda else None
    args = (plan.width, plan.height, plan.tile_w, plan.tile_h)
    n = plan.per_rank * plan.tile_h * plan.tile_w * 4   # equal-size payloads for the collective (<= one tile of padding)
    # payload and receive buffers live with the plan (allocated on first use, zeroed once: the pack kernel rewrites every
    # real entry on each call, the padding is never read by the un-permute kernel) -- nothing is allocated per frame
    key = ("native", str(accum.device), bool(via_cpu), rank == dst)
    if key not in plan._ids:
        packed = torch.zeros(n, dtype=torch.float32, device=accum.device)
        wire = torch.zeros(n, dtype=torch.float32, pin_memory=accum.is_cuda) if via_cpu else packed
        bufs = [torch.zeros_like(wire) for _ in range(plan.world)] if rank == dst else None
        plan._ids[key] = (packed, wire, bufs)
    packed, wire, bufs = plan._ids[key]
    if lib.ezrt_tiles_pack_device(accum.data_ptr(), *args, rank, plan.world, packed.data_ptr(), st) != 0:
        raise RuntimeError(lib.ezrt_last_error().decode())
    if via_cpu:
        wire.copy_(packed)   # (synchronises: host staging is the gloo debugging route only)
    if rank != dst:
        dist.gather(wire, gather_list=None, dst=dst)
        return accum
    dist.gather(wire, gather_list=bufs, dst=dst)
    for r in range(plan.world):
        if r == dst:
            continue
        src = bufs[r].to(accum.device) if via_cpu else bufs[r]
        if lib.ezrt_tiles_unpack_device(src.data_ptr(), *args, r, plan.world, accum.data_ptr(), st) != 0:
            raise RuntimeError(lib.ezrt_last_error().decode())
    return accum


def gather_frame(accum, plan, rank, dist, dst=0, via_cpu=False, lib=None):
    """Close a frame: one gather of every rank's packed tiles to `dst`.  Returns the assembled
    [H, W, C] frame on `dst` and the rank's own buffer elsewhere.  via_cpu stages the payload
    through host memory (gloo debugging of the GPU driver script; never used with RCCL).
    lib (a CDLL exporting include/ezrt_mgpu.h): pack / un-permute with the library's kernels instead of torch
    indexing, in place."""
    if lib is not None:
        return _gather_frame_native(accum, plan, rank, dist, dst, lib, via_cpu)
    packed = plan.pack(accum, rank)
    if via_cpu:
        packed = packed.cpu()
    if rank == dst:
        bufs = [torch.empty_like(packed) for _ in range(plan.world)]
        dist.gather(packed, gather_list=bufs, dst=dst)
        if via_cpu:
            bufs = [b.to(accum.device) for b in bufs]
        return plan.unpack(bufs)
    dist.gather(packed, gather_list=None, dst=dst)
    return accum
