"""Pixel-tile sharding of a frame over the GPUs of one node (host side).

The path shards by construction: a pixel's RNG key is (x, y, seed, counter) (rng.hlsli:35-47) and a
pixel owns all of its outputs, so ranks never exchange anything while tracing. The only exchange is
the assembly of the framebuffer at the end of a render call: every rank holds a full-size RGBA32F
image that is zero outside its tiles, and a sum-reduce over ranks (RCCL over xGMI, or gloo on CPU)
yields the frame. The ownership rule below is the one sthip_set_shard applies on the device
(tile t, row-major, is owned by rank t % world).
"""
import numpy as np


def tile_grid(width, height, tile_w=64, tile_h=32):
    return (width + tile_w - 1) // tile_w, (height + tile_h - 1) // tile_h


def owner_map(width, height, world, tile_w=64, tile_h=32):
    """(H, W) int32 array: the rank that owns each pixel."""
    tx, _ = tile_grid(width, height, tile_w, tile_h)
    ys, xs = np.mgrid[0:height, 0:width]
    tile = (ys // tile_h) * tx + (xs // tile_w)
    return (tile % world).astype(np.int32)


def owned_mask(width, height, rank, world, tile_w=64, tile_h=32):
    return owner_map(width, height, world, tile_w, tile_h) == rank


def reduce_framebuffer(tensor, dist, dst=0):
    """Sum-reduce of the per-rank framebuffers (disjoint support, so the sum is exact)."""
    dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


# ---- exchange without the zero padding: each rank sends only its own tiles (1 / world of the frame) ----
def slot_count(width, height, rank, world, tile_w=64, tile_h=32):
    """Entries of a rank's packed tile buffer (sthip_shard_slot_count): owned tiles * tile_w * tile_h."""
    tx, ty = tile_grid(width, height, tile_w, tile_h)
    tiles = tx * ty
    owned = (tiles - rank + world - 1) // world if tiles > rank else 0
    return owned * tile_w * tile_h


def slot_pixels(width, height, rank, world, tile_w=64, tile_h=32):
    """(slots, 2) int array of the (x, y) pixel of every slot of a rank's packed buffer (-1 for slots outside the image):
    the host mirror of slot_to_pixel (stratum_amd/csrc/kernels.h): owned tiles in order, 8x8 pixel blocks inside a tile."""
    tx, ty = tile_grid(width, height, tile_w, tile_h)
    n = slot_count(width, height, rank, world, tile_w, tile_h)
    slot = np.arange(n)
    per_tile = tile_w * tile_h
    local_tile, r = slot // per_tile, slot % per_tile
    tile = local_tile * world + rank
    tyi, txi = tile // tx, tile % tx
    b, lane = r >> 6, r & 63
    blocks_x = tile_w >> 3
    by, bx = b // blocks_x, b % blocks_x
    px = txi * tile_w + (bx << 3) + (lane & 7)
    py = tyi * tile_h + (by << 3) + (lane >> 3)
    inside = (px < width) & (py < height) & (tile < tx * ty)
    return np.where(inside[:, None], np.stack([px, py], 1), -1)


def assemble_tiles(packed_per_rank, width, height, tile_w=64, tile_h=32):
    """numpy mirror of sthip_assemble_tiles: list of (slots_r, 4) arrays -> (H, W, 4) frame."""
    world = len(packed_per_rank)
    frame = np.zeros((height, width, 4), np.float32)
    for rank, packed in enumerate(packed_per_rank):
        xy = slot_pixels(width, height, rank, world, tile_w, tile_h)
        ok = xy[:, 0] >= 0
        frame[xy[ok, 1], xy[ok, 0]] = packed[: xy.shape[0]][ok]
    return frame


def gather_tiles(packed, gathered, dist, dst=0, async_op=False):
    """packed: this rank's tile buffer (rank_stride, 4), the same size on every rank; gathered: (world, rank_stride, 4)
    on `dst` (None elsewhere). One message of 1 / world of the frame per rank, straight to dst over its xGMI link."""
    lst = [gathered[i] for i in range(gathered.shape[0])] if gathered is not None else None
    return dist.gather(packed, gather_list=lst, dst=dst, async_op=async_op)



# ---- the other split (SURVEY.md 8e "replicas + sum-reduce"): whole frames over disjoint seed ranges ----
def seed_range(rank, world, seed_count):
    """(first, count) of the seeds of a call that rank `rank` renders: the first seed_count % world ranks take one more
    (the rule of MultiDeviceBDPT::split_seeds, stratum_hip_multi.hpp); count may be 0 when there are more ranks than seeds."""
    base, extra = divmod(seed_count, world)
    first = base * rank + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def to_sums(image):
    """(mean over the seeds, their number) -> (sum, number) per pixel: what a sum-reduce can add (sthip_radiance_to_sums).
    Works on numpy arrays and torch tensors of shape (..., 4), in place."""
    image[..., :3] *= image[..., 3:4]
    return image


def from_sums(image):
    """(sum, number) -> (mean, number), pixels without samples stay zero."""
    n = image[..., 3:4]
    image[..., :3] /= n.clip(1) if hasattr(n, "clip") and not hasattr(n, "clamp") else n.clamp(min=1)
    return image


def reduce_seed_sums(tensor, dist, dst=0, async_op=False):
    """One sum-reduce of the accumulation buffer (the rank's whole-frame sums) to `dst`: north_star's collective."""
    return dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
