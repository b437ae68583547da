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
