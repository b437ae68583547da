"""The N > 1 path on CPU: world-size-2 gloo. Each rank produces a framebuffer that is zero outside the
tiles it owns (here the oracle stands in for the renderer, because the product has no CPU path), and the
sum-reduce of stratum_amd.shard assembles the frame exactly."""
import os

import numpy as np
import pytest

from stratum_amd import shard


def test_owner_map_partitions_the_frame():
    for (w, h, world, tw, th) in [(1920, 1080, 8, 64, 32), (256, 256, 2, 64, 32), (100, 70, 3, 16, 8), (64, 32, 4, 64, 32)]:
        om = shard.owner_map(w, h, world, tw, th)
        assert om.shape == (h, w) and om.min() >= 0 and om.max() < world
        total = sum(shard.owned_mask(w, h, r, world, tw, th).sum() for r in range(world))
        assert total == w * h
        # tile t (row-major) belongs to rank t % world: sthip_set_shard's rule
        tx, ty = shard.tile_grid(w, h, tw, th)
        for t in range(min(tx * ty, 50)):
            y, x = (t // tx) * th, (t % tx) * tw
            assert om[y, x] == t % world
    # balance at the bench configuration: every rank gets within 1 tile of the mean
    om = shard.owner_map(1920, 1080, 8)
    counts = np.bincount((om[::32, ::64]).ravel(), minlength=8)
    assert counts.max() - counts.min() <= 1


def test_packed_tiles_cover_the_frame_once():
    """slot_pixels (host mirror of slot_to_pixel): the packed buffers of all ranks hold every pixel exactly once, each
    on the rank owner_map names; assemble_tiles puts them back."""
    for (w, h, world, tw, th) in [(1920, 1080, 8, 64, 32), (100, 70, 3, 16, 8), (64, 32, 4, 64, 32), (96, 64, 2, 16, 8)]:
        om = shard.owner_map(w, h, world, tw, th)
        seen = np.zeros((h, w), np.int32)
        ref = np.random.RandomState(1).rand(h, w, 4).astype(np.float32)
        packed = []
        for r in range(world):
            xy = shard.slot_pixels(w, h, r, world, tw, th)
            assert xy.shape[0] == shard.slot_count(w, h, r, world, tw, th) <= shard.slot_count(w, h, 0, world, tw, th)
            ok = xy[:, 0] >= 0
            assert np.all(om[xy[ok, 1], xy[ok, 0]] == r)
            np.add.at(seen, (xy[ok, 1], xy[ok, 0]), 1)
            buf = np.zeros((shard.slot_count(w, h, 0, world, tw, th), 4), np.float32)  # equal-size messages
            buf[: xy.shape[0]][ok] = ref[xy[ok, 1], xy[ok, 0]]
            packed.append(buf)
        assert np.all(seen == 1)
        assert np.array_equal(shard.assemble_tiles(packed, w, h, tw, th), ref)


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    from oracle import oracle_py as orc
    from stratum_amd import camera, scenes, wire

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, cam = scenes.cornell_box()
    W, H = 96, 64
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    pc = wire.default_push_constants(W, H, sc.light_count)
    full = orc.OracleScene(sc).render(fr, pc, seed_begin=0, seed_count=world, threads=2, aovs=False)["radiance"]
    mine = full * shard.owned_mask(W, H, rank, world, 16, 8)[..., None]
    t = torch.from_numpy(mine.copy())
    shard.reduce_framebuffer(t, dist, dst=0)
    # the same frame through the packed exchange: every rank sends only its tiles, rank 0 gathers and scatters
    stride = shard.slot_count(W, H, 0, world, 16, 8)
    xy = shard.slot_pixels(W, H, rank, world, 16, 8)
    ok = xy[:, 0] >= 0
    packed = np.zeros((stride, 4), np.float32)
    packed[: xy.shape[0]][ok] = full[xy[ok, 1], xy[ok, 0]]
    gathered = torch.zeros((world, stride, 4)) if rank == 0 else None
    shard.gather_tiles(torch.from_numpy(packed), gathered, dist, dst=0)
    # the seed-split replica mode: every rank the whole frame for its own seeds, one sum-reduce of the accumulation buffer
    first, count = shard.seed_range(rank, world, 2 * world + 1)
    part = orc.OracleScene(sc).render(fr, pc, seed_begin=first, seed_count=count, threads=2, aovs=False)["radiance"]
    sums = torch.from_numpy(shard.to_sums(part.copy()))
    shard.reduce_seed_sums(sums, dist, dst=0)
    if rank == 0:
        np.save(os.path.join(tmp, "seed_split.npy"), shard.from_sums(sums.numpy()))
        np.save(os.path.join(tmp, "seed_split_ref.npy"), orc.OracleScene(sc).render(fr, pc, seed_begin=0, seed_count=2 * world + 1, threads=2, aovs=False)["radiance"])
    if rank == 0:
        np.save(os.path.join(tmp, "assembled.npy"), t.numpy())
        np.save(os.path.join(tmp, "gathered.npy"), shard.assemble_tiles([g.numpy() for g in gathered], W, H, 16, 8))
        np.save(os.path.join(tmp, "full.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_assembles_the_frame(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "assembled.npy")
    b = np.load(tmp_path / "full.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    g = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(g.view(np.uint32), b.view(np.uint32))
    # seed split: the same mean up to the order of a floating-point sum (~1e-7 relative), the sample counts exact
    s, sref = np.load(tmp_path / "seed_split.npy"), np.load(tmp_path / "seed_split_ref.npy")
    assert np.array_equal(s[..., 3], sref[..., 3]) and s[..., 3].max() == 5
    np.testing.assert_allclose(s[..., :3], sref[..., :3], rtol=2e-6, atol=1e-7)
    assert [shard.seed_range(r, 3, 7) for r in range(3)] == [(0, 3), (3, 2), (5, 2)] and shard.seed_range(2, 3, 1) == (1, 0)
