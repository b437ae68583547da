"""BASELINE.json's configurations at their stated sizes on one MI355X, through the C ABI.

The oracle cannot afford whole frames of these sizes, so every test has two parts: (1) a WINDOW of the full frame —
the very pixels the full frame has there, same rays, same RNG keys (oracle_py render(window=...)) — compared with the
oracle bit for bit (ids) and to the north-star tolerance (radiance), and (2) size-independent properties of the WHOLE
frame: finite, sample count = seeds everywhere, the pixel-tile shards of world 8 sum to the unsharded frame bit for bit.
"""
import numpy as np
import pytest

from stratum_amd import camera, scenes, shard

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b**2).sum()), 1e-300))


def _device_frame(r, frame, seed_count, torch, aovs=True, packed=False):
    """One sthip_render call with device outputs; returns the torch tensors."""
    W, H = frame.width, frame.height
    out = {"radiance": torch.zeros((r.shard_slot_count(frame), 4) if packed else (H, W, 4), dtype=torch.float32, device="cuda"), "ray_count": torch.zeros(2, dtype=torch.int64, device="cuda")}
    if aovs:
        out["visibility"] = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
        out["albedo"] = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    r.render(frame, 0, seed_count, device_outputs={k: v.data_ptr() for k, v in out.items()}, packed_tiles=packed)
    torch.cuda.synchronize()
    return out


def _check_window(o, r, frame, seeds, window, got_radiance, got_ids):
    x0, y0, x1, y1 = window
    ref = o.render(frame, r.push_constants(frame), r.mSamplingFlags, 0, seeds, threads=0, window=window)
    gr = got_radiance[y0:y1, x0:x1]
    rr = ref["radiance"][y0:y1, x0:x1]
    assert np.array_equal(got_ids[y0:y1, x0:x1], ref["visibility"]["instance_primitive_index"][y0:y1, x0:x1])
    d = rel_l2(gr, rr)
    nd = int((gr.view(np.uint32) != rr.view(np.uint32)).any(axis=-1).sum())
    print("window %s x %d seeds: rel-L2 %.3e, pixels that differ in any bit %d of %d, oracle rays %d" % (window, seeds, d, nd, (x1 - x0) * (y1 - y0), int(ref["ray_count"][0])))
    assert (rr[..., 3] == seeds).all()
    assert d <= 1e-4  # north_star tolerance on the HDR framebuffer
    return ref


def test_config5_instanced_forest_4k_16_seeds_8_bounces(built):
    """configs[4]: 10M-triangle instanced forest (1000 instances x 10K-triangle trees over a two-level BVH), 3840x2160,
    seeds 0..15, maxDiffuseVertices 8 / maxPathVertices 10 / minPathVertices 4, ~coherentrr."""
    import torch

    from oracle import oracle_py
    from stratum_amd.bdpt import BDPT

    sc, cam = scenes.forest()
    assert sc.triangle_count > 9_000_000 and sc.instances.shape[0] >= 1000
    args = {"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]}
    W, H, seeds = 3840, 2160, 16
    frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    r = BDPT(device=0, args=args)
    try:
        r.update(sc)
        full = _device_frame(r, frame, seeds, torch)
        rad = full["radiance"].cpu().numpy()
        ids = full["visibility"].cpu().numpy().view(np.uint32)[..., 0]
        rays = full["ray_count"].cpu().numpy()
        # (2) whole-frame properties
        assert np.isfinite(rad).all()
        assert (rad[..., 3] == seeds).all()
        assert rad[..., :3].min() >= 0 and rad[..., :3].mean() > 1e-3
        assert rays[0] > rays[1] >= W * H * seeds  # every pixel-sample traces at least its primary ray
        print("config 5: %d rays (%d path rays) for %d pixel-samples" % (rays[0], rays[1], W * H * seeds))
        # (1) windows of the full frame against the oracle: image centre (dense canopy), lower left (ground + trunks), horizon
        o = oracle_py.OracleScene(sc)
        for window in ((1824, 1016, 2016, 1144), (64, 1900, 256, 2028), (3000, 300, 3128, 396)):
            _check_window(o, r, frame, seeds, window, rad, ids)
        # world 8: the shards of the 8 ranks, rendered one after the other on this GPU, sum to the full frame bit for bit,
        # and their ray counts add up to the full frame's
        acc = torch.zeros_like(full["radiance"])
        shard_rays = np.zeros(2, np.int64)
        for rank in range(8):
            r.set_shard(rank, 8, 64, 32)
            part = _device_frame(r, frame, seeds, torch, aovs=False)
            owned = torch.from_numpy(shard.owned_mask(W, H, rank, 8)).cuda()
            assert torch.count_nonzero(part["radiance"][~owned]) == 0  # zero (alpha included) outside its own tiles
            acc += part["radiance"]
            shard_rays += part["ray_count"].cpu().numpy()
        r.set_shard(0, 1, 64, 32)
        assert torch.equal(acc, full["radiance"])
        assert np.array_equal(shard_rays, rays)
    finally:
        r.close()


def test_config2_cornell_1080p_64_seeds(built, cornell):
    """configs[1]: Cornell box, 1920x1080, seeds 0..63 (running mean A1), default flags."""
    import torch

    from oracle import oracle_py
    from stratum_amd.bdpt import BDPT

    sc, cam = cornell
    W, H, seeds = 1920, 1080, 64
    frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    r = BDPT(device=0)
    try:
        r.update(sc)
        full = _device_frame(r, frame, seeds, torch)
        rad = full["radiance"].cpu().numpy()
        ids = full["visibility"].cpu().numpy().view(np.uint32)[..., 0]
        assert np.isfinite(rad).all() and (rad[..., 3] == seeds).all()
        o = oracle_py.OracleScene(sc)
        # a full-width band through the boxes and the light's reflection, and a block around the light itself
        _check_window(o, r, frame, seeds, (0, 520, 1920, 552), rad, ids)
        _check_window(o, r, frame, seeds, (832, 0, 1088, 64), rad, ids)
        # the frame does not depend on how many seeds share a pass (k_resolve folds them in seed order)
        r.set_option("max_paths_in_flight", W * H)  # one seed per pass instead of two
        again = _device_frame(r, frame, seeds, torch, aovs=False)
        assert torch.equal(again["radiance"], full["radiance"])
        assert torch.equal(again["ray_count"], full["ray_count"])
    finally:
        r.close()


def test_config4_per_rank_work_of_8_gpus_256_seeds(built):
    """configs[3] on one GPU: the 1M-triangle atrium at 1920x1080 with seeds 0..255, as the 8 ranks of the tile shard
    would render it (packed tiles, the form bench.py gathers over RCCL), one rank after the other, assembled with
    sthip_assemble_tiles — equal to the unsharded 256-seed frame bit for bit; a window of it against the oracle."""
    import torch

    from oracle import oracle_py
    from stratum_amd.bdpt import BDPT

    sc, cam = scenes.atrium()
    W, H, seeds, world = 1920, 1080, 256, 8
    frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    r = BDPT(device=0)
    try:
        r.update(sc)
        full = _device_frame(r, frame, seeds, torch)
        stride = shard.slot_count(W, H, 0, world)
        gathered = torch.zeros((world, stride, 4), dtype=torch.float32, device="cuda")
        rays = np.zeros(2, np.int64)
        for rank in range(world):
            r.set_shard(rank, world, 64, 32)
            part = _device_frame(r, frame, seeds, torch, aovs=False, packed=True)
            n = part["radiance"].shape[0]
            assert n == shard.slot_count(W, H, rank, world) <= stride
            gathered[rank, :n] = part["radiance"]
            rays += part["ray_count"].cpu().numpy()
        assembled = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        r.assemble_tiles(frame, gathered.data_ptr(), stride, assembled.data_ptr())
        torch.cuda.synchronize()
        r.set_shard(0, 1, 64, 32)
        assert torch.equal(assembled, full["radiance"])
        assert np.array_equal(rays, full["ray_count"].cpu().numpy())
        rad = full["radiance"].cpu().numpy()
        assert np.isfinite(rad).all() and (rad[..., 3] == seeds).all()
        ids = full["visibility"].cpu().numpy().view(np.uint32)[..., 0]
        o = oracle_py.OracleScene(sc)
        _check_window(o, r, frame, seeds, (928, 508, 992, 540), rad, ids)  # one 64x32 tile at the image centre
    finally:
        r.close()
