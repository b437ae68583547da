"""Steps after the path: tonemap, image metric, HDR export.
CPU: the oracle's restatement against closed-form float64 values, the HDR writer against bytes produced by the
reference's own vendored writer (tests/golden/hdr_writer.npz). GPU: HIP kernels == oracle, bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle_py
from stratum_amd import _lib, wire
from stratum_amd.post import ImageComparer, Tonemapper, write_hdr

HERE = os.path.dirname(os.path.abspath(__file__))
T = wire.TONEMAP


def hdr_image(h=37, w=53, seed=5):
    """Radiance-like test data: several decades of dynamic range, some zeros, one NaN and one negative pixel."""
    rng = np.random.default_rng(seed)
    img = np.exp(rng.normal(0, 2.5, (h, w, 4))).astype(np.float32)
    img[..., 3] = 1
    img[0, :5, :3] = 0
    img[1, 1, 0] = np.nan
    img[2, 2, :3] = [-0.5, 0.25, 0.1]
    alb = rng.random((h, w, 4)).astype(np.float32)
    alb[3, :7, :3] = 0
    return img, alb


def lum64(c):
    return c[..., 0] * 0.2126 + c[..., 1] * 0.7152 + c[..., 2] * 0.0722


def curve64(mode, c, cmax, lmax):
    """float64 closed forms of tonemap.hlsl:21-102, written from the formulas' published definitions."""
    l = lum64(c)[..., None]
    if mode == T["Raw"]:
        return c
    if mode == T["Reinhard"]:
        tc = c / (1 + c)
        a = c / (1 + l)
        return a + tc * (tc - a)
    if mode == T["ReinhardExtended"]:
        m = np.where(cmax == 0, 1, cmax)
        return c / (1 + c) * (1 + c / m**2)
    if mode == T["ReinhardLuminance"]:
        return c * ((l / (1 + l)) / l)
    if mode == T["ReinhardLuminanceExtended"]:
        m = lmax if lmax != 0 else 1
        return c * (((l / (1 + l)) * (1 + l / m**2)) / l)
    if mode == T["Uncharted2"]:
        f = lambda x: ((x * (0.15 * x + 0.05) + 0.004) / (x * (0.15 * x + 0.5) + 0.06)) - 0.02 / 0.3
        return f(c) / f(lmax if lmax != 0 else 1)
    if mode == T["Filmic"]:
        x = np.maximum(0, c - 0.004)
        return (x * (6.2 * x + 0.5)) / (x * (6.2 * x + 1.7) + 0.06)
    if mode == T["ACES"]:
        mi = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
        mo = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
        v = c @ mi.T
        v = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
        return np.clip(v @ mo.T, 0, 1)
    if mode == T["ACESApprox"]:
        v = c * 0.6
        return np.clip((v * (2.51 * v + 0.03)) / (v * (2.43 * v + 0.59) + 0.14), 0, 1)
    if mode in (T["ViridisR"], T["ViridisLengthRGB"]):
        x = np.clip(l[..., 0] / ((lmax if lmax != 0 else 1) if mode == T["ViridisLengthRGB"] else 1), 0, 1)
        k = np.array(
            [
                [0.280268003, -0.143510503, 2.225793877, -14.815088879, 25.212752309, -11.772589584],
                [-0.002117546, 1.617109353, -1.909305070, 2.701152864, -1.685288385, 0.178738871],
                [0.300805501, 2.614650302, -12.019139090, 28.933559110, -33.491294770, 13.762053843],
            ]
        )
        p = np.stack([x**i for i in range(6)], -1)
        return p @ k.T
    raise AssertionError(mode)


def srgb64(c):
    with np.errstate(invalid="ignore"):
        return np.where(c <= 0.0031308, c * 12.92, np.power(np.maximum(c, 1e-30) * 1.055, 1 / 2.4) - 0.055)


@pytest.mark.parametrize("mode", range(len(wire.TONEMAP_MODES)))
@pytest.mark.parametrize("modulate,gamma,exposure", [(False, False, 0.0), (True, True, 1.5), (False, True, -2.0)])
def test_oracle_tonemap_known_answers(mode, modulate, gamma, exposure):
    img, alb = hdr_image()
    img = np.abs(np.nan_to_num(img, nan=1.0))  # closed forms below assume finite non-negative radiance
    out, mx = oracle_py.tonemap(img, alb, mode, modulate, gamma, exposure)
    c = img[..., :3].astype(np.float64)
    red = c * alb[..., :3] if modulate else c
    lred = lum64(red)
    ok = lred > 0
    cmax = np.array([np.floor(red[..., k][ok] * 16384).max() / 16384 for k in range(3)])
    lmax = np.floor(lred[ok] * 16384).max() / 16384
    assert np.allclose(mx[:3], cmax, rtol=1e-6, atol=1 / 16384) and abs(mx[3] - lmax) <= max(1 / 16384, 1e-6 * lmax)
    if modulate:
        c = c * (np.float64(np.float32(1e-2)) + alb[..., :3])
    c = c * 2.0**exposure
    with np.errstate(invalid="ignore", divide="ignore"):
        want = curve64(mode, c, mx[:3].astype(np.float64), float(mx[3]))
        if gamma:
            want = srgb64(want)
    good = np.isfinite(want).all(-1)
    assert good.mean() > 0.95
    assert np.allclose(out[..., :3][good], want[good], rtol=2e-5, atol=2e-6)
    assert np.all(out[..., 3] == 1)


def test_oracle_tonemap_max_ignores_nan_and_dark_pixels():
    img = np.zeros((2, 4, 4), np.float32)
    img[0, 0, :3] = [np.nan, 9, 9]
    img[0, 1, :3] = [1.0, 0.5, 0.25]
    img[1, 2, :3] = [0.25, 2.0, 0.125]
    img[1, 3, :3] = [-5.0, 0.0, 0.0]  # luminance <= 0: skipped
    _, mx = oracle_py.tonemap(img, None, T["ReinhardExtended"], False, False, 0.0)
    assert mx[:3].tolist() == [1.0, 2.0, 0.25]
    assert abs(mx[3] - (0.25 * 0.2126 + 2.0 * 0.7152 + 0.125 * 0.0722)) < 1e-4


def test_oracle_image_compare_known_answers():
    h, w = 16, 24
    a = np.full((h, w, 4), 0.5, np.float32)
    b = np.full((h, w, 4), 0.25, np.float32)
    q = 1 << 20
    # MSE: every channel differs by 0.25 -> mean squared error 0.0625; 6 groups of 64 pixels, truncation < 1 each
    s, ovf = oracle_py.image_compare(a, b, wire.COMPARE["MSE"], q)
    assert not ovf and abs(s / q - 0.0625) < 8 / q
    # SMAPE: |d| / (|a|+|b|) = 1/3 per channel
    s, _ = oracle_py.image_compare(a, b, wire.COMPARE["SMAPE"], q)
    assert abs(s / q - 1 / 3) < 8 / q
    # Average: signed mean difference; a negative mean truncates to 0 per group
    s, _ = oracle_py.image_compare(a, b, wire.COMPARE["Average"], q)
    assert abs(s / q - 0.25) < 8 / q
    s, _ = oracle_py.image_compare(b, a, wire.COMPARE["Average"], q)
    assert s == 0
    # identical images
    assert oracle_py.image_compare(a, a, wire.COMPARE["MSE"], q) == (0, False)
    # overflow: a huge error with the maximal quantisation
    big = np.full((h, w, 4), 1e6, np.float32)
    _, ovf = oracle_py.image_compare(big, b, wire.COMPARE["MSE"], 0xFFFFFFFF)
    assert ovf
    # a ragged tail (pixel count not a multiple of 64) counts every pixel exactly once
    a2 = np.full((5, 13, 4), 1.0, np.float32)
    b2 = np.zeros((5, 13, 4), np.float32)
    s, _ = oracle_py.image_compare(a2, b2, wire.COMPARE["MSE"], q)
    assert abs(s / q - 1.0) < 4 / q


# ---- HDR export: bytes against the reference's own writer ----
def _golden():
    return np.load(os.path.join(HERE, "golden", "hdr_writer.npz"))


@pytest.mark.parametrize("name", ["flat_5x3", "noise_8x4", "runs_300x6", "literals_200x2"])
def test_hdr_writer_matches_reference_bytes(built, tmp_path, name):
    g = _golden()
    path = tmp_path / (name + ".hdr")
    write_hdr(path, g[name + "_in"])
    got = np.frombuffer(path.read_bytes(), np.uint8)
    assert got.size == g[name + "_hdr"].size and np.array_equal(got, g[name + "_hdr"])


def _decode_hdr(blob):
    """Minimal Radiance reader (header, flat or new-RLE rows) -> (H, W, 4) uint8 RGBE."""
    head, _, rest = blob.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE") and b"FORMAT=32-bit_rle_rgbe" in head
    line, _, data = rest.partition(b"\n")
    t = line.split()
    assert t[0] == b"-Y" and t[2] == b"+X"
    h, w = int(t[1]), int(t[3])
    out = np.zeros((h, w, 4), np.uint8)
    p = 0
    for y in range(h):
        if 8 <= w < 32768:
            assert data[p : p + 4] == bytes([2, 2, w >> 8, w & 255])
            p += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = data[p]
                    p += 1
                    if n > 128:
                        out[y, x : x + n - 128, c] = data[p]
                        p += 1
                        x += n - 128
                    else:
                        out[y, x : x + n, c] = np.frombuffer(data[p : p + n], np.uint8)
                        p += n
                        x += n
                assert x == w
        else:
            out[y] = np.frombuffer(data[p : p + 4 * w], np.uint8).reshape(w, 4)
            p += 4 * w
    assert p == len(data)
    return out


def test_hdr_writer_round_trip_of_a_rendered_like_image(built, tmp_path):
    img, _ = hdr_image(31, 67, seed=9)
    img = np.abs(np.nan_to_num(img, nan=0.0))
    path = tmp_path / "x.hdr"
    write_hdr(path, img)
    rgbe = _decode_hdr(path.read_bytes())
    val = rgbe[..., :3].astype(np.float64) * np.exp2(rgbe[..., 3:4].astype(np.float64) - 136)
    m = img[..., :3].max(-1, keepdims=True).astype(np.float64)
    # truncating 8-bit mantissa under a shared exponent: error below one step of the largest channel
    assert np.all(np.abs(val - img[..., :3]) <= m / 128 + 1e-30)


def test_hdr_writer_against_reference_build_when_present(built, tmp_path):
    ref_path = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libstbiw_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (the reference only exists in the build container)")
    ref = C.CDLL(ref_path)
    ref.stbi_write_hdr.restype = C.c_int
    ref.stbi_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(3)
    for w, h in [(1, 1), (7, 2), (8, 1), (129, 3), (640, 8)]:
        img = np.exp(rng.normal(0, 3, (h, w, 4))).astype(np.float32)
        img[rng.random((h, w)) < 0.4] = 0.125  # plenty of runs
        a, b = tmp_path / "a.hdr", tmp_path / "b.hdr"
        write_hdr(a, img)
        assert ref.stbi_write_hdr(str(b).encode(), w, h, 4, img.ctypes.data) == 1
        assert a.read_bytes() == b.read_bytes(), (w, h)


def test_hdr_writer_rejects_bad_arguments(built, tmp_path):
    L = _lib.lib()
    img = np.zeros((2, 2, 4), np.float32)
    assert L.sthip_write_hdr(None, 2, 2, img.ctypes.data) != 0
    assert L.sthip_write_hdr(str(tmp_path / "a.hdr").encode(), 0, 2, img.ctypes.data) != 0
    assert L.sthip_write_hdr(str(tmp_path / "no_such_dir" / "a.hdr").encode(), 2, 2, img.ctypes.data) != 0
    with pytest.raises(ValueError):
        write_hdr(tmp_path / "b.hdr", np.zeros((2, 2, 3), np.float32))


# ---- GPU: kernels against the oracle ----
@pytest.fixture(scope="module")
def gpu_ctx(built):
    from stratum_amd.bdpt import BDPT

    r = BDPT(device=0)
    yield r
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("modulate,gamma,exposure", [(False, False, 0.0), (True, True, 1.5), (False, True, -2.25), (True, False, 0.3)])
def test_gpu_tonemap_equals_oracle(gpu_ctx, modulate, gamma, exposure):
    img, alb = hdr_image(211, 317, seed=11)
    for mode, name in enumerate(wire.TONEMAP_MODES):
        tm = Tonemapper(gpu_ctx, name, exposure, gamma)
        got, gmax = tm(img, alb, modulate_albedo=modulate, return_max=True)
        want, wmax = oracle_py.tonemap(img, alb, mode, modulate, gamma, exposure)
        assert np.array_equal(gmax, wmax), name
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s: %d pixels differ" % (name, (got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())


@pytest.mark.gpu
def test_gpu_tonemap_of_a_rendered_frame(gpu_ctx, cornell):
    from stratum_amd import camera

    sc, cam = cornell
    frame = camera.Frame(160, 96, cam["fovy"], cam["eye"], cam["target"])
    gpu_ctx.update(sc)
    res = gpu_ctx.render(frame, 0, 4)
    for name in ("Reinhard", "Uncharted2", "ACES", "ViridisLengthRGB"):
        got = Tonemapper(gpu_ctx, name, 0.5, True)(res["radiance"], res["albedo"], modulate_albedo=True)
        want, _ = oracle_py.tonemap(res["radiance"], res["albedo"], wire.TONEMAP[name], True, True, 0.5)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        assert np.isfinite(got).all() and got[..., :3].max() > 0.05


@pytest.mark.gpu
def test_gpu_tonemap_device_pointers(gpu_ctx):
    import torch

    img, alb = hdr_image(64, 96, seed=2)
    t_in, t_alb = torch.from_numpy(img).cuda(), torch.from_numpy(alb).cuda()
    t_out = torch.empty_like(t_in)
    tm = Tonemapper(gpu_ctx, "Filmic", 1.0, True)
    torch.cuda.synchronize()
    tm.device(96, 64, t_in.data_ptr(), t_alb.data_ptr(), t_out.data_ptr(), modulate_albedo=True)
    gpu_ctx.stats()  # joins the context's stream
    torch.cuda.synchronize()
    want, _ = oracle_py.tonemap(img, alb, wire.TONEMAP["Filmic"], True, True, 1.0)
    assert np.array_equal(t_out.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_gpu_image_compare_equals_oracle(gpu_ctx):
    a, _ = hdr_image(101, 77, seed=21)
    b, _ = hdr_image(101, 77, seed=22)
    a, b = np.abs(np.nan_to_num(a, nan=0.5)), np.abs(np.nan_to_num(b, nan=0.5))
    a, b = np.minimum(a, 50), np.minimum(b, 50)
    for mode, name in enumerate(wire.COMPARE_MODES):
        for q in (1024, 1 << 20):
            ic = ImageComparer(gpu_ctx, name, q)
            assert ic.raw(a, b) == oracle_py.image_compare(a, b, mode, q), (name, q)
    assert ImageComparer(gpu_ctx, "MSE").raw(a, a) == (0, False)
    big = np.full_like(a, 1e6)
    assert ImageComparer(gpu_ctx, "MSE", 0xFFFFFFFF).raw(big, a)[1] is True
    assert oracle_py.image_compare(big, a, 1, 0xFFFFFFFF)[1] is True
    v = ImageComparer(gpu_ctx, "MSE", 1 << 20).value(a, b)
    ref = np.sqrt(((a[..., :3].astype(np.float64) - b[..., :3]) ** 2).mean())
    assert abs(v - ref) / ref < 1e-3


@pytest.mark.gpu
def test_gpu_post_error_reporting(gpu_ctx):
    img, _ = hdr_image(8, 8)
    with pytest.raises(_lib.StratumHipError, match="unknown mode"):
        Tonemapper(gpu_ctx, 99)(img)
    with pytest.raises(_lib.StratumHipError, match="gAlbedo"):
        Tonemapper(gpu_ctx, "Raw")(img, None, modulate_albedo=True)
    with pytest.raises(_lib.StratumHipError, match="unknown metric"):
        ImageComparer(gpu_ctx, 7).raw(img, img)


# ---- temporal accumulation (temporal_accumulation.hlsl) ----
def _oracle_frames(n_frames, moving, w=96, h=64):
    """Renders n_frames of the Cornell box with the oracle: one seed per frame, optionally a camera that slides."""
    from stratum_amd import camera, scenes

    sc, cam = scenes.cornell_box()
    o = oracle_py.OracleScene(sc)
    frames, outs, prev = [], [], None
    for k in range(n_frames):
        eye = np.array(cam["eye"]) + (np.array([0.05, 0.02, 0.0]) * k if moving else 0)
        fr = camera.Frame(w, h, cam["fovy"], tuple(eye), cam["target"], prev=prev)
        outs.append(o.render(fr, wire.default_push_constants(w, h, sc.light_count), wire.DEFAULT_SAMPLING_FLAGS, k, 1, threads=4))
        frames.append(fr)
        prev = fr
    return sc, frames, outs


def _empty_history(h, w):
    vis = np.zeros((h, w), wire.VisibilityInfo)
    vis["instance_primitive_index"] = wire.MISS
    return {"accum_color": np.zeros((h, w, 4), np.float32), "accum_moments": np.zeros((h, w, 2), np.float32), "visibility": vis, "depth": np.zeros((h, w), wire.DepthInfo)}


def test_oracle_accumulation_without_reprojection_is_the_running_mean():
    _, frames, outs = _oracle_frames(4, moving=False, w=48, h=32)
    hist = _empty_history(32, 48)
    for out in outs:
        c, m = oracle_py.accumulate(out, hist, frames[0].views, reprojection=False)
        hist = {"accum_color": c, "accum_moments": m, "visibility": out["visibility"], "depth": out["depth"]}
    mean = np.mean([o["radiance"][..., :3].astype(np.float64) for o in outs], axis=0)
    assert np.all(c[..., 3] == 4)
    assert np.allclose(c[..., :3], mean, rtol=2e-6, atol=1e-7)
    lum = [o["radiance"][..., :3].astype(np.float64) @ np.array([0.2126, 0.7152, 0.0722]) for o in outs]
    assert np.allclose(m[..., 0], np.mean(lum, axis=0), rtol=1e-5, atol=1e-7)
    assert np.allclose(m[..., 1], np.mean(np.square(lum), axis=0), rtol=1e-5, atol=1e-7)
    # gHistoryLimit caps the sample count, i.e. turns the mean into an exponential average with alpha = 1 / limit
    c2, _ = oracle_py.accumulate(outs[0], hist, frames[0].views, reprojection=False, history_limit=2.0)
    assert np.all(c2[..., 3] == 2)
    want = 0.5 * c[..., :3].astype(np.float64) + 0.5 * outs[0]["radiance"][..., :3]
    assert np.allclose(c2[..., :3], want, rtol=2e-6, atol=1e-7)
    # a NaN sample is dropped, not accumulated
    bad = {k: v.copy() for k, v in outs[1].items() if hasattr(v, "copy")}
    bad["radiance"][3, 5, 0] = np.nan
    c3, _ = oracle_py.accumulate(bad, hist, frames[0].views, reprojection=False)
    assert np.isfinite(c3).all() and c3[3, 5, 3] == 4


def test_oracle_reprojection_static_and_moving_camera():
    _, frames, outs = _oracle_frames(3, moving=False, w=48, h=32)
    hist = _empty_history(32, 48)
    hist_n = _empty_history(32, 48)
    for out in outs:
        c, m = oracle_py.accumulate(out, hist, frames[0].views, reprojection=True)
        cn, mn = oracle_py.accumulate(out, hist_n, frames[0].views, reprojection=False)
        hist = {"accum_color": c, "accum_moments": m, "visibility": out["visibility"], "depth": out["depth"]}
        hist_n = {"accum_color": cn, "accum_moments": mn, "visibility": out["visibility"], "depth": out["depth"]}
    hit = outs[-1]["visibility"]["instance_primitive_index"] != wire.MISS
    # a static camera reprojects every surface pixel onto itself (weight 1 on one tap): same result as without
    assert hit.mean() > 0.5
    same = np.isclose(c[..., 3], 3) & hit
    assert same.sum() > 0.97 * hit.sum()
    assert np.allclose(c[same], cn[same], rtol=1e-5, atol=1e-6)
    # pixels that see nothing have no history to reproject: they restart
    assert np.all(c[~hit][:, 3] == 1)
    # moving camera: most pixels keep a history, disocclusions and frame edges restart at n = 1
    _, frames, outs = _oracle_frames(3, moving=True, w=48, h=32)
    hist = _empty_history(32, 48)
    for fr, out in zip(frames, outs):
        c, m = oracle_py.accumulate(out, hist, fr.views, reprojection=True)
        hist = {"accum_color": c, "accum_moments": m, "visibility": out["visibility"], "depth": out["depth"]}
    n = c[..., 3]
    assert (n > 2.5).mean() > 0.5 and (n == 1).mean() > 0.02 and n.max() <= 3.0001


@pytest.mark.gpu
@pytest.mark.parametrize("moving,reprojection,demodulate,limit", [(False, False, False, 0.0), (False, True, False, 0.0), (True, True, False, 0.0), (True, True, True, 2.0)])
def test_gpu_accumulation_equals_oracle(gpu_ctx, moving, reprojection, demodulate, limit):
    from stratum_amd.post import TemporalAccumulation

    _, frames, outs = _oracle_frames(3, moving=moving)
    acc = TemporalAccumulation(gpu_ctx, reprojection, demodulate, limit)
    hist = _empty_history(64, 96)
    for fr, out in zip(frames, outs):
        gc, gm = acc(out, fr.views)
        oc, om = oracle_py.accumulate(out, hist, fr.views, reprojection, demodulate, limit)
        hist = {"accum_color": oc, "accum_moments": om, "visibility": out["visibility"], "depth": out["depth"]}
        assert np.array_equal(gc.view(np.uint32), oc.view(np.uint32))
        assert np.array_equal(gm.view(np.uint32), om.view(np.uint32))
    assert gc[..., 3].max() == (limit if limit else 3)


@pytest.mark.gpu
def test_gpu_accumulation_errors(gpu_ctx):
    from stratum_amd.post import TemporalAccumulation

    _, frames, outs = _oracle_frames(1, moving=False, w=32, h=32)
    out = {"radiance": outs[0]["radiance"]}  # no AOVs
    with pytest.raises(_lib.StratumHipError, match="gReprojection needs"):
        TemporalAccumulation(gpu_ctx, reprojection=True)(out, frames[0].views)
    with pytest.raises(_lib.StratumHipError, match="gAlbedo"):
        TemporalAccumulation(gpu_ctx, reprojection=False, demodulate_albedo=True)(out, frames[0].views)
    c, _ = TemporalAccumulation(gpu_ctx, reprojection=False)(out, frames[0].views)
    assert np.array_equal(c, outs[0]["radiance"])


@pytest.mark.gpu
def test_exposure_smoothing_over_frames():
    """gExposureAlpha (tonemap.hlsl:168-182): the maxima the extended curves use are blended with the previous frame's
    (the state the reference keeps in gMax / gPrevMax); three frames of changing brightness against the oracle."""
    from oracle import oracle_py
    from stratum_amd.bdpt import BDPT
    from stratum_amd.post import Tonemapper

    rng = np.random.default_rng(5)
    r = BDPT(device=0)
    try:
        tm = Tonemapper(r, "ReinhardLuminanceExtended", 0.5, True, exposure_alpha=0.3)
        state = np.zeros(6, np.float32)
        for k, gain in enumerate((1.0, 20.0, 0.2)):
            img = (np.abs(rng.normal(size=(40, 56, 4))) * gain).astype(np.float32)
            got, gmax = tm(img, return_max=True)
            ref, rmax = oracle_py.tonemap_state(img, state, None, wire.TONEMAP["ReinhardLuminanceExtended"], False, True, 0.5, 0.3)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), k
            assert np.array_equal(tm.state.view(np.uint32), state.view(np.uint32)), k
        raw = Tonemapper(r, "ReinhardLuminanceExtended", 0.5, True)(img)
        assert not np.array_equal(raw, got)  # the third frame is dark: its own maximum alone gives another image
        assert gmax[3] < state[3]  # ... because the blended maximum still remembers the bright frame
    finally:
        r.close()


def test_exposure_smoothing_arithmetic():
    img = np.full((4, 4, 4), 2.0, np.float32)
    state = np.zeros(6, np.float32)
    orc_out, _ = oracle_py_tonemap_state(img, state, 0.25)
    lum = np.float32(2.0) * np.float32(0.2126) + np.float32(2.0) * np.float32(0.7152) + np.float32(2.0) * np.float32(0.0722)
    assert abs(state[3] - lum) < 1e-3 and abs(state[5] - lum * lum) < 1e-2  # first frame: nothing to blend with
    img2 = np.full((4, 4, 4), 4.0, np.float32)
    prev = state.copy()
    oracle_py_tonemap_state(img2, state, 0.25)
    assert abs(state[3] - (prev[3] + 0.25 * (2 * prev[3] - prev[3]))) < 1e-3  # lerp(prev, cur, alpha), cur = 2 x prev
    assert abs(state[4] - (prev[4] + 0.5 * (2 * prev[4] - prev[4]))) < 1e-3  # moments with sqrt(alpha)


def oracle_py_tonemap_state(img, state, alpha):
    from oracle import oracle_py

    return oracle_py.tonemap_state(img, state, None, wire.TONEMAP["Reinhard"], False, True, 0.0, alpha)
