"""The C++ host side (stratum_amd/host/stratum_hip.hpp): Stratum's Node graph / Scene / BDPT surface above the
C ABI. tests/cpp/host_test.cpp plays main.cpp: it builds the node graph from a scene description, lets
Application fire OnUpdate / OnRenderWindow, and is checked against the Python packing and renderer."""
import os
import subprocess

import numpy as np
import pytest

from stratum_amd import camera, scenes, wire
from stratum_amd.scene import SceneBuilder, dump_description, rotate_y, scale, translate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_test")


@pytest.fixture(scope="module")
def host_test(built):
    src = os.path.join(ROOT, "tests", "cpp", "host_test.cpp")
    hdr = os.path.join(ROOT, "stratum_amd", "host", "stratum_hip.hpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(
            ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-o", EXE, src, "-L" + os.path.join(ROOT, "stratum_amd"), "-lstratum_hip", "-Wl,-rpath," + os.path.join(ROOT, "stratum_amd")]
        )
    return EXE


def foggy_cornell():
    import os

    return scenes.cornell_box(fog=np.load(os.path.join(os.path.dirname(__file__), "golden", "fog_sphere.npz"))["grid"], anisotropy=0.3)


def shared_mesh_scene():
    b = SceneBuilder("shared")
    m0 = b.add_material((0.7, 0.7, 0.7), roughness=0.3)
    m1 = b.add_material((0.2, 0.5, 0.9), metallic=0.8, roughness=0.2)
    li = b.add_emitter((9.0, 8.0, 7.0))
    q = scenes._quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1), (0, 1, 0))
    quad = b.add_mesh(*q, index_stride=2)
    for k in range(5):
        b.add_instance(quad, m0 if k % 2 else m1, translate((0.3 * k, 0.1 * k, -0.2 * k)) @ rotate_y(0.4 * k) @ scale((1.0 + 0.1 * k, 1.0, 0.7)))
    b.add_instance(quad, li, translate((0, 2.5, 0)) @ scale((0.5, 1.0, 0.5)) @ np.diag([1.0, -1.0, 1.0, 1.0]))
    return b.build(), {"eye": (0.5, 1.5, 4.0), "target": (0.5, 0.3, 0.0), "fovy": np.radians(50.0)}


@pytest.mark.parametrize("make", [scenes.cornell_box, shared_mesh_scene, scenes.textured_box, scenes.spheres_room, scenes.environment_scene, scenes.foliage, foggy_cornell])
def test_scene_update_packs_like_the_reference_layouts(host_test, tmp_path, make):
    sc, cam = make()
    fr = camera.Frame(64, 48, cam["fovy"], cam["eye"], cam["target"])
    desc = str(tmp_path / "scene.bin")
    dump_description(desc, sc, fr)
    out = subprocess.run([host_test, "pack", desc], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("PACK OK"), out.stdout + out.stderr


def bidirectional_cornell():
    return scenes.cornell_box()


BDPT_ARGS = {foggy_cornell: {"maxDiffuseVertices": 3}, bidirectional_cornell: {"bdptFlag": ["connectToLightPaths", "connecttoviews", "~deferShadowRays"], "maxDiffuseVertices": 3, "maxPathVertices": 6}}


@pytest.mark.gpu
@pytest.mark.parametrize("make", [scenes.cornell_box, shared_mesh_scene, scenes.textured_box, scenes.spheres_room, scenes.environment_scene, scenes.foliage, bidirectional_cornell, foggy_cornell])
def test_cpp_host_renders_what_the_python_host_renders(host_test, tmp_path, make):
    from stratum_amd.bdpt import BDPT
    from stratum_amd.post import Tonemapper, write_hdr

    sc, cam = make()
    args = BDPT_ARGS.get(make, {})
    argv = ["--%s=%s" % (k, x) for k, v in args.items() for x in (v if isinstance(v, list) else [v])]
    W, H, seeds = 96, 64, 3
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    desc, outp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    dump_description(desc, sc, fr)
    hdr = str(tmp_path / "image.hdr")
    out = subprocess.run([host_test, "render", desc, outp, str(seeds), str(wire.TONEMAP["ACES"]), "0.75", hdr] + argv, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("RENDER OK"), out.stdout + out.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    rad = raw[: W * H * 16].view(np.float32).reshape(H, W, 4)
    vis = raw[W * H * 16 : W * H * 24].view(wire.VisibilityInfo).reshape(H, W)
    rays = raw[W * H * 24 : W * H * 24 + 16].view(np.uint64)
    tm = raw[W * H * 24 + 16 :].view(np.float32).reshape(H, W, 4)
    r = BDPT(device=0, args=args)
    try:
        r.update(sc)
        ref = r.render(fr, 0, seeds)
        ref_tm = Tonemapper(r, "ACES", 0.75, True)(ref["radiance"])
    finally:
        r.close()
    assert np.array_equal(tm.view(np.uint32), ref_tm.view(np.uint32))
    write_hdr(tmp_path / "py.hdr", ref["radiance"])
    assert open(hdr, "rb").read() == (tmp_path / "py.hdr").read_bytes()
    assert np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32))
    assert np.array_equal(vis["instance_primitive_index"], ref["visibility"]["instance_primitive_index"])
    assert np.array_equal(rays, ref["ray_count"])


@pytest.mark.gpu
@pytest.mark.parametrize("flags, extra", [(["neereservoirs", "neereservoirreuse"], {"reservoirM": 2}),
                                          (["connecttolightpaths", "lightvertexcache", "lvcreservoirs", "lvcreservoirreuse"], {"lightPathCount": 3000, "reservoirM": 2, "maxDiffuseVertices": 3})])
def test_cpp_host_reuses_the_previous_frames_reservoirs(host_test, tmp_path, flags, extra):
    """A host that renders one frame per call: frame n looks into the hash grids frame n - 1 left (BDPT.cpp:482-483,621-627;
    `reuse_grids_persist`). The C++ host's third frame of an unchanged scene and camera is the third seed of the chain — the
    Python host's three one-seed calls with the option set, itself pinned to one call of three seeds in test_gpu_parity — and not
    the third seed rendered on its own."""
    from stratum_amd.bdpt import BDPT

    sc, cam = scenes.cornell_box()
    W, H = 96, 64
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    desc, outp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    dump_description(desc, sc, fr)
    argv = ["--bdptFlag=%s" % f for f in flags] + ["--%s=%s" % kv for kv in extra.items()] + ["--frames=3"]
    out = subprocess.run([host_test, "render", desc, outp, "1", "0", "0", str(tmp_path / "image.hdr")] + argv, capture_output=True, text=True)
    assert out.returncode == 0 and "FRAMES 3" in out.stdout, out.stdout + out.stderr
    raw = np.fromfile(outp + ".last", dtype=np.uint8)
    rad = raw[: W * H * 16].view(np.float32).reshape(H, W, 4)
    rays = raw[W * H * 16 :].view(np.uint64)
    r = BDPT(device=0, args=dict(extra, bdptFlag=flags))
    try:
        r.update(sc)
        alone = r.render(fr, 2, 1, aovs=False)
        r.set_option("reuse_grids_persist", 1)
        for seed in range(3):
            ref = r.render(fr, seed, 1, aovs=False)
    finally:
        r.close()
    assert np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32)) and np.array_equal(rays, ref["ray_count"])
    assert not np.array_equal(rad, alone["radiance"])


@pytest.mark.gpu
def test_cpp_host_moves_instances_with_a_top_level_rebuild(host_test, tmp_path):
    """Nodes move between two frames: Scene::update repacks (motion transforms from the previous frame's), BDPT::update
    finds only transforms changed and calls sthip_scene_update_transforms; the second frame and its prev-uv output equal
    what the Python host gets from SceneData.set_instance_transform + BDPT.update_transforms."""
    from stratum_amd.bdpt import BDPT
    from stratum_amd.scene import translate

    sc, cam = scenes.cornell_box()
    W, H, seeds = 96, 64, 2
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    desc, outp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    dump_description(desc, sc, fr)
    out = subprocess.run([host_test, "render", desc, outp, str(seeds), "0", "0", str(tmp_path / "x.hdr"), "--move=0.1,0.0,-0.15"], capture_output=True, text=True)
    assert out.returncode == 0 and "MOVED transforms_only=1" in out.stdout, out.stdout + out.stderr
    raw = np.fromfile(outp + ".moved", dtype=np.float32)
    rad, prev_uv = raw[: W * H * 4].reshape(H, W, 4), raw[W * H * 4 :].reshape(H, W, 2)
    r = BDPT(device=0)
    try:
        r.update(sc)
        r.render(fr, 0, seeds)
        moved = [i for i in range(sc.instances.shape[0]) if not np.array_equal(sc.transforms["m"][i], np.eye(4, dtype=np.float32)[:3])]
        assert len(moved) == 2  # the two blocks
        for i in moved:
            m = sc.transforms["m"][i].copy()
            m[:, 3] += np.array([0.1, 0.0, -0.15], np.float32)  # float32 adds, as the C++ node transform is moved
            sc.set_instance_transform(i, m)
        r.update_transforms(sc)
        ref = r.render(fr, seeds, seeds)  # the C++ host's frame number went on: seeds `seeds` .. 2 seeds - 1
    finally:
        r.close()
    assert np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32))
    assert np.array_equal(prev_uv.view(np.uint32), ref["prev_uv"].view(np.uint32))


# ---- the plugin boundary: `--plugin=libstratum_hip_plugin.so;stratum_hip_register` (main.cpp:11-24,148-149) ----
def _plugin_host(built, spec):
    import __graft_entry__ as g

    g.build_plugin()
    return subprocess.run([g.PLUGIN_HOST, spec], capture_output=True, text=True)


def test_plugin_loads_and_its_entry_point_runs(built):
    """dlopen(RTLD_NOW) of the plugin resolves every symbol (it links libstratum_hip.so by $ORIGIN), dlsym finds the entry
    point, and calling it with a stm::Node& constructs the renderer — which needs a HIP device: on a box without one the
    only acceptable outcome is the sthip_create error (exit 3), never a load failure or a missing symbol."""
    import __graft_entry__ as g

    out = _plugin_host(built, g.PLUGIN + ";stratum_hip_register")
    assert out.returncode in (0, 3), out.stdout + out.stderr
    assert ("PLUGIN OK" in out.stdout) if out.returncode == 0 else ("sthip_create" in out.stdout)


def test_plugin_loader_errors_are_the_references(built):
    """dynamic_library.hpp:29,50: a library that does not load throws runtime_error, a missing symbol invalid_argument."""
    import __graft_entry__ as g

    out = _plugin_host(built, g.PLUGIN + ";no_such_entry_point")
    assert out.returncode == 4 and "Could not find function no_such_entry_point" in out.stdout
    out = _plugin_host(built, "/nonexistent/libnope.so;stratum_hip_register")
    assert out.returncode == 5 and "Failed to load /nonexistent/libnope.so" in out.stdout


@pytest.mark.gpu
def test_plugin_installs_the_renderer_on_a_gpu(built):
    import __graft_entry__ as g

    out = _plugin_host(built, g.PLUGIN + ";stratum_hip_register")
    assert out.returncode == 0 and "PLUGIN OK" in out.stdout, out.stdout + out.stderr


def test_packed_node_planes_are_conservative(tmp_path):
    """The 48-byte node (bvh.h: BvhNodePacked) keeps its child references in the low mantissa byte of eight planes;
    tests/cpp/pack_test.cpp checks that pack_plane's outward rounding keeps every plane conservative for EVERY byte value,
    finite, within 511 ulp, and that references round-trip (200 k random + special values, host only)."""
    exe = str(tmp_path / "pack_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cpp", "pack_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("PACK OK"), out.stdout + out.stderr


# ---- the multi-GPU driver of the C++ host (host/stratum_hip_multi.hpp) ----
def _multi_host(built):
    import __graft_entry__ as g

    return g.build_multi_host()


@pytest.mark.parametrize("w,h,world,tw,th", [(1920, 1080, 8, 64, 32), (100, 70, 3, 16, 8), (64, 32, 4, 64, 32), (3840, 2160, 8, 64, 32)])
def test_cpp_shard_layout_is_the_python_and_device_layout(built, w, h, world, tw, th):
    """stm::ShardLayout (owner rule, slot -> pixel, host assembly) against stratum_amd.shard, the mirror the device
    kernels are tested against: same slot counts, same pixel per slot; packing by ownership and assembling is lossless.
    This is the N-rank logic of MultiDeviceBDPT without N GPUs."""
    from stratum_amd import shard

    out = subprocess.run([_multi_host(built), "layout", str(w), str(h), str(world), str(tw), str(th)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("LAYOUT OK"), out.stdout + out.stderr
    lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("rank")]
    assert len(lines) == world
    for rank, ln in enumerate(lines):
        assert int(ln[3]) == shard.slot_count(w, h, rank, world, tw, th)
        xy = shard.slot_pixels(w, h, rank, world, tw, th)
        s = 0
        for x, y in xy:  # the same FNV-style fold the program prints
            s = (s * 1099511628211 + (int(y) * w + int(x) + 1 if x >= 0 else 0)) & 0xFFFFFFFFFFFFFFFF
        assert int(ln[5]) == s


@pytest.fixture(scope="module")
def multi_mock_exe(tmp_path_factory):
    """tests/cpp/multi_mock.cpp: the C++ multi-GPU driver over stand-ins for the C ABI, HIP and RCCL (no GPU, no libraries)."""
    exe = str(tmp_path_factory.mktemp("multi_mock") / "multi_mock")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "multi_mock.cpp"), "-lpthread"])
    return exe


@pytest.mark.parametrize("world,w,h,tw,th", [(2, 192, 96, 64, 32), (3, 100, 70, 16, 8), (5, 200, 120, 64, 32), (8, 320, 200, 64, 32), (8, 64, 32, 64, 32)])
def test_cpp_multi_device_driver_at_world_2_to_8_over_stand_ins(multi_mock_exe, tmp_path, world, w, h, tw, th):
    """stm::MultiDeviceBDPT with N ranks and no GPU: its persistent rank threads, both phases of a frame, the packing of the
    G-buffer outputs, the send / recv exchange, the assembly through ShardLayout, frames in flight and seed bookkeeping run
    over host stand-ins for sthip_*, hip* and nccl* (tests/cpp/multi_mock.cpp). A rank that fails makes render() throw
    BEFORE any collective is posted (the old single-phase form left the other ranks waiting for ever), and the
    driver renders the next frame as if nothing had happened. Frames larger and smaller than a tile round, ragged edges."""
    sc, cam = scenes.cornell_box()
    fr = camera.Frame(w, h, cam["fovy"], cam["eye"], cam["target"])
    desc = str(tmp_path / "scene.bin")
    dump_description(desc, sc, fr)
    out = subprocess.run([multi_mock_exe, desc, str(world), str(tw), str(th)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and ("MULTI MOCK OK world %d" % world) in out.stdout, out.stdout + out.stderr[-2000:]


def test_cpp_multi_device_driver_under_thread_sanitizer(tmp_path):
    """The same program under ThreadSanitizer at world 4: the rank threads' job hand-over, the frames in flight and the
    failure path hold no data race."""
    exe = str(tmp_path / "multi_mock_tsan")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "multi_mock.cpp"), "-lpthread"])
    sc, cam = scenes.cornell_box()
    desc = str(tmp_path / "scene.bin")
    dump_description(desc, sc, camera.Frame(160, 100, cam["fovy"], cam["eye"], cam["target"]))
    out = subprocess.run([exe, desc, "4", "32", "16"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "MULTI MOCK OK world 4" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-3000:]


@pytest.mark.gpu
def test_cpp_multi_device_driver_seed_split_on_one_gpu(built, tmp_path):
    """MultiDeviceBDPT::split_seeds with the one GPU of this box: the whole frame rendered for the call's seeds, turned into
    sums (sthip_radiance_to_sums), reduced with the real ncclReduce(sum) and turned back: the single-device frame up to the
    rounding of (mean * n) / n; light tracing's splats — refused on a tile shard of more than one rank — go through it."""
    from stratum_amd.bdpt import BDPT

    sc, cam = scenes.cornell_box()
    W, H, seeds = 192, 96, 3
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    desc, outp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    dump_description(desc, sc, fr)
    for extra, args in (([], {}), (["--bdptFlag=connecttoviews"], {"bdptFlag": ["connecttoviews"]})):
        out = subprocess.run([_multi_host(built), "render", desc, outp, str(seeds), "0", "--splitSeeds=1"] + extra, capture_output=True, text=True)
        assert out.returncode == 0 and "RENDER OK world 1" in out.stdout, out.stdout + out.stderr
        raw = np.fromfile(outp, dtype=np.uint8)
        rad = raw[: W * H * 16].view(np.float32).reshape(H, W, 4)
        rays = raw[W * H * 16 : W * H * 16 + 16].view(np.uint64)
        r = BDPT(device=0, args=args)
        try:
            r.update(sc)
            ref = r.render(fr, 0, seeds)
        finally:
            r.close()
        assert np.array_equal(rad[..., 3], ref["radiance"][..., 3])
        np.testing.assert_allclose(rad[..., :3], ref["radiance"][..., :3], rtol=3e-7, atol=0)
        assert np.array_equal(rays, ref["ray_count"])


@pytest.mark.gpu
@pytest.mark.parametrize("make", [scenes.cornell_box, shared_mesh_scene])
def test_cpp_multi_device_driver_on_one_gpu(built, tmp_path, make):
    """MultiDeviceBDPT with the one GPU of this box: its whole path runs (one thread per rank, packed tiles, the RCCL
    send / recv group, sthip_assemble_tiles on rank 0) and the frame equals the single-device renderer's bit for bit."""
    from stratum_amd.bdpt import BDPT

    sc, cam = make()
    W, H, seeds = 192, 96, 3
    fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    desc, outp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    dump_description(desc, sc, fr)
    out = subprocess.run([_multi_host(built), "render", desc, outp, str(seeds), "0"], capture_output=True, text=True)
    assert out.returncode == 0 and "RENDER OK world 1" in out.stdout, out.stdout + out.stderr
    raw = np.fromfile(outp, dtype=np.uint8)
    rad = raw[: W * H * 16].view(np.float32).reshape(H, W, 4)
    rays = raw[W * H * 16 : W * H * 16 + 16].view(np.uint64)
    r = BDPT(device=0)
    try:
        r.update(sc)
        ref = r.render(fr, 0, seeds)
    finally:
        r.close()
    assert np.array_equal(rad.view(np.uint32), ref["radiance"].view(np.uint32))
    assert np.array_equal(rays, ref["ray_count"])


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_bvh_builder_under_sanitizers(tmp_path, sanitizer):
    """stratum_amd/csrc/bvh_build.cpp on the CPU under ASan + UBSan and under TSan (the bottom levels are built by a pool of
    threads; GPU sanitizers are not available on the pool): validation, binned SAH, embedded leaves, top level,
    transforms-only rebuild, treetop, node packing — on a merged mesh (atrium), shared instanced meshes (forest) and
    spheres / tiny meshes; tests/cpp/bvh_host_check.cpp checks that every reference stays in range and every triangle
    is referenced."""
    from stratum_amd import scenes

    exe = str(tmp_path / "bvh_host_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizer, "-o", exe, os.path.join(ROOT, "tests", "cpp", "bvh_host_check.cpp"),
                           os.path.join(ROOT, "stratum_amd", "csrc", "bvh_build.cpp"), "-lpthread"])
    for name, (sc, _) in (("atrium", scenes.atrium(target_tris=60000)), ("forest", scenes.forest(n_instances=40, tree_tris=900, tree_kinds=3)), ("spheres", scenes.spheres_room())):
        d = tmp_path / name
        d.mkdir()
        for arr, fn in ((sc.vertices, "vertices"), (sc.indices, "indices"), (sc.instances, "instances"), (sc.transforms, "xf"), (sc.inverse_transforms, "inv_xf"), (sc.materials, "materials")):
            np.ascontiguousarray(arr).tofile(str(d / (fn + ".bin")))
        env = dict(os.environ, STHIP_BUILD_THREADS="6", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1", TSAN_OPTIONS="halt_on_error=1")
        out = subprocess.run([exe, str(d)], capture_output=True, text=True, env=env)
        assert out.returncode == 0 and "BVH HOST OK" in out.stdout, name + ": " + out.stdout + out.stderr[-3000:]
