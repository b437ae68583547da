#!/usr/bin/env python3
"""bench.py — Mray/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): the procedural ~1M-triangle atrium, 1920x1080, one sample per
pixel per GPU per step, default reference flags (NEE + MIS + BSDF sampling + deferred shadow rays,
4/8 path vertices, 2 diffuse vertices), every output of sample_visibility written (gRadiance and the
G-buffer AOVs: albedo, VisibilityInfo, DepthInfo, previous-frame uv — bdpt.hlsl:222-296). A step = one
pass of the hot path (generate, trace, shade, shadow, resolve) over that batch; scene, BVH and all
buffers are resident in HBM before the timed region. With N > 1 GPUs the frame is cut into 64x32
pixel tiles dealt round-robin to the ranks; a step renders N seeds of the frame (each rank: its tiles
x N seeds = one frame's worth of paths, so per-GPU work is fixed: weak scaling) and ends with the RCCL
gather of the ranks' packed tiles to rank 0.

Ranks: under torchrun (WORLD_SIZE set) this process is one rank. Started as plain `python bench.py
--gpus N` with N > 1 it spawns its own N ranks (one child process per GPU, before anything in the
parent touches the GPU) and exits with the first non-zero child code.

value = rays of all ranks / max-over-ranks wall time of K steps (median over --reps repetitions of the
K-step timed region, each bracketed by barrier + synchronize); a ray is one trace_ray invocation,
shadow rays included (gRayCount[0], src/Shaders/common/intersection.hlsli:66).

The JSON line also carries
  roofline      for the dominant kernel (k_trace): algorithmic bytes per launch (48 B ray+hit, + node
                bytes x nodes visited, + 48 B x triangles tested) over the kernel's mean
                launch duration measured here with HIP events on the launch stream, against the HBM
                peak AND against ceilings measured in this run on this box
                (sthip_measure_ceiling: stream triad; the traversal's own 64-byte node fetch at random
                nodes without dependence, served from the whole BVH / from L2 / from L1), each with
                its own fraction. `frac` is SURVEY 8d's: algorithmic GB/s / 8000; `valu_issue` is the share of
                the chip's vector-issue cycles the kernel's instructions take, `occupancy` what the committed
                profile says binds it (the dependent round trip per step at 4 waves / SIMD);
  cpu_baseline  the CPU oracle (a port of the reference shaders; the reference has no CPU path), built
                -O3 -march=native on this box, timed on the host cores on a bounded sample of the same
                workload (median of 5 runs) and on one thread, plus the rel-L2 between the GPU and the
                oracle on that sample and the oracle's own node / triangle counts per ray.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
SIMD_COUNT = 1024  # 256 CUs x 4 SIMD-32
CLOCK_HZ = 2.4e9
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 bandwidth


def host_threads():
    """CPU threads this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def committed_profile():
    """Counter-derived figures of the newest committed profile (profiles/rNN/counters.json or traffic.json): PMC
    counters cannot be read from inside this process, so these are NOT measurements of this run and are labelled so."""
    import glob

    out = {}
    for name in ("traffic.json", "counters.json", "occupancy.json"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", name)))
        if files:
            try:
                d = json.load(open(files[-1]))
                d["source"] = os.path.relpath(files[-1], ROOT)
                out[name[:-5]] = d
            except (OSError, ValueError):
                pass
    return out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region; value = their median")
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the measured ceilings (profiling runs)")
    ap.add_argument("--no-last-ray-filter", action="store_true", help="skip the extra timed pass with answer_last_rays = 1 (profiling runs: one configuration per trace)")
    ap.add_argument("--radiance-only", action="store_true", help="do not write the G-buffer AOVs (not the headline configuration)")
    ap.add_argument("--bdpt-flag", action="append", default=[], help="as the reference's --bdptFlag (e.g. connecttolightpaths, ~nee); not the headline configuration")
    ap.add_argument("--max-diffuse-vertices", type=int, default=None)
    ap.add_argument("--light-path-count", type=int, default=None, help="as the reference's --lightPathCount: light paths behind the light vertex cache (lightvertexcache); 64 upstream")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="sthip_set_option before the scene upload (e.g. wide_bvh=3: the 8-wide walk); recorded in config.options; not the headline configuration")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): a step renders N seeds of the frame on N GPUs (per-GPU work fixed); strong: a step renders --strong-seeds seeds whatever N (total work fixed)")
    ap.add_argument("--strong-seeds", type=int, default=8, help="seeds per step in --scaling strong (BASELINE.md row 3b's 8 by default)")
    ap.add_argument("--shard", choices=("tiles", "seeds"), default="tiles",
                    help="N > 1: tiles (default): every rank renders its 64x32 tiles of every seed, one gather of the packed tiles; seeds: every rank renders the WHOLE frame for its "
                    "part of the step's seeds and one sum-reduce of the accumulation buffer adds them (SURVEY 8e 'replicas + sum-reduce': what the whole-frame estimators need)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="after the timed repetitions: one pass of back-to-back steps (the same seeds cycling) at least this long, reported as `sustained` beside `value` (0: skip)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the outputs of one more step (seed block 0) as <path>.npz after the measurements: what tests compare between N = 1 and N > 1")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the config-5 forest sub-record (other_workloads.forest; N = 1 only)")
    args = ap.parse_args()
    os.environ.setdefault("STHIP_STRICT_FLAGS", "1")
    from stratum_amd.bdpt import known_flag

    unknown = [f for f in args.bdpt_flag if not known_flag(f)]
    if unknown:  # (the reference ignores an unknown name silently, BDPT.cpp:94-127: a measurement must not)
        ap.error("unknown --bdpt-flag %s" % unknown)
    return args


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU. Nothing in this (parent)
    process has touched the GPU — torch is not even imported — and no process is replaced: children are ordinary
    subprocesses. The first child that fails ends the run with its exit code (the others are terminated by PID)."""
    # Under a profiler the parent is NOT untouched: rocprofv3's preloaded library initialises the GPU before main() runs, and
    # the children would inherit the preload. Ranks must then be started by a launcher before anything touches the GPU.
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        raise SystemExit("bench: --gpus %d under a profiler: profile one rank (--gpus 1), or start the ranks with torchrun before the profiler" % n)
    # (bind-and-close: another job may take the port before rank 0 listens on it; the ranks then fail loudly at rendezvous
    # and the run ends with their exit code — the driver's own launcher passes a port of its choosing instead)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code = 0
    live = list(procs)
    while live and code == 0:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is not None:
                live.remove(p)
                if rc != 0:
                    code = rc
    for p in live:  # a rank failed: the others would wait in a collective for ever
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    if code != 0:
        sys.stderr.write("bench: a rank exited with code %d\n" % code)
    return code if code >= 0 else 1


def forest_record(device):
    """Config 5 of BASELINE.json on one GPU: timed like tools/baseline_table.py's row 5 (median of 3 calls of 16 seeds after a
    warm-up call, device outputs, every ray traced), the G-buffer written as in the headline."""
    import torch

    from stratum_amd import camera, scenes
    from stratum_amd.bdpt import BDPT

    W, H, seeds = 3840, 2160, 16
    t0 = time.perf_counter()
    sc, cam = scenes.forest()
    t_scene = time.perf_counter() - t0
    r = BDPT(device=device, args={"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]})
    try:
        t0 = time.perf_counter()
        r.update(sc)
        t_upload = time.perf_counter() - t0
        r.set_option("answer_last_rays", 0)
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
        out = {
            "radiance": torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"),
            "albedo": torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"),
            "visibility": torch.zeros((H, W, 2), dtype=torch.int32, device="cuda"),
            "depth": torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"),
            "prev_uv": torch.zeros((H, W, 2), dtype=torch.float32, device="cuda"),
            "ray_count": torch.zeros(2, dtype=torch.int64, device="cuda"),
        }
        ptrs = {k: v.data_ptr() for k, v in out.items()}
        r.render(fr, 0, 2, device_outputs=ptrs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            r.render(fr, 0, seeds, device_outputs=ptrs)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        rays = int(out["ray_count"][0].item())
        st = r.stats()
        dt = float(np.median(ts))
        in_flight = int(min(seeds, max(1, (1 << 26) // (W * H))))  # (max_paths_in_flight as sthip_create sizes it on a 288 GB device; include/sthip.h)
        return {
            "workload": "BASELINE.json configs[4] on ONE GPU: procedural forest, %d triangles in %d instances, %dx%d, %d samples/pixel in one call, maxDiffuseVertices 8, maxPathVertices 10, minPathVertices 4, ~coherentrr, radiance + AOVs written, every ray traced"
            % (sc.triangle_count, len(sc.instances), W, H, seeds),
            "value": round(rays / dt / 1e6, 2),
            "unit": "Mray/s",
            "ms_per_call": round(dt * 1e3, 2),
            "calls_ms": [round(x * 1e3, 2) for x in ts],
            "rays_per_call": rays,
            "seeds_per_call": seeds,
            "seeds_in_flight": in_flight,
            "bvh_node_bytes": int(st["bvh_node_bytes"]),
            "bvh_build_ms": round(float(st.get("bvh_build_ms", 0.0)), 1),
            "scene_generation_s": round(t_scene, 2),
            "upload_s": round(t_upload, 2),
        }
    finally:
        r.close()


def config_record(device, sc, cam, W, H, seeds, what, shard_of=None):
    """One more BASELINE.json configuration on this GPU, timed like tools/baseline_table.py's rows (median of 5 calls after a
    warm-up call, radiance + ray counts as device outputs, every ray traced); `shard_of`: the work of rank 0 of that many
    (its 64x32 tiles, packed-tile output)."""
    import torch

    from stratum_amd import camera
    from stratum_amd.bdpt import BDPT

    r = BDPT(device=device)
    try:
        if shard_of:
            r.set_shard(0, shard_of, 64, 32)
        r.update(sc)
        r.set_option("answer_last_rays", 0)
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
        packed = shard_of is not None
        rad = torch.zeros((r.shard_slot_count(fr), 4) if packed else (H, W, 4), device="cuda")
        rc = torch.zeros(2, dtype=torch.int64, device="cuda")
        out = {"radiance": rad.data_ptr(), "ray_count": rc.data_ptr()}
        r.render(fr, 0, min(seeds, 2), device_outputs=out, packed_tiles=packed)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            r.render(fr, 0, seeds, device_outputs=out, packed_tiles=packed)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        rays = int(rc[0].item())
        dt = float(np.median(ts))
        return {"workload": what, "value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "ms_per_call": round(dt * 1e3, 2), "rays_per_call": rays, "seeds_per_call": seeds}
    finally:
        r.close()


def host_outputs_record(device, sc, cam, W, H):
    """The headline step through HOST output pointers (the boundary's other form, include/sthip.h `device_ptrs` = 0): the frame
    and its AOVs come back over PCIe inside sthip_render, which synchronises before it returns. Never `value`: the rate a
    caller without device buffers sees, one sample per pixel per call, median of 7 calls after a warm-up."""
    from stratum_amd import camera
    from stratum_amd.bdpt import BDPT

    r = BDPT(device=device)
    try:
        r.update(sc)
        r.set_option("answer_last_rays", 0)
        fr = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
        bufs = r.render(fr, 0, 1)  # the caller's buffers, allocated and touched once (pageable memory, as a plugin host's would be)
        out = {}
        for aovs in (True, False):
            ts = []
            for i in range(8):
                t = time.perf_counter()
                r.render(fr, i, 1, aovs=aovs, host_outputs=bufs)
                ts.append(time.perf_counter() - t)
            dt = float(np.median(ts[1:]))
            rays = int(bufs["ray_count"][0])
            nbytes = sum(int(v.nbytes) for k, v in bufs.items() if aovs or k in ("radiance", "ray_count"))
            out["radiance_and_aovs" if aovs else "radiance_only"] = {"value": round(rays / dt / 1e6, 2), "unit": "Mray/s", "ms_per_call": round(dt * 1e3, 3), "bytes_read_back": nbytes}
        out["workload"] = "the headline step (atrium %dx%d, one sample per pixel per call, default flags, every ray traced) with HOST output pointers: PCIe read-back and the synchronisation inside the call" % (W, H)
        return out
    finally:
        r.close()


def scene_build_record(sc, device):
    """Acceleration-structure build and update times of the bench scene (outside every timed region): both builders, a
    transforms-only update, and the rebuild the library runs itself when an instance of the merged world-space mesh has moved
    (Scene.cpp:345,435-459,614-629 rebuild whenever dirty)."""
    import copy

    from stratum_amd.bdpt import BDPT

    rec = {"unit": "ms", "triangles": int(sc.triangle_count), "instances": int(len(sc.instances))}
    for name, builder in (("host_sah", 0), ("device_ploc", 1)):
        r = BDPT(device=device)
        try:
            r.set_option("bvh_builder", builder)
            t = time.perf_counter()
            r.update(sc)
            up = (time.perf_counter() - t) * 1e3
            st = r.stats()
            rec[name] = {"sthip_scene_upload_ms": round(up, 1), "bvh_build_ms": round(float(st["bvh_build_ms"]), 1), "bvh_build_gpu_ms": round(float(st["bvh_build_gpu_ms"]), 2), "bvh_nodes": int(st["bvh_nodes"])}
            moved = copy.deepcopy(sc)
            ident = [i for i in range(len(moved.instances)) if np.array_equal(moved.transforms["m"][i], np.eye(4, dtype=np.float32)[:3])]
            others = [i for i in range(len(moved.instances)) if i not in ident]
            if others:  # an instance with a transform of its own moves: the top level alone is rebuilt
                m = moved.transforms["m"][others[0]].copy()
                m[:, 3] += np.float32(0.01)
                moved.set_instance_transform(int(others[0]), m)
                t = time.perf_counter()
                r.update_transforms(moved)
                rec[name]["transforms_only_update_ms"] = round((time.perf_counter() - t) * 1e3, 2)
            if ident:  # an instance of the merged mesh moves: the library rebuilds from the scene it kept (no STHIP_ERR_UNSUPPORTED)
                m = moved.transforms["m"][ident[0]].copy()
                m[:, 3] += np.float32(0.01)
                moved.set_instance_transform(int(ident[0]), m)
                t = time.perf_counter()
                r.update_transforms(moved)
                rec[name]["moved_merged_instance_update_ms"] = round((time.perf_counter() - t) * 1e3, 1)
                rec[name]["full_rebuilds"] = int(r.stats().get("full_rebuilds", 0))
        finally:
            r.close()
    return rec


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    # The CPU-baseline library is compiled for THIS box's cores (-march=native) and before the GPU is initialised:
    # make / g++ are children of a process that has not touched the GPU yet.
    want_cpu = world == 1 and not args.no_cpu_baseline
    if want_cpu:
        from oracle import oracle_py

        oracle_py.build()
        oracle_py.build_native()

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # rehearsal on a one-GPU box (several ranks sharing device 0 over gloo): STHIP_BENCH_ONE_DEVICE=1 STHIP_BENCH_BACKEND=gloo
    one_device = os.environ.get("STHIP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible" % (world, torch.cuda.device_count()))
    backend = os.environ.get("STHIP_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from stratum_amd import camera, scenes, shard
    from stratum_amd.bdpt import BDPT

    W, H = args.width, args.height
    sc, cam = scenes.SCENES[args.scene]()
    frame = camera.Frame(W, H, cam["fovy"], cam["eye"], cam["target"])
    bargs = {"bdptFlag": args.bdpt_flag}
    if args.max_diffuse_vertices is not None:
        bargs["maxDiffuseVertices"] = args.max_diffuse_vertices
    if args.light_path_count is not None:
        bargs["lightPathCount"] = args.light_path_count
    r = BDPT(device=local_rank, args=bargs)
    options = {}
    for kv in args.option:
        k, v = kv.split("=", 1)
        options[k] = int(v)
        r.set_option(k, int(v))
    if any("reuse" in f.lower() and f[0] not in "~!" for f in args.bdpt_flag) and "reuse_grids_persist" not in options:
        # a step is a frame: with a reservoir-reuse estimator it looks into the grids the step before left, as upstream's frames do
        options["reuse_grids_persist"] = 1
        r.set_option("reuse_grids_persist", 1)
    r.update(sc)
    # The headline traces EVERY ray. The library's default answers the last ray of a path from the emitters' bounds when it cannot
    # reach one (sthip.h "answer_last_rays": same frames, same gRayCount, 1/5 of the rays of this workload never walk the tree);
    # that configuration is timed separately below and reported as `last_ray_filter`, never as `value`.
    r.set_option("answer_last_rays", 0)
    seed_split = world > 1 and args.shard == "seeds"
    if seed_split:
        r.set_shard(0, 1, 64, 32)  # the whole frame on every rank; the ranks differ in the seeds they render
    else:
        r.set_shard(rank, world, 64, 32)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    radiance = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    # the G-buffer the reference's sample_visibility always writes (bdpt.hlsl:222-296); every rank writes its own pixels
    aov = {}
    if not args.radiance_only:
        aov = {
            "albedo": torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"),
            "visibility": torch.zeros((H, W, 2), dtype=torch.int32, device="cuda"),  # VisibilityInfo, 8 B
            "depth": torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"),  # DepthInfo, 16 B
            "prev_uv": torch.zeros((H, W, 2), dtype=torch.float32, device="cuda"),
        }
    aov_ptrs = {k: v.data_ptr() for k, v in aov.items()}
    dev_out = dict(aov_ptrs, radiance=radiance.data_ptr())
    # weak scaling (the default, what the driver's 1 -> 8 curve is): every rank renders one frame's worth of paths per step.
    # strong: the same --strong-seeds seeds per step at every N, each rank its tiles of them.
    seeds_per_step = world if args.scaling == "weak" else max(1, args.strong_seeds)

    if world == 1:

        def step(i):
            r.render(frame, seed_begin=i * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)

        def drain():
            pass

        exchange = "none"
        exchange_bytes_per_step = 0
    elif seed_split:
        # Replicas: rank r renders the whole frame for its share of the step's seeds into its own accumulation buffer, turns
        # (mean, count) into (sum, count) and ONE sum-reduce adds the buffers on rank 0 (RCCL reduce over xGMI, behind the next
        # step's render), which divides by the count. The G-buffer does not depend on the seed: rank 0's is the frame's.
        first, count = shard.seed_range(rank, world, seeds_per_step)
        sums = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        exchange_bytes_per_step = world * W * H * 16
        host_staged = backend != "nccl"
        pending = [None, None]

        def finish(k):
            if pending[k] is None:
                return
            work, cpu = pending[k]
            if work is not None:
                work.wait()
            if rank == 0:
                radiance.copy_(cpu if cpu is not None else sums[k])
                r.radiance_to_sums(radiance.data_ptr(), W * H, back=True)
            pending[k] = None

        def step(i):
            k = i & 1
            finish(k)
            if count:
                r.render(frame, seed_begin=i * seeds_per_step + first, seed_count=count, device_outputs=dict(aov_ptrs, radiance=sums[k].data_ptr()))
                r.radiance_to_sums(sums[k].data_ptr(), W * H)
            else:
                sums[k].zero_()
            if host_staged:
                cpu = sums[k].cpu()
                shard.reduce_seed_sums(cpu, dist, dst=0)
                pending[k] = (None, cpu)
            else:
                pending[k] = (shard.reduce_seed_sums(sums[k], dist, dst=0, async_op=True), None)
            finish(k ^ 1)

        def drain():
            finish(0)
            finish(1)

        exchange = "sum-reduce of the whole-frame accumulation buffers to rank 0 (%s), pipelined behind the next step" % ("RCCL" if backend == "nccl" else backend)
    else:
        # Sharded frame: every rank renders only its tiles (packed, 1 / world of the frame) and the one exchange of the
        # path is a gather of those to rank 0, which scatters them into the image (sthip_assemble_tiles). The gather of
        # step i runs on RCCL's stream while step i + 1 renders into the other buffer. A failure of the exchange is a
        # failure of the run: nothing here falls back to another form.
        stride = shard.slot_count(W, H, 0, world)  # rank 0 owns the most tiles: equal-size messages
        # One message per rank and step: its tiles of EVERY output of the frame, in slot order — the radiance (16 B per slot,
        # written packed by the render) and, as the reference's pass always yields them (bdpt.hlsl:222-296), the G-buffer:
        # albedo 16 B, DepthInfo 16 B, VisibilityInfo 8 B, previous-frame uv 8 B (sthip_pack_tiles of the rank's own images):
        # 64 B per slot, so an N-GPU step delivers what the one-GPU step delivers.
        segments = {"radiance": (0, 16)}  # output -> (first float of its segment in a rank's message, bytes per slot)
        entry_floats = 4
        for name, nbytes in (("albedo", 16), ("depth", 16), ("visibility", 8), ("prev_uv", 8)):
            if name in aov:
                segments[name] = (entry_floats * stride, nbytes)
                entry_floats += nbytes // 4
        packed = [torch.zeros(stride * entry_floats, dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros((world, stride * entry_floats), dtype=torch.float32, device="cuda") if rank == 0 else None for _ in range(2)]
        # what the rank renders its own G-buffer tiles into (W x H images, zero elsewhere); `aov` holds the assembled frame's on rank 0
        aov_local = {k: torch.zeros_like(v) for k, v in aov.items()}
        aov_local_ptrs = {k: v.data_ptr() for k, v in aov_local.items()}
        exchange_bytes_per_step = world * stride * entry_floats * 4
        host_staged = backend != "nccl"  # gloo rehearsal: collectives on CPU tensors
        pending = [None, None]

        def finish(k):
            if pending[k] is None:
                return
            work, g_cpu = pending[k]
            if work is not None:
                work.wait()
            if rank == 0:
                if g_cpu is not None:
                    gathered[k].copy_(g_cpu)
                base = gathered[k].data_ptr()
                r.assemble_tiles(frame, base, stride * entry_floats // 4, radiance.data_ptr())
                for name, t in aov.items():
                    first, nbytes = segments[name]
                    r.assemble_tiles_bytes(frame, base + 4 * first, stride * entry_floats * 4 // nbytes, t.data_ptr(), nbytes)
            pending[k] = None

        def step(i):
            k = i & 1
            finish(k)  # buffer k is free again once its gather has been consumed
            r.render(frame, seed_begin=i * seeds_per_step, seed_count=seeds_per_step, device_outputs=dict(aov_local_ptrs, radiance=packed[k].data_ptr()), packed_tiles=True)
            for name, t in aov_local.items():
                first, nbytes = segments[name]
                r.pack_tiles(frame, t.data_ptr(), nbytes, packed[k].data_ptr() + 4 * first)
            if host_staged:
                src = packed[k].cpu()
                g_cpu = torch.zeros((world, stride * entry_floats)) if rank == 0 else None
                shard.gather_tiles(src, g_cpu, dist, dst=0)
                pending[k] = (None, g_cpu)
            else:
                pending[k] = (shard.gather_tiles(packed[k], gathered[k], dist, dst=0, async_op=True), None)
            finish(k ^ 1)  # the previous step's tiles have arrived meanwhile: assemble them behind this render

        def drain():
            finish(0)
            finish(1)

        exchange = "gather of packed tiles (%s; %d B per slot) to rank 0 (%s), pipelined behind the next step" % (" + ".join(segments), entry_floats * 4, "RCCL" if backend == "nccl" else backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    barrier()
    # ---- the timed region: EXACTLY K steps between barrier + synchronize on both sides, `reps` times ----
    rep_dt = []
    for _ in range(max(1, args.reps)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        drain()
        barrier()
        rep_dt.append(time.perf_counter() - t0)
    # rays of the timed region: the same steps again with the counters read back (untimed; every repetition traces
    # the same seeds, hence the same rays)
    rays_local = answered_local = 0
    for i in range(args.steps):
        sb, sn = (args.warmup + i) * seeds_per_step, seeds_per_step
        if seed_split:  # this rank's share of the step's seeds, the whole frame
            f0, sn = shard.seed_range(rank, world, seeds_per_step)
            sb += f0
        if sn == 0:
            continue
        r.render(frame, seed_begin=sb, seed_count=sn, device_outputs=dev_out)
        st = r.stats()
        rays_local += st["rays_total"]
        answered_local += st["rays_answered"]  # of them: last rays of paths answered from the emitters' bounds, no traversal (sthip.h)
    # the exchange alone (untimed region): the gather of one step's packed tiles and their assembly, between syncs, so that
    # a scaling record can be split into render time and exchange time
    exchange_ms = 0.0
    if dist is not None:
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            k = i & 1
            if seed_split:
                if host_staged:
                    cpu = sums[k].cpu()
                    shard.reduce_seed_sums(cpu, dist, dst=0)
                    pending[k] = (None, cpu)
                else:
                    pending[k] = (shard.reduce_seed_sums(sums[k], dist, dst=0, async_op=True), None)
            elif host_staged:
                g_cpu = torch.zeros((world, stride * entry_floats)) if rank == 0 else None
                shard.gather_tiles(packed[k].cpu(), g_cpu, dist, dst=0)
                pending[k] = (None, g_cpu)
            else:
                pending[k] = (shard.gather_tiles(packed[k], gathered[k], dist, dst=0, async_op=True), None)
            finish(k)
        barrier()
        exchange_ms = (time.perf_counter() - t0) / args.steps * 1e3
    t = torch.tensor(rep_dt + [float(answered_local), float(rays_local)], dtype=torch.float64, device="cuda")
    devices = [{"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(local_rank), "median_ms_per_step": round(float(np.median(rep_dt)) / args.steps * 1e3, 3), "rays_per_step": int(rays_local / args.steps)}]
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)  # per repetition: the slowest rank
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rep_all, rays_all, answered_all = [float(x) for x in tmax[:-2]], float(t[-1]), float(t[-2])
        gathered_devices = [None] * world
        dist.all_gather_object(gathered_devices, devices[0])
        devices = gathered_devices
        world_seen, backend_seen = dist.get_world_size(), dist.get_backend()
    else:
        rep_all, rays_all, answered_all = [float(x) for x in t[:-2]], float(t[-1]), float(t[-2])
        world_seen, backend_seen = 1, "none"
    dt_all = float(np.median(rep_all))
    # ---- steady state: ONE pass of back-to-back steps at least --sustained-seconds long (the same K seeds cycling, so the rays
    # of a cycle are the counted ones), between the same barriers. Reported beside `value`, never instead of it: `value` is the
    # median of short bursts; this shows what survives the clocks and power the chip settles at.
    sustained = None
    if args.sustained_seconds > 0:
        cycles = 0
        barrier()
        t0 = time.perf_counter()
        sdt = 0.0
        while sdt < args.sustained_seconds:  # (normally once: the estimate comes from the timed region; every rank sees the same all-reduced times)
            batch = max(1, int(np.ceil(1.05 * (args.sustained_seconds - sdt) / max(dt_all, 1e-6))))
            for _ in range(batch):
                for i in range(args.steps):
                    step(args.warmup + i)
            drain()
            barrier()
            cycles += batch
            sdt = time.perf_counter() - t0
            if dist is not None:
                ts = torch.tensor([sdt], dtype=torch.float64, device="cuda")
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
                sdt = float(ts[0])
        sustained = {
            "value": round(rays_all * cycles / sdt / 1e6, 2),
            "unit": "Mray/s",
            "ms_per_step": round(sdt / (cycles * args.steps) * 1e3, 3),
            "seconds": round(sdt, 3),
            "steps": cycles * args.steps,
            "what": "one uninterrupted pass of the timed region's steps (the same %d seeds cycling), barrier + synchronize on both sides, max over ranks" % args.steps,
        }

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel (k_trace: BVH traversal of closest-hit and shadow rays), this rank ----
        r.set_option("time_kernels", 1)
        ms_trace, ms_primary, ms_shade, ms_total, launches, launches_primary = 0.0, 0.0, 0.0, 0.0, 0, 0
        for i in range(args.steps):
            r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
            s = r.stats()
            ms_trace += s["ms_trace"]  # k_trace alone: the first bounce runs in k_trace_primary (wave packets), timed apart
            ms_primary += s["ms_trace_primary"]
            ms_shade += s["ms_shade"]
            ms_total += s["ms_total"]
            launches += s["launches_trace"]
            launches_primary += s["launches_primary"]
        r.set_option("time_kernels", 0)
        r.set_option("count_traversal", 1)
        nodes = tris = rays_closest = rays_shadow = nodes_sh = tris_sh = nodes_pr = tris_pr = rays_pr = 0
        lanes = {"inner": [0, 0], "tri": [0, 0]}
        for i in range(args.steps):
            r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
            s = r.stats()
            # the rays, node visits and triangle tests of k_trace: everything but the first bounce's packets
            nodes += s["nodes_visited"] - s["nodes_visited_primary"]
            tris += s["tris_tested"] - s["tris_tested_primary"]
            nodes_pr += s["nodes_visited_primary"]
            tris_pr += s["tris_tested_primary"]
            rays_pr += s["rays_primary_packets"]
            rays_closest -= s["rays_primary_packets"]
            nodes_sh += s["nodes_visited_shadow"]
            tris_sh += s["tris_tested_shadow"]
            rays_closest += s["rays_path"] - s["rays_answered"]  # (answered last rays never reach k_trace)
            rays_shadow += s["rays_shadow"]
            lanes["inner"][0] += s["nodes_visited"] - s["nodes_visited_primary"] + s["nodes_visited_shadow"]
            lanes["inner"][1] += s["inner_slots"][0] + s["inner_slots"][1]
            lanes["tri"][0] += s["tris_tested"] - s["tris_tested_primary"] + s["tris_tested_shadow"]
            lanes["tri"][1] += s["tri_slots"][0] + s["tri_slots"][1]
        r.set_option("count_traversal", 0)
        node_bytes, tri_bytes = s["bvh_node_bytes"], s["bvh_tri_bytes"]
        rays = rays_closest + rays_shadow
        # B_ray = 48 (ray in + hit out; a shadow record is 48 B too) + node bytes * nodes + triangle bytes * tris
        node_alg_bytes = float(node_bytes) * (nodes + nodes_sh)
        alg_bytes = 48.0 * rays + node_alg_bytes + float(tri_bytes) * (tris + tris_sh)
        achieved = alg_bytes / (ms_trace * 1e-3) / 1e9 if ms_trace > 0 else 0.0
        node_rate = node_alg_bytes / (ms_trace * 1e-3) / 1e9 if ms_trace > 0 else 0.0
        # ---- ceilings measured now, on this box (sthip_measure_ceiling) ----
        ceilings = {"hbm_spec": {"peak": HBM_PEAK_GBS, "frac": round(achieved / HBM_PEAK_GBS, 4), "what": "algorithmic bytes of k_trace / HBM3E spec peak (SURVEY 8d)"}}
        if not args.no_ceilings:
            triad = r.measure_ceiling("triad")
            ceilings["hbm_triad_measured"] = {"peak": round(triad, 1), "frac": round(achieved / triad, 4), "what": "algorithmic bytes / stream triad measured in this run"}
            for key, what in (
                ("node_gather_table", "k_trace's node bytes / rate of independent random 64-B node fetches over this BVH's node array (L2 + Infinity Cache)"),
                ("node_gather_l2", "k_trace's node bytes / the same fetch over a 2 MiB prefix (L2 hits)"),
                ("node_gather_l1", "k_trace's node bytes / the same fetch over a 16 KiB prefix (vector-memory front end)"),
            ):
                g = r.measure_ceiling(key)
                ceilings[key] = {"peak": round(g, 1), "frac": round(node_rate / g, 4) if g > 0 else None, "what": what}
        # Every ceiling is reported with its own fraction; one the kernel runs ABOVE is marked "exceeded" (it is not a ceiling
        # for this access pattern: most node fetches are served above that level of the hierarchy). The headline `frac` is
        # SURVEY 8d's figure and nothing else: algorithmic bytes per launch / launch time / the 8 TB/s HBM3E peak. It rises
        # when the kernel gets faster on the same rays and cannot be picked to look good.
        for c in ceilings.values():
            c["exceeded"] = bool(c["frac"] is not None and c["frac"] > 1.0)
        prof = committed_profile()
        # The kernel's share of vector-instruction issue according to the committed PMC profile. A wave64 VALU
        # instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md), 1024 SIMDs at 2.4 GHz:
        #   valu_issue.frac = SQ_INSTS_VALU per launch * 2 / (1024 * 2.4e9 * launch time)
        # The instruction count is the committed profile's (PMC counters need rocprofv3), the launch time is this run's.
        launch_ms = ms_trace / max(launches, 1)
        cnt = prof.get("counters") or {}
        valu = None
        if cnt.get("valu_insts_per_launch") and launch_ms > 0:
            valu = {
                "frac": round(cnt["valu_insts_per_launch"] * 2.0 / (SIMD_COUNT * CLOCK_HZ * launch_ms * 1e-3), 4),
                "valu_insts_per_launch": cnt["valu_insts_per_launch"],
                "lane_utilisation": cnt.get("lane_utilisation"),
                "cycles_per_wave_instruction": 2,
                "simds": SIMD_COUNT,
                "clock_hz": CLOCK_HZ,
                "source": cnt.get("source"),
                "what": "share of the chip's vector-issue cycles k_trace's VALU instructions occupy (instruction count: committed profile; launch time: this run); lane_utilisation = active lanes per issued VALU instruction",
            }
        roofline = {
            "bound": "hbm",  # SURVEY 8d's roofline: divergent gathers, no MFMA on this path
            # what the counters say binds it (profiles/r03): neither HBM nor vector issue (0.44) nor the L1 front end, but
            # the dependent fetch -> test -> fetch round trip of a step at the 4 waves / SIMD its registers and LDS stacks allow
            "bound_observed": "latency_at_occupancy" if prof.get("occupancy") else ("valu_issue" if valu else None),
            "occupancy": prof.get("occupancy"),
            "kernel": "k_trace",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ceilings["hbm_spec"]["frac"],
            "valu_issue": valu,
            "traffic": prof.get("traffic", {}).get("bytes_per_launch"),
            "traffic_source": prof.get("traffic", {}).get("source", None),  # a committed profile, NOT this run (PMC needs rocprofv3)
            "algorithmic_gbs": round(achieved, 2),
            "hbm_spec_frac": ceilings["hbm_spec"]["frac"],  # SURVEY 8d's figure: algorithmic bytes / 8 TB/s
            "node_fetch_gbs": round(node_rate, 2),
            "ceilings": ceilings,
            "counters": prof.get("counters"),
            "bytes_per_launch": round(alg_bytes / max(launches, 1), 1),
            "launch_ms": round(launch_ms, 4),
            "launches_per_step": round(launches / args.steps, 2),
            "nodes_per_ray": round((nodes + nodes_sh) / max(rays, 1), 2),
            "node_bytes": int(node_bytes),  # of the nodes k_trace walks: 48 = binary, 64 = the 4-wide form (wide_bvh)
            "counts_from": "the kernel's own visit counters on its own tree (count_traversal pass of the same steps); the oracle's counts on its tree are in cpu_baseline",
            "tris_per_ray": round((tris + tris_sh) / max(rays, 1), 2),
            "bytes_per_ray": round(alg_bytes / max(rays, 1), 1),
            "kernel_ms_per_step": {
                "trace": round(ms_trace / args.steps, 3),
                "trace_primary": round(ms_primary / args.steps, 3),
                "shade": round(ms_shade / args.steps, 3),
                "all": round(ms_total / args.steps, 3),
            },
            "closest_nodes_per_ray": round(nodes / max(rays_closest, 1), 2),
            "shadow_nodes_per_ray": round(nodes_sh / max(rays_shadow, 1), 2),
            "rays_per_launch": round(rays / max(launches, 1), 1),
            "lane_utilisation": {"node_loop": round(lanes["inner"][0] / max(lanes["inner"][1], 1), 3), "triangle_loop": round(lanes["tri"][0] / max(lanes["tri"][1], 1), 3)},
            # the other kernels of a step, priced the same way (HIP events of this run)
            "other_kernels": {
                "k_trace_primary": {
                    "ms_per_launch": round(ms_primary / max(launches_primary, 1), 4),
                    "rays_per_launch": round(rays_pr / max(launches_primary, 1), 1),
                    "nodes_per_packet": round(nodes_pr / max(rays_pr / 64.0, 1), 1),
                    "tris_per_packet": round(tris_pr / max(rays_pr / 64.0, 1), 1),
                    "mray_per_s": round(rays_pr / max(ms_primary, 1e-9) / 1e3, 1),
                    "note": "first bounce as 8x8-pixel wave packets: one wave-uniform (scalar) node fetch serves 64 rays; ~50 % of the vector-issue cycles at 6 waves / SIMD (profiles/r03/pmc_sq.txt) — the 64 slab / triangle tests per step, not bytes and not the scalar-load chain (EXPERIMENTS.md)",
                },
                "k_shade": {
                    "ms_per_step": round(ms_shade / args.steps, 4),
                    "algorithmic_bytes_per_vertex": 244 + 208,
                    "achieved_gbs": round((244 + 208) * float(rays_closest + rays_pr) / max(ms_shade * 1e-3, 1e-12) / 1e9, 1),
                    "peak_gbs": HBM_PEAK_GBS,
                    "note": "per path vertex 244 B of gathers (3 vertices, indices, instance, transform, material; SURVEY 8d) + 208 B of path state read and written (algorithmic); the committed PMC profile shows 1.94 GB of HBM-side traffic per step (3.2 TB/s) and ~2300 VALU instructions per vertex (correctly rounded divisions, pcg4d, software transcendentals: the arithmetic contract): 31 % of the vector-issue cycles, waves parked on dependent gathers (hit -> instance -> indices -> vertices -> material) 55 % of theirs at the 3 waves / SIMD its 168 registers allow; not HBM (DESIGN.md 4)",
                },
            },
            "note": "k_trace = closest-hit rays of bounce >= 1 and all shadow rays (the first bounce runs as wave packets in k_trace_primary: other_kernels). "
            "algorithmic_gbs counts 48 B/ray + node and triangle bytes per visit (SURVEY 8d); the BVH lives in L2 / Infinity Cache, so that figure can exceed the HBM peak "
            "HBM-side traffic (`traffic`, committed PMC profile) is about half of it: HBM does not bind this kernel; vector-instruction issue stands at valu_issue.frac, and `occupancy` (committed profile) shows what does: the dependent round trip of a step at 4 waves / SIMD. "
            "frac = algorithmic_gbs / 8000 (SURVEY 8d); ceilings lists every measured ceiling with its own fraction, `exceeded` where the kernel runs above it.",
        }

        # ---- the library's default: last rays answered from the emitters' bounds (identical frames and ray counts). Timed the same
        # way as the headline, reported beside it: rays = the reference's trace_ray calls (gRayCount), of which `answered` walk no tree.
        last_ray_filter = None
        if world == 1 and not args.no_last_ray_filter:
            r.set_option("answer_last_rays", 1)
            for i in range(2):
                step(i)
            reps_f = []
            for _ in range(3):
                barrier()
                tf0 = time.perf_counter()
                for i in range(args.steps):
                    step(args.warmup + i)
                barrier()
                reps_f.append(time.perf_counter() - tf0)
            frays = fans = 0
            for i in range(args.steps):
                r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
                st = r.stats()
                frays += st["rays_total"]
                fans += st["rays_answered"]
            r.set_option("answer_last_rays", 0)
            dtf = float(np.median(reps_f))
            last_ray_filter = {
                "option": "answer_last_rays = 1 (the library's default; `value` above is measured with 0: every ray traced)",
                "value": round(frays / dtf / 1e6, 2),
                "unit": "Mray/s (gRayCount semantics: the reference's trace_ray calls of the identical frame)",
                "ms_per_step": round(dtf / args.steps * 1e3, 3),
                "rays_per_step": int(frays / args.steps),
                "rays_answered_without_traversal_per_step": int(fans / args.steps),
                "value_traversed_rays_only": round((frays - fans) / dtf / 1e6, 2),
                "what": "a path's last ray (the path / diffuse budget ends at its far end: it can only still find an emitter) is queued only if it can reach the bounds of an emissive instance; frames and ray counts are bit-identical (tests/test_gpu_parity.py)",
            }

        # ---- the same steps through HOST output pointers (what a caller without device buffers pays): the frame comes
        # back over PCIe inside the call. Reported next to `value`, never as `value`.
        host_rate = None
        if world == 1:
            r.render(frame, seed_begin=0, seed_count=1, aovs=False)
            th = time.perf_counter()
            hrays = 0
            for i in range(args.steps):
                hrays += int(r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, aovs=False)["ray_count"][0])
            host_rate = round(hrays / (time.perf_counter() - th) / 1e6, 2)

        # ---- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N = 1 only) ----
        cpu = None
        if want_cpu:
            from oracle import oracle_py

            sw, sh, nseed = W, H, 4  # the same workload: full frame, four samples per pixel (a few seconds on the box's cores)
            sframe = camera.Frame(sw, sh, cam["fovy"], cam["eye"], cam["target"])
            r.set_shard(0, 1, 64, 32)
            r.set_option("count_traversal", 1)
            got = r.render(sframe, 0, nseed, aovs=False)
            gs = r.stats()
            r.set_option("count_traversal", 0)
            o = oracle_py.OracleScene(sc, native=True)
            threads = host_threads()
            pc = r.push_constants(sframe)
            wframe = camera.Frame(sw // 8, sh // 8, cam["fovy"], cam["eye"], cam["target"])
            o.render(wframe, r.push_constants(wframe), r.mSamplingFlags, 0, 1, threads=threads, aovs=False)  # warm-up
            runs = []
            for _ in range(5):
                t1 = time.perf_counter()
                ref = o.render(sframe, pc, r.mSamplingFlags, 0, nseed, threads=threads, aovs=False)
                runs.append(time.perf_counter() - t1)
            cdt = float(np.median(runs))
            # one thread, on 1/16 of the frame (same camera, 480x270) and the same seeds
            t1 = time.perf_counter()
            ref1 = o.render(camera.Frame(sw // 4, sh // 4, cam["fovy"], cam["eye"], cam["target"]), r.push_constants(camera.Frame(sw // 4, sh // 4, cam["fovy"], cam["eye"], cam["target"])), r.mSamplingFlags, 0, nseed, threads=1, aovs=False)
            cdt1 = time.perf_counter() - t1
            a = got["radiance"][..., :3].astype(np.float64)
            b = ref["radiance"][..., :3].astype(np.float64)
            rel = float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b**2).sum()), 1e-300))
            orays = float(ref["ray_count"][0])
            gnodes = gs["nodes_visited"] + gs["nodes_visited_shadow"]
            gtris = gs["tris_tested"] + gs["tris_tested_shadow"]
            cpu = {
                "value": round(orays / cdt / 1e6, 3),
                "unit": "Mray/s",
                "cores": threads,
                "kind": "port",
                "sample": "%s %dx%d x %d samples, default flags (%d rays; median of 5 runs: %s s); oracle built -O3 -march=native on this box"
                % (args.scene, sw, sh, nseed, int(orays), ", ".join("%.2f" % x for x in runs)),
                "single_thread_value": round(float(ref1["ray_count"][0]) / cdt1 / 1e6, 3),
                "single_thread_sample": "%dx%d x %d samples (%d rays, %.2f s)" % (sw // 4, sh // 4, nseed, int(ref1["ray_count"][0]), cdt1),
                "rel_l2_gpu_vs_oracle": rel,
                # SURVEY 8d's shared figure counted by the oracle (its own binned-SAH BVH2, exact boxes, one ray at a time)
                # on the same rays, next to what the GPU's kernels visited for them (padded boxes, packets for bounce 0)
                "oracle_nodes_per_ray": round(float(ref["stats"][2]) / orays, 2),
                "oracle_tris_per_ray": round(float(ref["stats"][3]) / orays, 2),
                "gpu_nodes_per_ray": round(gnodes / max(gs["rays_total"], 1), 2),
                "gpu_tris_per_ray": round(gtris / max(gs["rays_total"], 1), 2),
                "gpu_over_oracle_node_visits": round(gnodes / max(float(ref["stats"][2]), 1.0), 3),
            }
        # ---- BASELINE.json configs[4] on this one GPU, as a clearly labelled second record (never `value`): the 10M-triangle
        # instanced forest, 3840x2160, 16 seeds in one call, 8 diffuse / 10 path vertices, ~coherentrr, every output written
        other_workloads = None
        if world == 1 and not args.no_other_workloads and args.scene == "atrium":
            box, box_cam = scenes.cornell_box()
            other_workloads = {
                "cornell_1080p_64spp": config_record(local_rank, box, box_cam, 1920, 1080, 64, "BASELINE.json configs[1]: Cornell box, 1920x1080, 64 samples/pixel in one call, default flags, radiance only, every ray traced"),
                "atrium_256spp_share_of_one_of_8_ranks": config_record(local_rank, sc, cam, W, H, 256, "BASELINE.json configs[3], the work of ONE of its 8 ranks on this GPU: atrium 1920x1080, 256 samples/pixel, the 64x32 tiles t % 8 == 0, packed-tile output, radiance only, every ray traced", shard_of=8),
                "forest": forest_record(local_rank),
                "scene_build": scene_build_record(sc, local_rank),
                "host_outputs": host_outputs_record(local_rank, sc, cam, W, H),
            }
        flags_text = "default BDPT flags" if not (args.bdpt_flag or args.max_diffuse_vertices) else "flags %s maxDiffuseVertices %s" % (args.bdpt_flag, args.max_diffuse_vertices)
        result = {
            "metric": "Mray/s at 1920x1080x1spp (1M-tri scene)",
            "value": round(rays_all / dt_all / 1e6, 2),
            "unit": "Mray/s",
            "n_gpus": world_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "procedural %s, %d triangles, %dx%d, %d sample(s)/pixel/step, %s, %s, pixel-tile shard 64x32 over %d GPU(s)"
                % (args.scene, sc.triangle_count, W, H, seeds_per_step, flags_text, "radiance only" if args.radiance_only else "radiance + albedo/visibility/depth/prev-uv AOVs written", world)
                + (" (--shard seeds: whole frames over disjoint seed ranges, one sum-reduce)" if seed_split else ""),
                "rays_per_step": int(rays_all / args.steps),
                "rays_answered_without_traversal_per_step": int(answered_all / args.steps),  # 0: the headline traces every ray (last_ray_filter below is the other configuration)
                "parallelism": ("seed-split replicas x%d" % world if seed_split else "tile-shard x%d" % world) if world > 1 else "single GPU",
                "exchange": exchange,
                "exchange_bytes_per_step": exchange_bytes_per_step,  # what reaches rank 0 per step: every rank's tiles of every output
                "exchange_alone_ms_per_step": round(exchange_ms, 4) if world > 1 else None,  # gather + assembly of one step's tiles, not overlapped (in the timed region it runs behind the next step's render)
                "seeds_per_step": seeds_per_step,
                "options": options or None,  # sthip_set_option overrides of this run (none: the library's defaults, answer_last_rays aside)
                "world_size": world_seen,
                "backend": backend_seen,
                "devices": devices,
            },
            "repetitions": {"n": len(rep_all), "ms_per_step": [round(x / args.steps * 1e3, 3) for x in rep_all], "value_is": "median"},
            "sustained": sustained,
            "other_workloads": other_workloads,
            "last_ray_filter": last_ray_filter,
            "host_output_value": host_rate,  # Mray/s with the radiance image copied to host memory inside every call
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    if args.dump_frame:
        step(0)
        drain()
        barrier()
        if rank == 0:
            np.savez(args.dump_frame, radiance=radiance.cpu().numpy(), **{k: v.cpu().numpy() for k, v in aov.items()})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result))
    r.close()


if __name__ == "__main__":
    main()
