#!/usr/bin/env python3
"""bench.py — Mray/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): the procedural ~1M-triangle atrium, 1920x1080, one sample per
pixel per GPU per step, default reference flags (NEE + MIS + BSDF sampling + deferred shadow rays,
4/8 path vertices, 2 diffuse vertices). A step = one pass of the hot path (generate, trace, shade,
shadow, resolve) over that batch; scene, BVH and all buffers are resident in HBM before the timed
region. With N > 1 GPUs the frame is cut into 64x32 pixel tiles dealt round-robin to the ranks; a step
renders N seeds of the frame (each rank: its tiles x N seeds = one frame's worth of paths, so per-GPU
work is fixed: weak scaling) and ends with the RCCL sum-reduce of the RGBA32F framebuffer to rank 0.

value = rays of all ranks / max-over-ranks wall time; a ray is one trace_ray invocation, shadow rays
included (gRayCount[0], src/Shaders/common/intersection.hlsli:66).

The JSON line also carries
  roofline      for the dominant kernel (k_trace): algorithmic bytes per launch
                (48 B ray+hit, + node bytes x nodes visited, + 48 B x triangles tested) over the
                kernel's mean launch duration, measured here with HIP events on the launch stream;
  cpu_baseline  the CPU oracle (a port of the reference shaders; the reference has no CPU path) timed on
                the host cores on a bounded sample of the same workload, plus the rel-L2 between the GPU
                and the oracle on that sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E


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


def measured_traffic():
    """HBM-side bytes per live k_trace launch from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    their own rocprofv3 runs and corrected as MI355X_MICROARCH.md prescribes; tools/pmc.sh + tools/traffic.py).
    Counters cannot be read from inside this process, so this is the value of the newest committed profile."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json")))
    if not files:
        return None
    try:
        return round(json.load(open(files[-1]))["bytes_per_launch"], 1)
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bdpt-flag", action="append", default=[], help="as the reference's --bdptFlag (e.g. connecttolightpaths, ~nee); not the headline configuration")
    ap.add_argument("--max-diffuse-vertices", type=int, default=None)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    # rehearsal on a one-GPU box (several ranks sharing device 0 over gloo): STHIP_BENCH_ONE_DEVICE=1 STHIP_BENCH_BACKEND=gloo
    if os.environ.get("STHIP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
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
    r = BDPT(device=local_rank, args=bargs)
    r.update(sc)
    r.set_shard(rank, world, 64, 32)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    radiance = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    dev_out = {"radiance": radiance.data_ptr()}
    seeds_per_step = world  # weak scaling: every rank renders one frame's worth of paths per step

    if world == 1:

        def step(i):
            r.render(frame, seed_begin=i * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)

        def drain():
            pass

    else:
        # Sharded frame: every rank renders only its tiles (packed, 1 / world of the frame) and the one exchange of the
        # path is a gather of those to rank 0, which scatters them into the image (sthip_assemble_tiles). The gather of
        # step i runs on RCCL's stream while step i + 1 renders into the other buffer.
        stride = shard.slot_count(W, H, 0, world)  # rank 0 owns the most tiles: equal-size messages
        packed = [torch.zeros((stride, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros((world, stride, 4), dtype=torch.float32, device="cuda") if rank == 0 else None for _ in range(2)]
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
                r.assemble_tiles(frame, gathered[k].data_ptr(), stride, radiance.data_ptr())
            pending[k] = None

        def step(i):
            k = i & 1
            finish(k)  # buffer k is free again once its gather has been consumed
            r.render(frame, seed_begin=i * seeds_per_step, seed_count=seeds_per_step, device_outputs={"radiance": packed[k].data_ptr()}, packed_tiles=True)
            if host_staged:
                src = packed[k].cpu()
                g_cpu = torch.zeros((world, stride, 4)) if rank == 0 else None
                shard.gather_tiles(src, g_cpu, dist, dst=0)
                pending[k] = (None, g_cpu)
            else:
                pending[k] = (shard.gather_tiles(packed[k], gathered[k], dist, dst=0, async_op=True), None)
            finish(k ^ 1)  # the previous step's tiles have arrived meanwhile: assemble them behind this render

        def drain():
            finish(0)
            finish(1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exchange = "none" if world == 1 else "gather of packed tiles, pipelined"
    try:
        for i in range(args.warmup):
            step(i)
        drain()
        barrier()
    except Exception as e:  # harness plumbing only: the zero-padded reduce is the other documented exchange form
        if world == 1:
            raise
        sys.stderr.write("bench: packed gather failed (%s); using the sum-reduce of zero-padded frames\n" % e)
        exchange = "sum-reduce of zero-padded frames"

        def step(i):  # noqa: F811
            r.render(frame, seed_begin=i * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
            shard.reduce_framebuffer(radiance, dist, dst=0)

        def drain():  # noqa: F811
            pass

        for i in range(args.warmup):
            step(i)
        barrier()
    t0 = time.perf_counter()
    rays_local = 0
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    barrier()
    dt = time.perf_counter() - t0
    # rays of the timed region: re-run the same steps with the counters read back (untimed)
    for i in range(args.steps):
        r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
        rays_local += r.stats()["rays_total"]
    t = torch.tensor([dt, float(rays_local)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt_all, rays_all = float(tmax[0]), float(t[1])
    else:
        dt_all, rays_all = float(t[0]), float(t[1])

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel (k_trace: BVH traversal of closest-hit and shadow rays), this rank ----
        r.set_option("time_kernels", 1)
        ms_trace, ms_primary, ms_shade, ms_total, launches = 0.0, 0.0, 0.0, 0.0, 0
        for i in range(args.steps):
            r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
            s = r.stats()
            ms_trace += s["ms_trace"]  # k_trace alone: the first bounce runs in k_trace_primary (wave packets), timed apart
            ms_primary += s["ms_trace_primary"]
            ms_shade += s["ms_shade"]
            ms_total += s["ms_total"]
            launches += s["launches_trace"]
        r.set_option("time_kernels", 0)
        r.set_option("count_traversal", 1)
        nodes = tris = rays_closest = rays_shadow = nodes_sh = tris_sh = 0
        for i in range(args.steps):
            r.render(frame, seed_begin=(args.warmup + i) * seeds_per_step, seed_count=seeds_per_step, device_outputs=dev_out)
            s = r.stats()
            # the rays, node visits and triangle tests of k_trace: everything but the first bounce's packets
            nodes += s["nodes_visited"] - s["nodes_visited_primary"]
            tris += s["tris_tested"] - s["tris_tested_primary"]
            rays_closest -= s["rays_primary_packets"]
            nodes_sh += s["nodes_visited_shadow"]
            tris_sh += s["tris_tested_shadow"]
            rays_closest += s["rays_path"]
            rays_shadow += s["rays_shadow"]
        r.set_option("count_traversal", 0)
        node_bytes, tri_bytes = s["bvh_node_bytes"], s["bvh_tri_bytes"]
        rays = rays_closest + rays_shadow
        # B_ray = 48 (ray in + hit out; a shadow record is 48 B too) + node bytes * nodes + triangle bytes * tris
        alg_bytes = 48.0 * rays + float(node_bytes) * (nodes + nodes_sh) + float(tri_bytes) * (tris + tris_sh)
        achieved = alg_bytes / (ms_trace * 1e-3) / 1e9 if ms_trace > 0 else 0.0
        roofline = {
            "bound": "hbm",
            "kernel": "k_trace",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(),
            "bytes_per_launch": round(alg_bytes / max(launches, 1), 1),
            "launch_ms": round(ms_trace / max(launches, 1), 4),
            "launches_per_step": round(launches / args.steps, 2),
            "nodes_per_ray": round((nodes + nodes_sh) / max(rays, 1), 2),
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
            "note": "k_trace = closest-hit rays of bounce >= 1 and all shadow rays (the first bounce runs as wave packets in "
            "k_trace_primary and is not part of these figures); achieved counts ALGORITHMIC bytes (48 B/ray + node and triangle bytes per visit, SURVEY 8d); most node "
            "fetches hit L2 / Infinity Cache (compare traffic), so frac > 1 means the kernel runs above what HBM alone could "
            "feed: it is bound by dependent-load latency and lane divergence, see profiles/README.md",
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
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_py

            sw, sh, nseed = W, H, 8  # the same workload: full frame, eight samples per pixel (about 10 s of CPU work on 16 cores)
            sframe = camera.Frame(sw, sh, cam["fovy"], cam["eye"], cam["target"])
            r.set_shard(0, 1, 64, 32)
            got = r.render(sframe, 0, nseed, aovs=False)
            o = oracle_py.OracleScene(sc)
            threads = host_threads()
            pc = r.push_constants(sframe)
            wframe = camera.Frame(sw // 8, sh // 8, cam["fovy"], cam["eye"], cam["target"])
            o.render(wframe, r.push_constants(wframe), r.mSamplingFlags, 0, 1, threads=threads, aovs=False)  # warm-up
            t1 = time.perf_counter()
            ref = o.render(sframe, pc, r.mSamplingFlags, 0, nseed, threads=threads, aovs=False)
            cdt = time.perf_counter() - t1
            a = got["radiance"][..., :3].astype(np.float64)
            b = ref["radiance"][..., :3].astype(np.float64)
            rel = float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b**2).sum()), 1e-300))
            cpu = {
                "value": round(float(ref["ray_count"][0]) / cdt / 1e6, 3),
                "unit": "Mray/s",
                "cores": threads,
                "kind": "port",
                "sample": "%s %dx%d x %d samples, default flags (%d rays, %.2f s)" % (args.scene, sw, sh, nseed, int(ref["ray_count"][0]), cdt),
                "rel_l2_gpu_vs_oracle": rel,
            }
        result = {
            "metric": "Mray/s at 1920x1080x1spp (1M-tri scene)",
            "value": round(rays_all / dt_all / 1e6, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "procedural %s, %d triangles, %dx%d, %d sample(s)/pixel/step, %s, pixel-tile shard 64x32 over %d GPU(s)"
                % (args.scene, sc.triangle_count, W, H, seeds_per_step, "default BDPT flags" if not (args.bdpt_flag or args.max_diffuse_vertices) else "flags %s maxDiffuseVertices %s" % (args.bdpt_flag, args.max_diffuse_vertices), world),
                "rays_per_step": int(rays_all / args.steps),
                "parallelism": "tile-shard x%d" % world if world > 1 else "single GPU",
                "exchange": exchange,
            },
            "host_output_value": host_rate,  # Mray/s with the radiance image copied to host memory inside every call
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result))
    r.close()


if __name__ == "__main__":
    main()
