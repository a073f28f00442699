#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Workload (N=1): the reference's default Cornell-box scene (src/main.cpp:7-84),
path tracing, 784x784 at 1024 spp (the configuration the headline metric is quoted on).  One "step" is one
full render of that image; the scene is already resident in HBM when the timed
region starts and the framebuffer stays in HBM (a torch tensor): `value` does not include
the copy to the host that ends the reference's own wall time -- that rate is reported
beside it as `value_with_framebuffer_on_host`.

N>1 (launched by torch.distributed.run, one process per GPU): the same image is
cut into interleaved column tiles (8 columns or fewer, so that every rank gets the same
number of columns), rank r renders tiles t with t % N == r with
no traffic between GPUs, and one gather (RCCL) brings the framebuffer to rank 0,
which restores column order with one indexed copy.  Total work is fixed, so "scaling" is
"strong".  The line then also carries, per rank, the time its renders took and what the
gather cost, and one more record: the 2048 x 2048 image at 4096 spp (the configuration the
scaling curve is specified on) rendered once by the same ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Algorithmic work of one closest-hit query on the default scene, from the
# reference's own arithmetic: 25.6 box tests x 21 flop + 3.4 triangle tests x
# 100 flop + 0.75 sphere tests x 65 flop, and 36 streamed bytes (ray in 28, hit
# out 8).  Counts per query were measured on the reference's traversal.
# Shading one path vertex (light sample, two BSDF evaluations, BSDF sample): 200 flop.
FLOP_PER_QUERY = 25.6 * 21 + 3.4 * 100 + 0.75 * 65
FLOP_PER_VERTEX = 200.0
BYTES_PER_QUERY = 36.0
PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X vector fp32, FMA counted as 2 flop
PEAK_HBM_GBS = 8000.0


def cpu_baseline(W, H):
    """The reference's own std::thread renderer (oracle/_ref, built from its sources)
    on this host's cores; falls back to our CPU restatement where that build is absent."""
    import numpy as np
    from _oracle import Oracle, Ref, have_ref
    cores = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
        cores = min(cores, aff)
    except Exception:
        pass
    model = "unknown cpu"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    if have_ref():
        R = Ref()
        sr = R.default_scene()
        spp = 64  # about 15 s on 16 host cores
        # the reference prints progress with std::cout: keep fd 1 clean for the JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            t0 = time.perf_counter()
            R.render(sr, W, H, spp, "/tmp/bench_ref_%d" % os.getpid())
            dt = time.perf_counter() - t0
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": min(16, cores),
                "threads": 16, "kind": "reference",
                "sample": "%dx%d at %d spp, Render::renderPathTracing with its 16 std::threads "
                          "(includes its PPM write), %.1f s; host exposes %d cpus (%s)"
                          % (W, H, spp, dt, cores, model)}
    O = Oracle()
    so = O.default_scene()
    threads = min(cores, 16)
    spp = 8
    t0 = time.perf_counter()
    O.render(so, W, H, spp, seed=0, threads=threads, recursive=True)
    dt = time.perf_counter() - t0
    return {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": threads,
            "kind": "port", "sample": "%dx%d at %d spp, %.1f s (%s)" % (W, H, spp, dt, model)}


def timed_render(torch, scene, W, H, spp, dev, profile=True, **kw):
    """One render into a device buffer, after a short one that sizes the workspace."""
    buf = torch.empty(W * H * 3, dtype=torch.float32, device=dev)
    scene.render_device(W, H, max(1, spp // 64), buf.data_ptr(), **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = scene.render_device(W, H, spp, buf.data_ptr(), profile=profile, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"msamples_per_s": W * H * spp / dt / 1e6, "ms": dt * 1e3, "mrays_per_s_unique": st.rays / dt / 1e6,
            "vertices_per_sample": st.vertices / st.samples, "rays_per_sample": st.rays / st.samples,
            "mode": "fused" if st.mode == 2 else "wavefront", "chunk_spp": st.chunk_spp,
            "kernel_ms": {"fused": st.ms_fused, "trace": st.ms_trace, "logic": st.ms_logic,
                          "other": st.ms_other},
            "iterations": st.iterations, "boxes_tested": st.boxes_tested, "prims_tested": st.prims_tested,
            "queue_rays": st.rays - st.samples // max(1, spp)}


# MI355X random-gather rates by where a table is served from (measured per-CU rates times 256 CUs):
# a table that fits the 256 MiB Infinity Cache 7.4 - 8.6 TB/s, a table swept from HBM 6.0 - 6.1 TB/s
GATHER_PEAK_INFINITY_CACHE_GBS = 8600.0
GATHER_PEAK_HBM_GBS = 6100.0


def mesh_case(torch, rt, scenes, dev, seed, n, render):
    """Default scene + an n x n jittered height field: scene build times, closest-hit throughput
    of 2^24 incoherent rays against the bytes the reference's traversal names for them, and
    (render=True) a 784 x 784 x 256 spp render."""
    t0 = time.perf_counter()
    spec = scenes.heightfield_spec(n)
    t1 = time.perf_counter()
    sm = rt.HostScene.from_spec(spec)  # triangles (host objects) + trees (rtr_build_bvh on the device)
    t2 = time.perf_counter()
    sm.flatten_for_upload()            # pointer trees -> the arrays of rtr_scene_desc (host)
    t2b = time.perf_counter()
    sm.device_scene()                  # rtr_scene_create: copy, check and lay out on the device
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n_tri = 2 * n * n
    scene_mb = (2 * n_tri * 32 + n_tri * 64) / 1e6  # what the traversal reads: node records and leaf records
    in_cache = scene_mb < 256 * 1.048576
    nr = 1 << 24
    org, d = scenes.interior_rays(nr, seed=5)
    o_t, d_t = torch.from_numpy(org).to(dev), torch.from_numpy(d).to(dev)
    prim = torch.empty(nr, dtype=torch.int32, device=dev)
    t = torch.empty(nr, dtype=torch.float32, device=dev)
    boxes, tris, sphs = sm.count_work(o_t.data_ptr(), d_t.data_ptr(), nr)
    bytes_per_launch = 32.0 * boxes + 48.0 * tris + 16.0 * sphs + 36.0 * nr
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    stream = torch.cuda.current_stream(dev)
    sm.intersect_device(o_t.data_ptr(), d_t.data_ptr(), nr, prim.data_ptr(), t.data_ptr(),
                        stream=stream.cuda_stream)
    torch.cuda.synchronize()
    reps = 3
    ev[0].record(stream)
    for _ in range(reps):
        sm.intersect_device(o_t.data_ptr(), d_t.data_ptr(), nr, prim.data_ptr(), t.data_ptr(),
                            stream=stream.cuda_stream)
    ev[1].record(stream)
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    peak = GATHER_PEAK_INFINITY_CACHE_GBS if in_cache else GATHER_PEAK_HBM_GBS
    bound = ("l2-miss gather (scene in Infinity Cache)" if in_cache
             else "hbm gather (scene larger than the Infinity Cache)")
    rec = {"workload": "default scene + %dx%d jittered height field (%d triangles, %.1f M tree nodes, %d MB of node and leaf records)"
                       % (n, n, n_tri, 2 * n_tri / 1e6, scene_mb),
           "scene_ms": {"mesh_arrays": (t1 - t0) * 1e3, "objects_and_trees": (t2 - t1) * 1e3,
                        "host_flatten": (t2b - t2) * 1e3, "rtr_scene_create": (t3 - t2b) * 1e3,
                        "tree_builder": os.environ.get("RTR_TREE_BUILDER", "device (rtr_build_bvh)")},
           "trace": {"rays": nr, "mrays_per_s": nr / (ms * 1e-3) / 1e6, "ms_per_launch": ms,
                     "boxes_per_ray": boxes / nr, "triangles_per_ray": tris / nr,
                     "spheres_per_ray": sphs / nr,
                     "roofline": {"kernel": "k_trace_api_stream", "bound": bound,
                                  "achieved": gbs, "peak": peak, "unit": "GB/s",
                                  "frac": gbs / peak, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS, "traffic": None,
                                  "bytes_per_ray": bytes_per_launch / nr,
                                  "note": "algorithmic bytes (32 B per box the reference tests, 48 B per "
                                          "triangle, 16 B per sphere, 36 B per ray in and out), counted "
                                          "on the device for these rays; launches timed with events on the "
                                          "launch stream.  Incoherent fetches of 32- and 64-byte records: "
                                          "priced against the chip's measured random-gather rate from where "
                                          "the scene is served (Infinity Cache 7.4 - 8.6 TB/s, HBM 6.0 - 6.1 "
                                          "TB/s), not against the 8 TB/s streaming peak"}}}
    del o_t, d_t, prim, t
    if render:
        # (no per-kernel events here: they would keep the two streams of the loop from overlapping)
        r = timed_render(torch, sm, 784, 784, 256, dev, profile=False, seed=seed)
        # the render's rays are heavier than the isotropic test rays above: bytes counted per render
        rb = 32.0 * r["boxes_tested"] + 48.0 * r["prims_tested"] + 36.0 * r["queue_rays"]
        ach = rb / (r["ms"] * 1e-3) / 1e9
        r["roofline"] = {"kernel": "k_trace_queue_stream (whole render: path logic, pool fill and drain included)",
                         "bound": bound, "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                         "frac_of_hbm_peak": ach / PEAK_HBM_GBS, "traffic": None,
                         "bytes_per_ray": rb / max(1, r["queue_rays"])}
        rec["render_784x784_256spp"] = r
    sm.close()
    return rec


def extras(torch, rt, dev, seed):
    """The other measurement inputs on one GPU, a few seconds each: the 2048 x 2048 image at
    4096 spp, the specular / microfacet variant with max_depth 16 and continue probability
    0.95, and the default scene plus a height field of 1 M triangles (closest-hit throughput
    against the bytes its traversal reads, then a render) and of 4.5 M triangles (closest hit only)."""
    from raytracingrenderer_amd import scenes
    out = []
    sg = rt.default_scene()
    r = timed_render(torch, sg, 2048, 2048, 4096, dev, seed=seed)
    r["workload"] = "default Cornell box, 2048x2048 at 4096 spp, one GPU"
    out.append(r)
    sg.close()

    sv = rt.HostScene.from_spec(scenes.variant_scene_spec())
    r = timed_render(torch, sv, 784, 784, 1024, dev, seed=seed, max_depth=16, rr_prob=0.95)
    r["workload"] = ("Cornell box with specular / microfacet spheres, max_depth 16, continue "
                     "probability 0.95 (both knobs are extensions: the reference hard-codes 0.6 "
                     "and has no depth cap), 784x784 at 1024 spp")
    out.append(r)
    sv.close()

    out.append(mesh_case(torch, rt, scenes, dev, seed, 708, render=True))
    # more than the 256 MiB Infinity Cache holds: what the same traversal reaches from HBM
    out.append(mesh_case(torch, rt, scenes, dev, seed, 1500, render=False))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=784)
    ap.add_argument("--height", type=int, default=784)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-gather", action="store_true",
                    help="with one rank under torch.distributed.run: still initialise the process group "
                         "and run barrier, gather and reductions (the N>1 code path on one GPU)")
    ap.add_argument("--dump-image", default="",
                    help="rank 0 saves the last step's whole image (W*H x 3 float32, .npy) here")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other measurement inputs reported after the headline line's timing")
    ap.add_argument("--big", default="2048x2048x4096",
                    help="N > 1: WxHxSPP of the one render added as an extra record (the configuration the "
                         "scaling curve is specified on)")
    ap.add_argument("--pool", type=int, default=0, help="path pool size (0 = library default)")
    ap.add_argument("--chunk", type=int, default=0, help="samples per partial sum (0 = default)")
    ap.add_argument("--mode", type=int, default=0,
                    help="0: library default (fused kernel for this scene), 1: wavefront, 2: fused")
    ap.add_argument("--streams", type=int, default=0,
                    help="wavefront mode: 1 = kernels never overlap, else two streams")
    args = ap.parse_args()

    import numpy as np
    import torch
    import raytracingrenderer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run"
                             % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    collective = world > 1 or args.rehearse_gather
    if collective:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    W, H, spp = args.width, args.height, args.spp
    # column tiles are dealt round-robin to the ranks: the widest tile (up to 8 columns)
    # that gives every rank the same number of columns, else 8
    tile = next((t for t in (8, 7, 6, 5, 4, 3, 2, 1) if W % t == 0 and (W // t) % world == 0), 8)
    scene = rt.default_scene()
    scene.set_device(local_rank)
    scene.device_scene()  # upload now: the scene is resident before the timed region

    from raytracingrenderer_amd.distributed import ColumnGather, local_buffer
    local = local_buffer(torch, W, H, world, tile, dev)
    gather = (ColumnGather(torch, dist, W, H, rank, world, tile, dev, dst=0, always_collective=True)
              if collective else None)

    agg = {"rays": 0, "vertices": 0, "ms_trace": 0.0, "ms_logic": 0.0, "ms_other": 0.0,
           "ms_fused": 0.0, "trace_launches": 0, "fused_launches": 0, "iterations": 0,
           "samples": 0, "mode": 0, "render_ms": 0.0, "gather_ms": 0.0}

    last = {"image": None}

    def step(record):
        t_a = time.perf_counter()
        st = scene.render_device(W, H, spp, local.data_ptr(), seed=args.seed, part_index=rank,
                                 part_count=world, tile_cols=tile, profile=True,
                                 pool_size=args.pool, chunk_spp=args.chunk, mode=args.mode,
                                 streams=args.streams)  # returns when this rank's columns are done
        t_b = time.perf_counter()
        if gather is not None:
            last["image"] = gather(local)  # one RCCL gather + one indexed copy on rank 0
            # the gather reads `local` on torch's stream, the next render writes it on the
            # library's: wait for the gather before going on
            torch.cuda.current_stream(dev).synchronize()
        t_c = time.perf_counter()
        if record:
            agg["render_ms"] += (t_b - t_a) * 1e3
            agg["gather_ms"] += (t_c - t_b) * 1e3  # includes waiting for the slowest rank
            agg["rays"] += st.rays
            agg["vertices"] += st.vertices
            agg["ms_trace"] += st.ms_trace
            agg["ms_logic"] += st.ms_logic
            agg["ms_other"] += st.ms_other
            agg["ms_fused"] += st.ms_fused
            agg["trace_launches"] += st.trace_launches
            agg["fused_launches"] += st.fused_launches
            agg["iterations"] += st.iterations
            agg["samples"] += st.samples
            agg["mode"] = st.mode
            agg["chunk_spp"] = st.chunk_spp
        return st

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(agg["rays"]), float(agg["vertices"])], dtype=torch.float64,
                     device=dev)
    if collective:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_total, vert_total = float(tsum[1]), float(tsum[2])
    else:
        rays_total, vert_total = float(t[1]), float(t[2])

    # what each rank's renders took, and what the gather cost it (per step)
    mine = torch.tensor([agg["render_ms"] / max(1, args.steps), agg["gather_ms"] / max(1, args.steps)],
                        dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(world)] if collective else [mine]
    if collective:
        dist.all_gather(per_rank, mine)
    per_rank = [[float(x[0]), float(x[1])] for x in per_rank]

    big = None
    if collective and not args.no_extras:
        # the configuration the scaling curve is specified on, once, by the same ranks
        W2, H2, spp2 = (int(x) for x in args.big.lower().split("x"))
        tile2 = next((t for t in (8, 7, 6, 5, 4, 3, 2, 1) if W2 % t == 0 and (W2 // t) % world == 0), 8)
        local2 = local_buffer(torch, W2, H2, world, tile2, dev)
        gather2 = ColumnGather(torch, dist, W2, H2, rank, world, tile2, dev, dst=0, always_collective=True)
        scene.render_device(W2, H2, 8, local2.data_ptr(), seed=args.seed, part_index=rank, part_count=world,
                            tile_cols=tile2)
        fence()
        t_a = time.perf_counter()
        scene.render_device(W2, H2, spp2, local2.data_ptr(), seed=args.seed, part_index=rank,
                            part_count=world, tile_cols=tile2)
        t_b = time.perf_counter()
        gather2(local2)
        fence()
        t_c = time.perf_counter()
        mine2 = torch.tensor([(t_b - t_a) * 1e3, (t_c - t_a) * 1e3], dtype=torch.float64, device=dev)
        all2 = [mine2.clone() for _ in range(world)]
        dist.all_gather(all2, mine2)
        renders = [float(x[0]) for x in all2]
        wall = max(float(x[1]) for x in all2)
        big = {"workload": "default Cornell box, %dx%d at %d spp, %d rank(s), one gather" % (W2, H2, spp2, world),
               "ms": wall, "msamples_per_s": W2 * H2 * spp2 / wall / 1e3, "tile_cols": tile2,
               "per_rank_render_ms": renders, "imbalance_max_over_mean": max(renders) / (sum(renders) / len(renders)),
               "gather_and_wait_ms_rank0": float(all2[0][1]) - float(all2[0][0])}
        del local2, gather2

    if rank == 0 and args.dump_image:
        img = last["image"] if last["image"] is not None else local[: W * H]
        np.save(args.dump_image, img.reshape(-1, 3).cpu().numpy())
    if rank == 0:
        samples = float(W) * H * spp * args.steps
        value = samples / elapsed / 1e6
        # Dominant kernel, timed with HIP events on the library's own launch stream
        # (rank 0's launches).  Fused mode: k_render_fused, one launch per step, does
        # all sampling (shading and both closest-hit queries of every vertex).
        # Wavefront mode: k_trace_queue, one launch per bounce (the primary-ray launch
        # of each step is in the average too, with its own ray count).
        fused = agg["mode"] == 2
        chunk_used = agg.get("chunk_spp", 0) or 1
        if fused:
            kernel = "k_render_fused"
            launches = max(agg["fused_launches"], 1)
            avg_ms = agg["ms_fused"] / launches
            queries = (agg["rays"] - agg["samples"] / spp) / launches  # rays minus the primary rays
            flop_per_launch = FLOP_PER_QUERY * queries + FLOP_PER_VERTEX * agg["vertices"] / launches
        else:
            kernel = "k_trace_queue"
            launches = max(agg["trace_launches"], 1)
            avg_ms = agg["ms_trace"] / launches
            queries = agg["rays"] / launches
            flop_per_launch = FLOP_PER_QUERY * queries
        rays_per_launch = queries
        tflops = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r03", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kernel + "_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": kernel, "bound": "valu", "achieved": tflops,
                    "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                    "frac": tflops / PEAK_FP32_VECTOR_TFLOPS,
                    "traffic": traffic,
                    "avg_launch_ms": avg_ms, "rays_per_launch": rays_per_launch,
                    "flop_per_query": FLOP_PER_QUERY, "flop_per_vertex": FLOP_PER_VERTEX,
                    "note": "fp32 VALU-issue bound (the 37-node scene sits in LDS and the "
                            "scalar cache); parity forbids FMA contraction, so at most half "
                            "of the FMA-counted peak is reachable.  `traffic` is not measured in this "
                            "run: it is the FETCH_SIZE x 2 + WRITE_SIZE of the committed counter passes "
                            "over this command (profiles/r03/traffic.json)"}
        if fused and avg_ms > 0:
            # Issue account: the launch's wave-level instruction counts by class (SQ_INSTS_* of
            # committed counter passes) priced with the cycles per instruction and SIMD that
            # tools/ubench/issue_cycles.hip measures for each class at four waves per SIMD
            # (tools/issue_account.py -> profiles/r03/issue_account.json); nothing is a literal here.
            try:
                acc = json.load(open(os.path.join(ROOT, "profiles", "r03", "issue_account.json")))
                scale = float(W) * H * spp / acc["samples_counted"]
                n_inst = acc["valu_instructions"] * scale
                ns = avg_ms * 1e6 / (n_inst / 1024.0)
                floor_ns = acc["floor_cycles_of_this_mix_per_simd"] * scale / acc["shader_ghz"]
                roofline["vector_issue"] = {
                    "valu_instructions_per_launch": n_inst,
                    "ns_per_wave_instruction_and_simd": ns,
                    "plain_fp32_floor_ns": acc["plain_fp32_cycles"] / acc["shader_ghz"],
                    "plain_fp32_floor_cycles": acc["plain_fp32_cycles"],
                    "floor_of_this_instruction_mix_ms": floor_ns * 1e-6,
                    "frac_of_mix_floor": floor_ns * 1e-6 / avg_ms,
                    "source": "profiles/r03/issue_account.json"}
            except Exception:
                pass
        if fused:
            # nothing streams through HBM in this kernel by design (path state, rays and hits
            # stay in registers): per work unit (pixel x chunk of samples) it reads the pixel's
            # first vertex (three 16-byte records and its list entry) and writes one partial sum
            roofline["algorithmic_hbm_bytes_per_launch"] = (16.0 + 52.0) * W * H * -(-spp // chunk_used)
        else:
            gbs = BYTES_PER_QUERY * rays_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            roofline["hbm"] = {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": gbs / PEAK_HBM_GBS, "bytes_per_query": BYTES_PER_QUERY}
        out = {
            "metric": "Msamples/sec (paths/s), Cornell box path tracing",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "default Cornell box (reference src/main.cpp), path tracing, "
                                   "%dx%d at %d spp; scene resident and framebuffer left in HBM (with the "
                                   "copy to the host: value_with_framebuffer_on_host)" % (W, H, spp),
                       "width": W, "height": H, "spp": spp,
                       "parallelism": "columns, %d-col interleaved tiles x %d gpu, 1 gather" % (tile, world)},
            "mrays_per_s_unique": rays_total / elapsed / 1e6,
            "mrays_per_s_reference_equivalent": 5.29 * samples / elapsed / 1e6,
            "iterations_per_step_rank0": agg["iterations"] / args.steps,
            "rays_per_sample": rays_total / samples,
            "vertices_per_sample": vert_total / samples,
            "mode": "fused" if fused else "wavefront",
            "kernel_ms_per_step_rank0": {"fused": agg["ms_fused"] / args.steps,
                                         "trace": agg["ms_trace"] / args.steps,
                                         "logic": agg["ms_logic"] / args.steps,
                                         "other": agg["ms_other"] / args.steps},
            "roofline": roofline,
        }
        if collective:
            renders = [x[0] for x in per_rank]
            out["multi_gpu"] = {"tile_cols": tile, "per_rank_render_ms": renders,
                                "render_ms_min": min(renders), "render_ms_max": max(renders),
                                "imbalance_max_over_mean": max(renders) / (sum(renders) / len(renders)),
                                "gather_and_wait_ms_per_rank": [x[1] for x in per_rank],
                                "gather_ms": min(x[1] for x in per_rank),
                                "note": "per step; a rank's gather time includes waiting for the slowest "
                                        "rank, so the smallest one is what the gather itself costs"}
            if big is not None:
                out["extra"] = [big]
        if not collective and not args.no_extras:
            # the reference's wall ends with the framebuffer on the host: the same render
            # through the host-output entry point (one D2H copy more)
            try:
                scene.render(W, H, max(1, spp // 64), seed=args.seed, pool_size=args.pool,
                             chunk_spp=args.chunk, mode=args.mode, streams=args.streams)
                t0 = time.perf_counter()
                scene.render(W, H, spp, seed=args.seed, pool_size=args.pool, chunk_spp=args.chunk,
                             mode=args.mode, streams=args.streams)
                dt = time.perf_counter() - t0
                out["value_with_framebuffer_on_host"] = {"value": W * H * spp / dt / 1e6,
                                                         "unit": "Msamples/s", "ms": dt * 1e3}
            except Exception as e:
                out["value_with_framebuffer_on_host"] = {"value": None, "error": repr(e)}
        if not collective and not args.no_extras:
            try:
                out["extra"] = extras(torch, rt, dev, args.seed)
            except Exception as e:  # reports, never a reason to lose the line
                out["extra"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(W, H)
                out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a report, never a reason to lose the line
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 0,
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
