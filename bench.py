#!/usr/bin/env python3
"""Headline benchmark: megapixels/s of the SLIC + MPEG-7 descriptor hot path on MI355X.

Workload (BASELINE.json configs[1]): a batch of synthetic 1920x1080 BGR images
(Voronoi mosaic + noise), K=400 superpixels, num_iter=10, colour weight 15,
then the full descriptor set.  One "step" = one pass of the whole path over one batch
whose input is already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the contract in the task description).  Images are
independent, so ranks shard the batch with no data-path collective ("weak" scaling: the
per-GPU batch is fixed); the descriptor vectors are all-gathered over RCCL at the end of
each step, as BASELINE.json's 8-GPU config asks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SLIC_ROUND_BYTES_PER_PX = 7    # 3 B Lab read + 4 B label write


def synth_batch(B, H, W, K, seed, device):
    """Voronoi mosaic (ceil(K/2) sites, uniform BGR colours) + N(0, 8^2), u8, on device."""
    g = torch.Generator(device=device)
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=device)
    ns = (K + 1) // 2
    sub = 8
    hs, ws = (H + sub - 1) // sub, (W + sub - 1) // sub
    yy = (torch.arange(hs, device=device, dtype=torch.float32) * sub).view(hs, 1, 1)
    xx = (torch.arange(ws, device=device, dtype=torch.float32) * sub).view(1, ws, 1)
    for b in range(B):
        g.manual_seed(seed + b)
        sx = torch.rand(ns, generator=g, device=device) * W
        sy = torch.rand(ns, generator=g, device=device) * H
        cols = torch.randint(0, 256, (ns, 3), generator=g, device=device).float()
        # nearest site per grid point, in row chunks (the 16384^2 / 25,000-site mosaic would
        # need 400 GB as one [hs, ws, ns] tensor)
        rows = max(1, (1 << 28) // (ws * ns))
        idx = torch.cat([((xx - sx.view(1, 1, ns)) ** 2 + (yy[r:r + rows] - sy.view(1, 1, ns)) ** 2)
                         .argmin(dim=2) for r in range(0, hs, rows)], dim=0)
        idx = idx.repeat_interleave(sub, 0).repeat_interleave(sub, 1)[:H, :W]
        img = cols[idx] + torch.randn((H, W, 3), generator=g, device=device) * 8.0
        out[b] = img.round().clamp_(0, 255).to(torch.uint8)
    return out


def cpu_baseline_numpy(H, W, K, cw, k=8):
    """The numpy oracle, which keeps the reference's own cost structure (per-centre window ops
    + per-label full-image scans, slic.py:88-126; per-segment Lloyd loops + per-colour
    connected components, mpeg7Descriptors.py:76-130), on a bounded sample of the same workload:
    one image, 2 of the 9 SLIC rounds, the whole-image descriptors, dominant colours of 12
    segments; each part is scaled to its full count."""
    from oracle.synth import voronoi_image
    from oracle.color import bgr2lab_u8
    from oracle.slic_ref import SLICOracle
    from oracle import mpeg7_std as M
    from oracle import dcd_ref
    rounds, nsample = 2, 12
    img = voronoi_image(H, W, K, seed=1234)
    t0 = time.perf_counter()
    lab = bgr2lab_u8(img)
    t_lab = time.perf_counter() - t0
    t0 = time.perf_counter()
    o = SLICOracle(lab=lab, num_clusters=K, colour_weight=cw, num_iter=rounds + 1, update="scan")
    o.run()
    t_slic = (time.perf_counter() - t0) * 9.0 / rounds            # cost is linear in rounds
    t0 = time.perf_counter()
    M.colour_layout(img)
    M.edge_histogram_counts(img)
    M.colour_structure_counts(img)
    t_desc = time.perf_counter() - t0
    labels = o.get_clusters()
    segs = [s for s in np.unique(labels) if s >= 0]
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    done = 0
    for s in segs[::max(1, len(segs) // nsample)][:nsample]:
        idx = labels == s
        pts = lab[idx]
        if pts.shape[0] < k:
            continue
        dcd_ref.dominant_colour(pts, np.transpose(np.nonzero(idx)),
                                rng.choice(pts.shape[0], k, replace=False), labels.shape)
        done += 1
    t_dcd = (time.perf_counter() - t0) * len(segs) / max(done, 1)
    full = t_lab + t_slic + t_desc + t_dcd
    return {"value": round(H * W / 1e6 / full, 4), "unit": "megapixels/sec", "cores": 1,
            "kind": "port",
            "sample": "1 image %dx%d K=%d, numpy oracle (the reference's cost structure), 1 thread: "
                      "Lab %.1f s + SLIC %.1f s (2 of 9 rounds timed, scaled) + CLD/EHD/CSD %.1f s + "
                      "dominant colours %.1f s (%d of %d segments timed, scaled) = %.0f s for the "
                      "whole path" % (W, H, K, t_lab, t_slic, t_desc, t_dcd, done, len(segs), full)}, \
        (t_lab, t_desc)


def cpu_baseline(H, W, K, cw, k=8):
    """Two CPU figures for the same workload, on the host cores of this box in this run:
    `cpu_baseline` = the C restatement (oracle/c: SLIC + per-superpixel dominant colours, bit
    identical to the numpy oracle, tests/test_oracle_c.py) with one image per OpenMP thread, plus
    the numpy Lab and CLD / EHD / CSD time of one image per thread -- the fair multi-core figure;
    `cpu_baseline_numpy` = the numpy oracle with the reference's cost structure on one thread."""
    from oracle.synth import voronoi_image
    from oracle.color import bgr2lab_u8
    from oracle import c_oracle
    slow, (t_lab, t_desc) = cpu_baseline_numpy(H, W, K, cw, k)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, 16))
    labs = np.stack([bgr2lab_u8(voronoi_image(H, W, K, seed=1234 + i)) for i in range(threads)])
    t0 = time.perf_counter()
    _, used, pixel_rounds = c_oracle.path_batch(labs, K, cw, 10, k, threads)
    t_c = time.perf_counter() - t0
    wall = t_c + t_lab + t_desc                    # every thread also converts / describes its image
    fast = {"value": round(threads * H * W / 1e6 / wall, 3), "unit": "megapixels/sec", "cores": used,
            "kind": "port",
            "sample": "%d images %dx%d K=%d, one per thread: C restatement (gcc -O2 -fopenmp) of SLIC "
                      "(9 rounds, literal float64 distance) + dominant colours k=%d of every superpixel "
                      "(%.1f Lloyd rounds per pixel) %.1f s, + numpy Lab %.2f s and CLD/EHD/CSD %.2f s "
                      "per image" % (threads, W, H, K, k, pixel_rounds / float(threads * H * W), t_c,
                                     t_lab, t_desc)}
    return fast, slow


def cpu_baseline_rag(cw=15):
    """CPU leg of configs[2] on a bounded sample: one 1920x1080 quarter of a 4K frame (K = 500,
    the same S = 64 grid), one thread: numpy median on a 270-row band (scaled x4), numpy Lab, the C
    restatement of SLIC (oracle/c), the numpy oracle of rag_segmentation.py:38-184."""
    from oracle.synth import voronoi_image
    from oracle.color import bgr2lab_u8, median_blur
    from oracle import c_oracle
    from oracle.rag_ref import RAGOracle
    H, W, K = 1080, 1920, 500
    img = voronoi_image(H, W, K, seed=4321)
    t0 = time.perf_counter()
    median_blur(img[:276], 7)
    t_med = (time.perf_counter() - t0) * H / 270.0
    t0 = time.perf_counter()
    lab = bgr2lab_u8(img)
    t_lab = time.perf_counter() - t0
    t0 = time.perf_counter()
    labels, _ = c_oracle.slic_run(lab, K, cw, 10)
    t_slic = time.perf_counter() - t0
    ora = RAGOracle.__new__(RAGOracle)
    ora.image, ora.image_lab, ora.image_mean, ora.slic_clust_num = img, lab, img.copy(), K
    t0 = time.perf_counter()
    cl = labels.astype(np.float64)
    col = ora.slic_mean_lab(cl, nchange=True)
    ng = ora.neighbours_regions(cl)
    ora.concat_similar_regs(ora.find_edges(ng, col), cl, c_factor=0.65)
    t_rag = time.perf_counter() - t0
    full = t_med + t_lab + t_slic + t_rag
    return {"value": round(H * W / 1e6 / full, 3), "unit": "megapixels/sec", "cores": 1, "kind": "port",
            "sample": "one 1920x1080 quarter of a 4K frame, K=500 (S=64 as at 4K), 1 thread: numpy "
                      "median %.1f s (270-row band timed, scaled) + numpy Lab %.2f s + C SLIC %.1f s + "
                      "numpy RAG means / adjacency / merge %.2f s" % (t_med, t_lab, t_slic, t_rag)}


def cpu_baseline_mosaic(cw=15, k=8):
    """CPU leg of configs[4] on a bounded sample: a 2048x2048 tile of the mosaic (K = 781, the same
    S = 73 grid), one thread: numpy Lab + the C restatement of SLIC and of the dominant colours
    of every superpixel."""
    from oracle.synth import voronoi_image
    from oracle.color import bgr2lab_u8
    from oracle import c_oracle
    H = W = 2048
    K = 781
    img = voronoi_image(H, W, K, seed=99)
    t0 = time.perf_counter()
    lab = bgr2lab_u8(img)
    t_lab = time.perf_counter() - t0
    t0 = time.perf_counter()
    c_oracle.path_batch(lab[None], K, cw, 10, k, 1)
    t_c = time.perf_counter() - t0
    return {"value": round(H * W / 1e6 / (t_lab + t_c), 3), "unit": "megapixels/sec", "cores": 1,
            "kind": "port",
            "sample": "one 2048x2048 tile of the mosaic, K=781 (S=73 as at 16384^2), 1 thread: numpy Lab "
                      "%.2f s + C SLIC and dominant colours k=%d of every superpixel %.1f s" % (t_lab, k, t_c)}


def roofline_line(kernel, ms, bytes_px, npix, traffic=None, profile=None, note=None, bound=None,
                  launches=1):
    """Priced against the HBM roofline (nothing on this path is a contraction); `limited_by` names
    what the kernel is measured to be bound by when that is not HBM bandwidth."""
    ach = bytes_px * npix / (ms * 1e-3) / 1e9
    d = {"bound": "hbm", "limited_by": bound or "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
         "launch_ms": round(ms, 4), "bytes_per_px": bytes_px,
         "algorithmic_bytes_per_launch": int(bytes_px * npix), "launches_per_step": launches}
    if profile:
        d["profile"] = profile
    if note:
        d["note"] = note
    return d


def pmc_traffic(key, kernel, npix):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        e = json.load(f).get(key, {}).get(kernel)
    return int(e["bytes_per_px"] * npix) if e else None


def dcd_traffic(key, npix):
    """HBM bytes of one imgan_dcd_segments_listed call: the sort pass, the plan and the three tier
    launches (pmc_traffic.json holds the mean over the launches of a kernel name)."""
    parts = [pmc_traffic(key, "seg_scatter_kernel", npix), pmc_traffic(key, "seg_plan_kernel", npix),
             pmc_traffic(key, "dcd_dedup_kernel", npix)]
    if any(p is None for p in parts):
        return None
    return parts[0] + parts[1] + 3 * parts[2]


def run_config_rag(steps, warmup):
    """configs[2]: 16 x 3840x2160, K = 2000, median 7 + Lab + SLIC + RAG merge + relabel."""
    from imgan.pipeline import RagPipeline
    B, H, W, K = 16, 2160, 3840, 2000
    bgr = synth_batch(B, H, W, K, 4321, "cuda")
    pipe = RagPipeline(B, H, W, K)
    for _ in range(warmup):
        pipe.step(bgr, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(bgr, timed=True, check=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    redone = pipe.rag.finish(pipe.lab, pipe.slic.labels, pipe.c_factor) if pipe.rag is not None else 0
    ms = pipe.stage_times_ms()
    npix = B * H * W
    prof = "profiles/r03_rag4k_kernel_summary.txt"
    path_bytes = 6 + 6 + 63 + 7 + 8
    rec = {"config": "configs[2]: %d x %dx%d BGR, K=%d (n=%d, S=%d), median 7x7 + Lab + SLIC (9 rounds) "
                     "+ per-label means and adjacency + MST merge (c=0.65, on the device) + relabel" %
                     (B, W, H, K, pipe.slic.n, pipe.slic.S),
           "value": round(npix * steps / dt / 1e6, 2), "unit": "megapixels/sec",
           "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "images_redone_on_host": int(redone),
           "roofline": roofline_line("slic_assign_kernel", ms["slic_assign"], 7, npix,
                                     traffic=pmc_traffic("3840x2160_K2000", "slic_assign_kernel", npix),
                                     profile=prof, launches=9,
                                     note="largest share of the step (9 launches)"),
           "roofline_kernels": [
               roofline_line("median_bits_kernel<7>", ms["median"], 6, npix, profile=prof, bound="valu",
                             traffic=pmc_traffic("3840x2160_K2000", "median_bits_kernel", npix),
                             note="bit-sliced selection, instruction bound by design"),
               roofline_line("bgr2lab_kernel", ms["bgr2lab"], 6, npix, profile=prof),
               roofline_line("rag_accumulate_kernel<false> (adjacency; sums from the last SLIC round) "
                             "+ rag_merge_kernel", ms["rag_accumulate+merge"], 7,
                             npix, profile=prof, bound="atomics/latency"),
               roofline_line("relabel4_kernel", ms["relabel"], 8, npix, profile=prof)],
           "path": {"bytes_per_px": path_bytes,
                    "achieved": round(path_bytes * npix * steps / dt / 1e9, 1), "unit": "GB/s",
                    "frac": round(path_bytes * npix * steps / dt / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "SURVEY 8d: 6 median + 6 Lab + 63 SLIC + 7 RAG + 8 relabel"},
           "cpu_baseline": cpu_baseline_rag()}
    del pipe, bgr
    torch.cuda.empty_cache()
    return rec


def run_config_mosaic(steps, warmup):
    """configs[4] on one GPU: one 16384x16384 mosaic, K = 50,000, descriptors per superpixel."""
    from imgan.pipeline import FullPipeline
    B, H, W, K = 1, 16384, 16384, 50000
    bgr = synth_batch(B, H, W, K, 99, "cuda")
    pipe = FullPipeline(B, H, W, K, colour_weight=15, num_iter=10, dominant=8, per_region=True)
    for _ in range(warmup):
        pipe.step(bgr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(bgr, timed=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    segments_left = pipe.finish()
    ms = pipe.stage_times_ms()
    npix = B * H * W
    t_bar = pipe.mean_lloyd_rounds()
    dcd_bytes = round(8.0 * t_bar + 13.0, 1)
    prof = "profiles/r03_mosaic_kernel_summary.txt"
    path_bytes = 6 + 63 + pipe.descriptor_bytes_per_px() + dcd_bytes
    rec = {"config": "configs[4] (one GPU's share): %d x %dx%d BGR, K=%d (n=%d, S=%d), Lab + SLIC + "
                     "CLD / EHD / CSD and dominant colours k=8 of every superpixel (%.1f Lloyd rounds per pixel)" %
                     (B, W, H, K, pipe.n, pipe.slic.S, t_bar),
           "value": round(npix * steps / dt / 1e6, 2), "unit": "megapixels/sec",
           "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "dcd_segments_at_round_cap": pipe.segments_at_round_cap(),
           "dcd_segments_left_to_streaming_rounds": int(segments_left),
           "roofline": roofline_line("dcd_dedup_kernel x3 tiers (+seg_scatter_kernel, seg_plan_kernel)",
                                     ms["dcd_segments"], dcd_bytes, npix, profile=prof, bound="valu/lds",
                                     note="timed around the C call (imgan_dcd_segments_listed); SURVEY 8d "
                                          "figure 8*T+13*P; the Lloyd rounds run in LDS, compulsory HBM "
                                          "traffic is 24 B/px (sort pass in and out, lists in, assignment out)"),
           "roofline_kernels": [roofline_line("slic_assign_kernel", ms["slic_assign"], 7, npix, profile=prof,
                                              launches=9),
                                roofline_line("region_desc_kernel (+cld_finalise)", ms["region_desc"], 21, npix,
                                              profile=prof, bound="lds/valu",
                                              note="CLD + EHD + CSD per superpixel in one pass, priced at "
                                                   "SURVEY 8d's 3 x 7 B/px"),
                                roofline_line("bgr2lab_kernel", ms["bgr2lab"], 6, npix, profile=prof)],
           "regions_too_large_for_region_desc": int(pipe.region_desc.too_big.item()),
           "path": {"bytes_per_px": round(path_bytes, 1),
                    "achieved": round(path_bytes * npix * steps / dt / 1e9, 1), "unit": "GB/s",
                    "frac": round(path_bytes * npix * steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
           "cpu_baseline": cpu_baseline_mosaic()}
    del pipe, bgr
    torch.cuda.empty_cache()
    return rec


def run_config_bear(steps, warmup, cw=15, k=8):
    """configs[0]: the reference's own CPU case, slic.py + mpeg7Descriptors on data/bear.jpg at full
    size (1573 x 1050, K = 200 -> S = 90, 176 superpixels).  The photograph does not travel with the
    repository, so a synthetic image of that shape stands in; GPU step and CPU leg on the same image
    size (the CPU leg is the whole workload here: one image)."""
    from imgan.pipeline import FullPipeline
    from oracle.synth import voronoi_image
    from oracle.color import bgr2lab_u8
    from oracle import c_oracle, mpeg7_std as M
    B, H, W, K = 1, 1050, 1573, 200
    bgr = synth_batch(B, H, W, K, 7, "cuda")
    pipe = FullPipeline(B, H, W, K, colour_weight=cw, num_iter=10, dominant=k)
    for _ in range(warmup):
        pipe.step(bgr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(bgr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img = voronoi_image(H, W, K, seed=7)
    t0 = time.perf_counter()
    lab = bgr2lab_u8(img)
    c_oracle.path_batch(lab[None], K, cw, 10, k, 1)
    M.colour_layout(img)
    M.edge_histogram_counts(img)
    M.colour_structure_counts(img)
    t_cpu = time.perf_counter() - t0
    npix = H * W
    return {"config": "configs[0]: one %dx%d image (bear.jpg's size, synthetic), K=%d (n=%d, S=%d), Lab + SLIC + "
                      "CLD / EHD / CSD + dominant colours k=%d per superpixel; launch-latency bound at this size" %
                      (W, H, K, pipe.n, pipe.slic.S, k),
            "value": round(npix * steps / dt / 1e6, 2), "unit": "megapixels/sec",
            "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "cpu_baseline": {"value": round(npix / 1e6 / t_cpu, 3), "unit": "megapixels/sec", "cores": 1,
                             "kind": "port",
                             "sample": "the same workload, whole: numpy Lab + C SLIC and dominant colours + numpy "
                                       "CLD / EHD / CSD, 1 thread, %.2f s" % t_cpu}}


def spawn_ranks(args):
    """`python bench.py --gpus N` on its own: start N ranks (one per GPU, RCCL) as a child job and
    pass its output and exit code on.  Nothing in this process has touched the GPU yet."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--clusters", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the configs[2] / configs[4] sub-records (N = 1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        rehearsal = os.environ.get("IMGAN_BENCH_REHEARSAL") == "1"   # all ranks on device 0, gloo
        if torch.cuda.device_count() < args.gpus and not rehearsal:  # counting does not initialise HIP
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" %
                             (args.gpus, torch.cuda.device_count()))
        raise SystemExit(spawn_ranks(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path is the only path)")
    # IMGAN_BENCH_REHEARSAL=1: every rank on device 0 over gloo, to walk the N > 1 control flow on
    # a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("IMGAN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # IMGAN_BENCH_FORCE_DIST=1: take the RCCL path (process group, barrier, all-gather, max-reduce)
    # with a single rank too, to exercise it on a one-GPU box; never set by the driver
    force_dist = os.environ.get("IMGAN_BENCH_FORCE_DIST") == "1" and world == 1
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    # RCCL prints a version banner on stdout when its first communicator comes up; stdout carries
    # the one JSON line, so file descriptor 1 points at stderr until the result is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    B, H, W, K = args.batch, args.height, args.width, args.clusters
    cw, num_iter = 15, 10
    bgr = synth_batch(B, H, W, K, 1234 + rank * B, "cuda")
    from imgan.pipeline import FullPipeline
    from imgan.shard import gather_descriptors_async
    from imgan._lib import LIB_PATH
    pipe = FullPipeline(B, H, W, K, colour_weight=cw, num_iter=num_iter, dominant=8)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    index, pending = None, None
    for _ in range(args.warmup):
        pipe.step(bgr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image_rows, region_rows = pipe.step(bgr, timed=True)
        # one collective per step: the descriptor rows of every rank -> replicated index.  It is
        # left in flight (RCCL's own stream) while the next step is enqueued, and joined after.
        if pending is not None:
            index = (pending[0].result(), pending[1].result())
        pending = (gather_descriptors_async(image_rows, n_total=B * world),
                   gather_descriptors_async(region_rows, n_total=B * world))
    index = (pending[0].result(), pending[1].result()) if pending is not None else None
    barrier()
    dt = time.perf_counter() - t0
    # what the collective costs on its own (after the timed region: the timed steps overlap it with
    # the next step's kernels): the two all-gathers of one step, issued and joined back to back
    rccl = None
    if dist.is_initialized():
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        reps = 5
        ev[0].record()
        for _ in range(reps):
            a = gather_descriptors_async(image_rows, n_total=B * world)
            b = gather_descriptors_async(region_rows, n_total=B * world)
            a.result(), b.result()
        ev[1].record()
        torch.cuda.synchronize()
        row_bytes = image_rows[0].numel() * image_rows.element_size() + \
            region_rows[0].numel() * region_rows.element_size()
        rccl = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                "collectives_per_step": 2,
                "allgather_bytes_per_step": int(row_bytes * B * world),
                "allgather_bytes_per_rank_contributed": int(row_bytes * B),
                "allgather_ms": round(ev[0].elapsed_time(ev[1]) / reps, 4),
                "index_rows": [int(index[0].shape[0]), int(index[1].shape[0])],
                "note": "descriptor rows of every rank -> replicated index; in the timed steps the "
                        "collective runs on the backend's stream under the next step's kernels, "
                        "allgather_ms is its cost alone (issued and joined back to back)"}
    segments_left = pipe.finish()                  # superpixels beyond the on-chip pass (deferred check)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        npix = B * H * W
        key = "%dx%d_K%d" % (W, H, K)
        prof = "profiles/r03_bench_kernel_summary.txt"
        stage_ms = pipe.stage_times_ms()           # HIP events on the launch stream, per C call
        # mean Lloyd rounds per pixel (a segment stops at its own round): the T of SURVEY 8d
        t_bar = pipe.mean_lloyd_rounds()
        dcd_survey = round(8.0 * t_bar + 13.0, 1)
        desc_bytes = pipe.descriptor_bytes_per_px()
        others = [
            roofline_line("slic_assign_kernel", stage_ms["slic_assign"], SLIC_ROUND_BYTES_PER_PX, npix,
                          traffic=pmc_traffic(key, "slic_assign_kernel", npix), profile=prof,
                          launches=num_iter - 1,
                          bound="valu",
                          note="moves the most HBM bytes of the step (9 x 7 B/px); bound by the vector pipe: "
                               "SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles = 0.85 (profiles/r03_pmc_B64.txt), its "
                               "integer ops (v_dot4, v_mad_i32_i24, v_med3, v_min) issue at 4 cycles per wave64 "
                               "(profiles/r03_issue_rate.txt)"),
            roofline_line("bgr2lab_kernel", stage_ms["bgr2lab"], 6, npix, profile=prof,
                          traffic=pmc_traffic(key, "bgr2lab_kernel", npix)),
            roofline_line("csd_kernel", stage_ms["csd"], 3, npix, profile=prof,
                          traffic=pmc_traffic(key, "csd_kernel", npix)),
            # colour layout and edge histogram share one pass over the image (imgan_cld_ehd): priced
            # with the 3 B/px the pass reads (SURVEY 8d prices each descriptor's read of the image, 6)
            roofline_line("cld_ehd_sums_kernel (+finalise, classify)", stage_ms["cld+ehd"], 3, npix, profile=prof,
                          traffic=pmc_traffic(key, "cld_ehd_sums_kernel", npix),
                          note="one pass for both descriptors: 3 B/px read (two passes at 3 B/px each before)"),
        ]
        path_bytes_px = 6 + 63 + desc_bytes + dcd_survey
        out = {
            "metric": "megapixels/sec (SLIC+RAG+MPEG-7) at 1/2/4/8 GPUs; % HBM roofline",
            "value": round(world * npix * args.steps / dt / 1e6, 2),
            "unit": "megapixels/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (integer keys; f32 Lloyd, f64 ties)", "data": "synthetic",
            "config": {"workload": "configs[1]: %d x %dx%d BGR per GPU, K=%d (n=%d, S=%d), "
                                   "num_iter=%d, cw=%d, %s" %
                                   (B, W, H, K, pipe.n, pipe.slic.S, num_iter, cw,
                                    "Lab + SLIC + CLD + EHD + CSD(256) + DCD(k=8 per superpixel, "
                                    "%.1f Lloyd rounds per pixel on chip)" % t_bar),
                       "global_batch": B * world, "parallelism": "batch-sharded x%d" % world,
                       "library": os.path.relpath(LIB_PATH, ROOT)},
            # the dominant kernel by time: one C call (imgan_dcd_segments_listed) = the pass that sorts
            # the pixels by segment + the three capacity tiers of dcd_dedup_kernel, one launch each
            "roofline": roofline_line(
                "dcd_dedup_kernel x3 tiers (+seg_scatter_kernel, seg_plan_kernel)",
                stage_ms["dcd_segments"], dcd_survey, npix,
                traffic=dcd_traffic(key, npix), profile=prof, bound="valu",
                note="largest share of the step time; timed around the C call, i.e. the sum of the five "
                     "kernels it launches back to back.  Algorithmic bytes = SURVEY 8d's streaming figure "
                     "8*T + 13*P B/px with T = %.1f measured Lloyd rounds per pixel, P = 1; the rounds run "
                     "in LDS, so the HBM traffic it must move is 15 B/px (4 labels in, 4 positions out, "
                     "4 positions + 3 colours in; traffic above that: partial lines of the sort pass and of the "
                     "colour gather) and the kernels are bound by the vector pipe (SQ_ACTIVE_INST_VALU x 4 / "
                     "SIMDs / cycles = 0.79 on the small tier, profiles/r03_pmc_B64.txt)" % t_bar),
            "roofline_kernels": others,
            "dcd_segments_at_round_cap": pipe.segments_at_round_cap(),
            "dcd_segments_left_to_streaming_rounds": int(segments_left),
            "path": {"bytes_per_px": round(path_bytes_px, 1),
                     "achieved": round(path_bytes_px * npix * args.steps / dt / 1e9 * 1.0, 1),
                     "unit": "GB/s", "frac": round(path_bytes_px * npix * args.steps / dt / 1e9
                                                   / HBM_PEAK_GBS, 4),
                     "note": "whole step against SURVEY 8d's algorithmic bytes "
                             "(6 Lab + 63 SLIC + %g CLD/EHD/CSD + DCD)" % desc_bytes},
        }
        if rccl is not None:
            out["rccl"] = rccl
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["cpu_baseline_numpy"] = cpu_baseline(H, W, K, cw)
        if world == 1 and not args.no_configs:
            del pipe, bgr, index, pending, image_rows, region_rows
            torch.cuda.empty_cache()
            out["configs"] = [run_config_bear(5, 2), run_config_rag(max(args.steps, 3), 1),
                              run_config_mosaic(2, 1)]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
