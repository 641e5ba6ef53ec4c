#!/usr/bin/env python3
"""bench.py -- HVQM4 picture-reconstruction throughput on MI355X.

Metric (BASELINE.json): decoded Mpixels/s (bit-exact YUV) and % of the HBM roofline.
Workload at every N (weak scaling, one rank per GPU, no collective on the data path): the per-GPU
share of SURVEY.md's config C5 -- 128 concurrent 640x480 HVQM4 1.5 streams, GOP I P B B P B B ...
(16 pictures), descriptors pre-parsed and resident in HBM.  One "step" = every stream decodes one
full GOP (128 * 16 = 2048 pictures) through the batched path (hvq_replay): the launches of the
dependency levels, nothing skipped.  `value` = luma pixels decoded by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      algorithmic bytes (1.5 B/px written + 1.5 B/px read for P/B)
                per launch / average launch duration from HIP events on the launch stream, vs 8 TB/s
  cpu_baseline  the reference decoder (oracle/_ref, kind "reference") or this repo's scalar
                restatement (kind "port") timed on one host core over a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=128, help="concurrent streams per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--gop", default="IPBBPBBPBBPBBPBB")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic clips per GPU (replicated over the streams)")
    ap.add_argument("--preset", default="dense", choices=["dense", "realistic", "flat", "natural"])
    ap.add_argument("--nslots", type=int, default=6)
    ap.add_argument("--no-gpu-parse", action="store_true", help="skip the GPU-entropy-parse end-to-end leg")
    ap.add_argument("--no-sdk", action="store_true", help="skip the SDK-boundary (PCIe-inclusive) leg")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--parse-threads", type=int, default=0, help="host parse threads for the end-to-end pass (0 = all cores, max 64)")
    ap.add_argument("--mv-bits", default="0,1,2", help="vector residual-bit choices of the synthetic P/B pictures (reach = 16 << bits samples)")
    args = ap.parse_args()

    from hvqm4_amd.distrib import Group
    grp = Group()                       # torch.distributed (nccl = RCCL) only when WORLD_SIZE > 1
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)

    import numpy as np
    from hvqm4_amd import batch
    from hvqm4_amd.container import video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip

    # ---- synthetic inputs (fixed seeds; clip i of rank r has seed 1000 + r*distinct + i) ----
    t0 = time.time()
    clips = [make_clip(SynthConfig(width=args.width, height=args.height, version="1.5", gop=args.gop,
                                   seed=1000 + rank * args.distinct + i, preset=args.preset,
                                   mv_res_bits=tuple(int(x) for x in args.mv_bits.split(","))))
             for i in range(args.distinct)]
    gen_s = time.time() - t0
    pics = [list(video_pictures(c.data)) for c in clips]
    n_pic = len(args.gop)

    # one rank per GPU; HVQM4_BENCH_SHARE_GPU=1 lets several ranks share device 0 (rehearsal on a 1-GPU box only)
    ctx = batch.Context(0 if os.environ.get("HVQM4_BENCH_SHARE_GPU") else local_rank)
    sids = []
    for s in range(args.streams):
        sids.append(ctx.open_stream(args.width, args.height, 2, 2, True, args.nslots))
    # decode-order interleave: picture k of every stream, then k+1 ... (the order a player would submit).
    # This first pass is also the END-TO-END measurement: host entropy parse (thread pool) + descriptor upload
    # + all launches, from bitstreams in host memory to pictures in HBM.
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(args.parse_threads or allowed, 64))
    a_sid, a_ft, a_pic = [], [], []
    for k in range(n_pic):
        for s, sid in enumerate(sids):
            ft, _d, pic = pics[s % args.distinct][k]
            a_sid.append(sid); a_ft.append(ft); a_pic.append(pic)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.submit_many(a_sid, a_ft, a_pic, threads)
    t_parse = time.perf_counter() - t0
    ctx.flush()
    ctx.sync()
    t_e2e = time.perf_counter() - t0
    st = ctx.stats()

    # ---- parity spot-check against the CPU oracle on what is still resident ----
    verified = None
    if not args.no_verify:
        from oracle import bridge
        verified = 0
        for i in range(min(args.distinct, args.streams)):
            want = bridge.oracle_decode(clips[i].data, n_pic)
            for k in range(n_pic):
                try:
                    got = ctx.read_picture(sids[i], k)
                except Exception:
                    continue        # slot already reused by a later picture
                if not np.array_equal(got, want[k]):
                    raise SystemExit(f"PARITY FAILURE: stream {i} picture {k} differs from the oracle")
                verified += 1

    def barrier():
        ctx.sync()
        grp.barrier()

    for _ in range(args.warmup):
        ctx.replay(1)
    barrier()
    t0 = time.perf_counter()
    gpu_ms = ctx.replay(args.steps)          # K steps, timed by HIP events on the launch stream
    barrier()
    wall = grp.max(time.perf_counter() - t0)

    # ---- end to end with the entropy parse ON THE GPU: raw bitstreams in host memory ->
    # H2D -> parse kernel -> reconstruction launches.  Second pass of a fresh context = steady state (buffers sized).
    gpu_e2e = None
    if rank == 0 and world == 1 and not args.no_gpu_parse:
        ctx2 = batch.Context(local_rank)
        sids2 = [ctx2.open_stream(args.width, args.height, 2, 2, True, args.nslots) for _ in range(args.streams)]
        a_sid2 = [sids2[sids.index(x)] for x in a_sid]
        a_raw = [bytes(p) for p in a_pic]
        t_pass, parse_ms = [], []
        for _ in range(3):
            ctx2.sync()
            t0 = time.perf_counter()
            ctx2.submit_many_device(a_sid2, a_ft, a_raw)
            t1 = time.perf_counter()
            ctx2.flush()
            t2 = time.perf_counter()
            ctx2.sync()
            t_pass.append(time.perf_counter() - t0)
            t_split = [round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((time.perf_counter() - t2) * 1e3, 2)]
            parse_ms.append(ctx2.stats().gpu_parse_ms)
        # streaming: batch n+1 is submitted (copied into the second pinned arena, uploaded on the copy stream) while
        # batch n is parsed.  Steady-state period = time between the completions of consecutive hvq_flush_end calls,
        # after two warm-up batches (the second arena and its staging are allocated on first use).
        nwarm, nbatch = 2, 8
        ctx2.sync()
        ctx2.submit_many_device(a_sid2, a_ft, a_raw)
        ctx2.flush_begin()
        t_calls = [0.0, 0.0, 0.0]
        t_end = []
        for k in range(nwarm + nbatch - 1):
            ta = time.perf_counter()
            ctx2.submit_many_device(a_sid2, a_ft, a_raw)
            tb = time.perf_counter()
            ctx2.flush_end()
            tc = time.perf_counter()
            ctx2.flush_begin()
            td = time.perf_counter()
            t_end.append(tc)
            if k >= nwarm:
                t_calls[0] += tb - ta; t_calls[1] += tc - tb; t_calls[2] += td - tc
        ctx2.flush_end()
        t_end.append(time.perf_counter())
        ctx2.sync()
        t_pipe = (t_end[-1] - t_end[nwarm]) / (len(t_end) - 1 - nwarm)
        t_calls = [round(x / (nbatch - 1) * 1e3, 2) for x in t_calls]
        parse_ms_streaming = ctx2.stats().gpu_parse_ms
        nbatch = nwarm + nbatch
        n_done = 3 + nbatch
        ok = 0
        for i in range(min(4, args.streams)):
            for k in range(n_pic):
                try:
                    a = ctx.read_picture(sids[i], k); b = ctx2.read_picture(sids2[i], k + (n_done - 1) * n_pic)
                except Exception:
                    continue
                if not np.array_equal(a, b):
                    raise SystemExit(f"PARITY FAILURE: GPU-parsed stream {i} picture {k} differs from the host-parsed one")
                ok += 1
        gpu_e2e = {"value": round(int(st.luma_pixels) / min(t_pass[1:]) / 1e6, 1), "unit": "Mpixels/s",
                   "streaming_value": round(int(st.luma_pixels) / t_pipe / 1e6, 1), "streaming_ms_per_batch": round(t_pipe * 1e3, 2), "streaming_submit_end_begin_ms": t_calls, "streaming_parse_kernel_ms": round(parse_ms_streaming, 3),
                   "parse_kernel_ms": round(min(parse_ms[1:]), 3), "pass_ms": [round(t * 1e3, 2) for t in t_pass], "submit_flush_sync_ms": t_split,
                   "host_copy_threads": 4, "pictures_checked_against_host_parsed": ok,
                   "what": "raw bitstreams in host memory -> H2D -> entropy parse kernel (one workgroup per picture) -> "
                           "reconstruction launches -> pictures in HBM; no host entropy parse.  value: one batch start to finish; "
                           "streaming_value: steady-state period of hvq_flush_begin / submit next / hvq_flush_end (next batch "
                           "copied and uploaded while this one is parsed), 8 batches after 2 warm-up batches"}
        ctx2.close()

    px_step = int(st.luma_pixels)
    value = px_step * args.steps * world / wall / 1e6
    launches = int(st.launches)
    avg_launch_s = gpu_ms * 1e-3 / (args.steps * launches)
    achieved = st.algorithmic_bytes / launches / avg_launch_s / 1e9

    out = {
        "metric": "decoded Mpixels/s (bit-exact YUV)",
        "value": round(value, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"C5 share: {args.streams} concurrent {args.width}x{args.height} HVQM4 1.5 streams per GPU, "
                        f"GOP {args.gop}, {args.preset} synthetic streams, descriptors resident in HBM",
            "streams_per_gpu": args.streams, "pictures_per_step": int(st.pictures),
            "distinct_clips_per_gpu": args.distinct, "launches_per_step": launches,
            "workgroups_per_step": int(st.workgroups), "nslots": args.nslots, "sharding": "one clip per stream, streams split across GPUs, no collective",
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args),
            "kernel": "hvq_recon_kernel", "algorithmic_bytes_per_launch": int(st.algorithmic_bytes // launches),
            "avg_launch_us": round(avg_launch_s * 1e6, 2),
            "descriptor_bytes_per_launch": int(st.descriptor_bytes // launches),
        },
        "gpu_event_ms_per_step": round(gpu_ms / args.steps, 4),
        "end_to_end_gpu_parse": gpu_e2e,
        "end_to_end": {"value": round(px_step / t_e2e / 1e6, 1), "unit": "Mpixels/s", "parse_threads": threads,
                       "parse_only_mpix_s": round(px_step / t_parse / 1e6, 1),
                       "what": "first pass: host entropy parse + descriptor H2D + kernels, bitstreams in host memory -> pictures in HBM"},
        "verified_pictures": verified,
        "flags_or": int(st.flags_or),
    }

    # display epilogue: YUV420 -> RGB24 of the newest picture of every stream, one launch
    try:
        ctx.rgb_bench(2)
        rms, rbytes, rpics = ctx.rgb_bench(50)
        out["rgb_epilogue"] = {"kernel": "hvq_yuv420_rgb_kernel", "pictures_per_launch": rpics,
                               "bytes_per_launch": rbytes, "avg_launch_us": round(rms * 1e3 / 50, 2),
                               "achieved_GBs": round(rbytes * 50 / (rms * 1e-3) / 1e9, 1),
                               "frac_of_8TBs": round(rbytes * 50 / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    except Exception as e:      # never let the optional epilogue measurement break the headline number
        out["rgb_epilogue"] = {"error": str(e)}

    # the SDK boundary itself (PCIe-inclusive, never `value`): one stream, one synchronous call per picture -- upload
    # of the caller's reference pictures, host parse, launch, download of the decoded picture
    if rank == 0 and world == 1 and not args.no_sdk:
        try:
            from hvqm4_amd import sdk
            pl = sdk.Player(args.width, args.height, 2, 2, True)
            seq = pics[0]
            for ft, _d, pic in seq:                                  # warm-up pass
                pl.decode(ft, bytes(pic))
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                for ft, _d, pic in seq:
                    pl.decode(ft, bytes(pic))
            dt = (time.perf_counter() - t0) / (reps * len(seq))
            pl.close()
            out["sdk_path"] = {"value": round(args.width * args.height / dt / 1e6, 1), "unit": "Mpixels/s",
                               "ms_per_picture": round(dt * 1e3, 3),
                               "what": "HVQM4DecodeIpic/Ppic/Bpic through the C ABI with host picture buffers, one "
                                       "synchronous picture at a time (PCIe both ways inside the call)"}
        except Exception as e:
            out["sdk_path"] = {"error": str(e)}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(clips[0], args.cpu_seconds)
    if rank == 0:
        grp.emit(json.dumps(out))
    ctx.close()
    grp.close()


def pmc_traffic(args):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 --pmc passes of this same
    default workload (tools/pmc_passes.sh -> tools/pmc_traffic.py -> profiles/*_pmc_traffic.json); None for
    any other workload."""
    default = (args.streams, args.width, args.height, args.gop, args.preset, args.distinct, args.mv_bits) == \
              (128, 640, 480, "IPBBPBBPBBPBBPBB", "dense", 8, "0,1,2")
    if not default:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    return int(json.load(open(files[-1]))["hbm_bytes_per_launch"])


def cpu_baseline(clip, budget_s: float):
    """One host core, decode calls only, same clip as stream 0 of the GPU run."""
    from oracle import bridge
    if bridge.have_ref():
        kind, timer = "reference", bridge.ref_time
    else:
        kind, timer = "port", bridge.oracle_time
    t, px = timer(clip.data, 1)
    reps = max(1, int(budget_s / max(t, 1e-6)))
    t, px = timer(clip.data, reps)
    return {"value": round(px / t / 1e6, 1), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"{reps} passes over one {clip.width}x{clip.height} {len(clip.kinds)}-picture clip "
                      f"(decode calls only, {t:.1f} s)"}


if __name__ == "__main__":
    main()
