#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: 768x768 images/sec decoded.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one full decode of a batch of 16 synthetic 768x768 bitstreams per GPU
(BASELINE configs[1]: "768x768 batch=16 decode, lambda=4.6, bf16, 1xMI355X"): container parse,
FSQ index unpack, hyperprior synthesis, the four host rANS hops of the four-part prior, semantic
adaptor + codec synthesis, one SD1.5-UNet step, x0, SD-VAE decoder, crop.  Weights are
deterministic random-init tensors of the real architecture (no checkpoints offline), bitstreams
are produced beforehand by the product's own encode side of the prior.  N > 1: image-batch data
parallel, one process per GPU, weak scaling, one RCCL broadcast of the weights, no data-path
collective.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      for the dominant kernel (the implicit-GEMM instantiation with the most device time): achieved =
                algorithmic FLOPs of its launches / their summed duration, measured live with HIP
                events on the launch stream during the timed region; peak = dense bf16 MFMA.
  cpu_baseline  the fp32 oracle (oracle/torch_ref.py, "port") timed on the host cores of this box
                on a bounded sample, scaled to the metric's unit by algorithmic FLOPs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (guide: MI355X_MICROARCH.md)
GFLOP_PER_IMAGE = {768: 8046.7, 512: 3419.5, 256: 831.4}   # SURVEY.md 8(d), algorithmic, 2*MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="no pipelining across batches")
    ap.add_argument("--s-streams", type=int, default=2, help="concurrent synthesis/UNet/VAE stages in the pipeline")
    ap.add_argument("--cpu-sample", type=int, default=768, help="picture size of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads of the CPU baseline (16 was the fastest of 16/32/64 on the 2x64-core host)")
    return ap.parse_args()


def cpu_baseline(sample, threads, codec_sd, unet_sd, vae_sd):
    """Oracle decode (everything after the entropy decode) of one sample x sample picture on the
    host; scaled to 768^2-equivalent images/s by algorithmic FLOPs when sample != 768."""
    from oracle import torch_ref as R
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    unet_folded = R.fold_lora({k: v.cpu() for k, v in unet_sd.items()})
    codec_cpu = {k: v.cpu() for k, v in codec_sd.items()}
    vae_cpu = {k: v.cpu() for k, v in vae_sd.items()}
    g = torch.Generator().manual_seed(0)
    y_hat = torch.randn(1, 128, sample // 16, sample // 16, generator=g)
    z_sem = torch.randn(1, 128, sample // 64, sample // 64, generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        R.decode_from_latents(codec_cpu, unet_folded, vae_cpu, y_hat, z_sem)
        dt = time.perf_counter() - t0
    scale = GFLOP_PER_IMAGE[sample] / GFLOP_PER_IMAGE[768]
    return {"value": round(scale / dt, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_ref.py fp32 decode_from_latents of one {sample}x{sample} picture in "
                      f"{dt:.2f} s on {cores} of {os.cpu_count()} host threads, scaled to 768x768 by algorithmic "
                      f"GFLOP ({GFLOP_PER_IMAGE[sample]}/{GFLOP_PER_IMAGE[768]}); host rANS not included"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from onedc_amd import arch, synth
    from onedc_amd.dataparallel import broadcast_state_dicts, max_over_ranks
    from onedc_amd import kernels as K
    from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

    specs = [arch.unet_spec(True), arch.codec_decode_spec(), arch.vae_decoder_spec()]
    if rank == 0:
        sds = [synth.unet_state_dict(0), synth.codec_state_dict(0), synth.vae_state_dict(0)]
    else:
        sds = [None, None, None]
    if world > 1:
        sds_dev = broadcast_state_dicts(sds, specs, device, rank, world)
    else:
        sds_dev = sds
    model = SD15_1step_codec_stage1({"vae_attn_patch": 16}, device=device).load_state_dicts(*sds_dev)
    model.codec_model.update(force=True)

    # ---- synthetic bitstreams of the right shape (made by the product's own encoder side) ----------
    B, S = args.batch, args.size
    zh = S // 64
    rng = np.random.default_rng(1000 + rank)
    idx = torch.from_numpy(rng.integers(0, 16384, (B, zh, zh)).astype(np.int32)).to(device)
    blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=2 + rank)
    stream_bytes = sum(len(b) for b in blobs)
    del sds_dev
    torch.cuda.empty_cache()

    # One step = one whole decode of one batch.  Default: the two-stage pipeline (the rANS hops of
    # batch n+1 overlap the UNet / VAE of batch n; all K batches start and finish inside the timed
    # region).  --sequential times K back-to-back model.decode() calls instead.
    pipe = None if args.sequential else model.pipeline(s_streams=args.s_streams)

    def run(k):
        if pipe is None:
            return [model.decode(stream=blobs) for _ in range(k)][-1]
        return pipe.map([blobs] * k)[-1]

    if args.warmup:
        run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    K.PROFILE = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof, K.PROFILE = K.PROFILE, None
    assert out.shape == (B, 3, S, S) and bool(torch.isfinite(out).all())

    elapsed = max_over_ranks(elapsed, device, world)

    # The timed region runs several batches at once (the pipeline's streams): a kernel's event-bracketed duration
    # there includes the time it shares the GPU with kernels of other batches.  One more, untimed, sequential decode
    # gives the same per-launch figures with the GPU to itself ("isolated" in the roofline object).
    iso = None
    if rank == 0 and pipe is not None:
        K.PROFILE = []
        model.decode(stream=blobs)
        torch.cuda.synchronize()
        iso, K.PROFILE = K.PROFILE, None

    if rank == 0:
        by_tile = {}
        for tile, flops, e0, e1, nbytes in prof:
            d = by_tile.setdefault(tile, [0.0, 0.0, 0, 0.0])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
            d[3] += nbytes
        dom = max(by_tile, key=lambda k: by_tile[k][1])
        flops, secs, launches, dom_bytes = by_tile[dom]
        # HBM traffic of the dominant kernel: measured offline with rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in
        # separate passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md) and committed under
        # profiles/; PMC passes serialise every dispatch, so they cannot run inside the timed region.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get({"256x256": "conv_gemm256_kernel", "128x256": "conv_gemm128x256_kernel",
                                            "win3x3": "conv_win_kernel", "ups2x2": "conv_ups_kernel"}.get(dom, dom))
            if t:
                traffic = t["hbm_bytes_per_launch"]
        all_flops = sum(v[0] for v in by_tile.values())
        all_secs = sum(v[1] for v in by_tile.values())
        achieved = flops / secs / 1e12
        images = B * world * args.steps
        value = images / elapsed
        line = {
            "metric": "768x768 images/sec decode (whole node)", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{S}x{S} batch={B}/GPU decode, lambda=4.6-shaped model (random-init weights "
                                   f"of the real architecture), bf16 MFMA / fp32 accumulate, {world}xMI355X",
                       "images_per_gpu_per_step": B, "picture": f"{S}x{S}", "parallelism": f"dp{world}",
                       "stream_bytes_per_image": stream_bytes // B, "host_rans_threads": model.codec_model.host_threads,
                       "timed": "container parse + host rANS + H2D/D2H + all device kernels + crop",
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated(device) / 1e9, 1),
                       "schedule": "sequential decode() calls" if args.sequential else
                                   f"pipeline over batches: entropy stage of batch n+1 on its own stream || synthesis+UNet+VAE "
                                   f"of {args.s_streams} batches on {args.s_streams} streams"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(dom_bytes / launches),
                         "kernel": ("conv_gemm256_kernel" if dom == "256x256" else
                                    "conv_gemm128x256_kernel" if dom == "128x256" else
                                    "conv_win_kernel" if dom == "win3x3" else
                                    f"conv_gemm_kernel<Tile<{dom.replace('f32_', '').replace('x', ',')}>>"
                                    + (" f32" if dom.startswith("f32") else "")),
                         "by_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / args.steps * 1e3, 2),
                                           "launches_per_step": v[2] // args.steps} for k, v in sorted(by_tile.items())},
                         "launches_per_step": launches // args.steps,
                         "avg_launch_us": round(secs / launches * 1e6, 2),
                         "algorithmic_gflop_per_step": round(flops / args.steps / 1e9, 1),
                         "all_conv_gemm_tflops": round(all_flops / all_secs / 1e12, 2),
                         "conv_gemm_share_of_step": round(all_secs / elapsed, 3),
                         "whole_path_tflops": round(value / world * GFLOP_PER_IMAGE.get(S, 0) / 1e3, 2)},
        }
        if iso:
            it = [(f, e0.elapsed_time(e1) * 1e-3) for tile, f, e0, e1, _ in iso if tile == dom]
            if it:
                f_iso, s_iso = sum(f for f, _ in it), sum(t for _, t in it)
                line["roofline"]["isolated"] = {
                    "achieved": round(f_iso / s_iso / 1e12, 2), "frac": round(f_iso / s_iso / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "avg_launch_us": round(s_iso / len(it) * 1e6, 2), "launches": len(it),
                    "how": "same kernel, one untimed sequential decode after the timed region (no other batch on the GPU)"}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_threads, sds[1], sds[0], sds[2])
        print(json.dumps(line), flush=True)
    if pipe is not None:
        pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
