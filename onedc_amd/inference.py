"""Command-line harness with the reference's surface (src/inference.py:138-164).

    python -m onedc_amd.inference --config_path src/config_inference.yaml --checkpoint_path CKPT \
        --decoder_only --decoder_bin_path a.bin --decoder_out_path a.png [--vae_path vae.safetensors]

Same flags, same YAML keys (config_inference.yaml), CLI values override the file
(inference.py:149-152; OmegaConf is replaced by PyYAML + a dict merge), LOCAL_RANK override
(:154-156), checkpoint directory = model.safetensors + model_1.safetensors loaded strictly (:87-93).
Additions: --vae_path (the reference downloads the SD-2.1 VAE from the HF hub), and
--decoder_bin_path may be a directory: all *.bin files of one padded size are decoded as batches.
Without --decoder_only the harness runs the reference's evaluate loop (:109-134): every .jpg/.png of
--eval_image_path is encoded to <output_path>/stream/<name>.bin, decoded to <output_path>/recon/<name>.png,
and the bpp dictionaries land in detail_bpp_caption.xlsx / summary_bpp.xlsx (CSV content, like the
reference, which calls to_csv on those names).

Several processes (one per GPU: `torchrun --nproc-per-node N -m onedc_amd.inference ...`, or any launcher that sets RANK /
WORLD_SIZE / LOCAL_RANK): the reference shards the pictures with `accelerator.prepare(img_dataloader)`, removes the output
directory on the main process only and meets at a barrier per picture (inference.py:37,49-59,82,127).  Here rank r takes
pictures r, r + world, ... (dataparallel.shard), rank 0 alone prepares the output directory and the others wait for it, every
rank appends its bpp rows to a part file as it goes, and rank 0 merges the parts into the two tables (the reference lets every
process overwrite them with its own shard).  The rendezvous is through files in --output_path: no process group is needed.
The noise of latent_dist.sample() is drawn per picture from a generator seeded with the picture's index, so a picture's
stream does not depend on how many processes shared the folder.  A directory given to --decoder_only goes through the
overlapped pipeline (`model.pipeline().map(..., out="u8", to_host=True)`), sharded the same way.
"""
import argparse
import glob
import os

import numpy as np
import torch
import yaml


def save_image(img, path):
    """torchvision.utils.save_image for one [3,H,W] tensor in [0,1] (x255 + 0.5, clamp, uint8)."""
    from PIL import Image
    if not bool(torch.isfinite(img).all()):
        # the only way a decode produces them: an activation left the IEEE-half range (act_dtype "f16", +-65504 after
        # the VAE's 2^-6 stream scale) -- decode this picture with act_dtype: bf16 in the config instead
        raise FloatingPointError(f"{path}: the decoded picture has non-finite values (half-precision range exceeded); "
                                 "set act_dtype: bf16 in the config")
    arr = img.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def save_image_u8(arr, path):
    """One [H,W,3] uint8 picture as the device's output kernel quantised it (model.decode(out="u8"): the same
    x255 + 0.5, clamp, truncation as `save_image`, without moving f32 samples over PCIe)."""
    from PIL import Image
    Image.fromarray(arr.cpu().numpy()).save(path)


def prepare_output_dir(out):
    """The reference removes an existing --output_path before writing (inference.py:41-45).  Same here, but only for a
    directory that looks like a previous run of this harness (it has recon/) or is empty: a mistyped path must not cost
    somebody a directory tree."""
    import shutil
    if os.path.exists(out):
        if not os.path.isdir(out):
            raise SystemExit(f"--output_path {out!r} exists and is not a directory")
        if os.listdir(out) and not os.path.isdir(os.path.join(out, "recon")):
            raise SystemExit(f"--output_path {out!r} exists, is not empty and has no recon/ of an earlier run: "
                             "refusing to delete it")
        shutil.rmtree(out)
    recon_path, stream_path = os.path.join(out, "recon"), os.path.join(out, "stream")
    os.makedirs(recon_path)
    os.makedirs(stream_path)
    return recon_path, stream_path


def rank_world():
    """(rank, world) of this process from the launcher's environment (torchrun / accelerate launch); (0, 1) alone."""
    return int(os.environ.get("RANK", "0") or 0), max(1, int(os.environ.get("WORLD_SIZE", "1") or 1))


def run_token():
    """What the ranks of ONE launch attempt share and another launch (or a restarted attempt of this one) does not.
    torchrun / accelerate give every rank of every node the same MASTER_ADDR:MASTER_PORT and TORCHELASTIC_RUN_ID, and count
    restarts in TORCHELASTIC_RESTART_COUNT (a restarted attempt must not accept the failed attempt's marker / done files).
    That alone is NOT unique per launch: plain `torchrun --nproc_per_node=N` and `accelerate launch` export the same
    `127.0.0.1:29500-none-restart0` every time, so a re-run after a failed or killed run would read the dead run's marker and
    .failed<r> files as its own.  On one node (LOCAL_WORLD_SIZE == WORLD_SIZE) the launcher's pid -- the ranks' common parent --
    is appended; a launch over several nodes has no such common process and must carry a real run id (`torchrun --rdzv-id`,
    or ONEDC_RUN_ID: the same on every node, new for every launch).  No launcher variables at all: the parent's pid."""
    env = os.environ
    if env.get("ONEDC_RUN_ID"):
        return env["ONEDC_RUN_ID"]
    rid = env.get("TORCHELASTIC_RUN_ID", "")
    if env.get("MASTER_ADDR") and env.get("MASTER_PORT"):
        token = f"{env['MASTER_ADDR']}:{env['MASTER_PORT']}-{rid}-restart{env.get('TORCHELASTIC_RESTART_COUNT', '0')}"
        world = max(1, int(env.get("WORLD_SIZE", "1") or 1))
        local_world = int(env.get("LOCAL_WORLD_SIZE", "0") or 0) or world
        if local_world >= world:
            return f"{token}-ppid{os.getppid()}"
        if rid in ("", "none"):
            raise RuntimeError("a launch over several nodes needs a run id that is new for every launch: torchrun --rdzv-id <id>, "
                               "or ONEDC_RUN_ID=<id> on every node (MASTER_ADDR:MASTER_PORT repeat from launch to launch)")
        return token
    return f"{rid}-ppid{os.getppid()}"


class RankFailed(RuntimeError):
    """Another rank of this launch left a .failed<r> file: it raised, and waiting for it would only time out."""


def wait_for(predicate, what, timeout=1800.0, poll=0.05, abort=None):
    """File-system rendezvous: poll until predicate() is true; abort() (optional) is polled beside it and returns a
    message when the wait has become pointless (a rank of this launch failed)."""
    import time
    t0 = time.monotonic()
    while not predicate():
        why = abort() if abort else None
        if why:
            raise RankFailed(f"rank {rank_world()[0]}: stopped waiting for {what}: {why}")
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank_world()[0]}: waited {timeout:.0f} s for {what}")
        time.sleep(poll)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def check_mode_args(parser, cli):
    """argparse leaves every path optional like the reference does (inference.py:138-147); what each mode needs is
    checked here instead of failing later inside os.path.join(None, ...)."""
    if cli["decoder_only"]:
        for k in ("decoder_bin_path", "decoder_out_path"):
            if not cli[k]:
                parser.error(f"--decoder_only needs --{k}")
    else:
        for k in ("output_path", "eval_image_path"):
            if not cli[k]:
                parser.error(f"--{k} is required unless --decoder_only is given")
        if not os.path.isdir(cli["eval_image_path"]):
            parser.error(f"--eval_image_path {cli['eval_image_path']!r} is not a directory")


def load_image(path):
    """data/evalset.py:15-20: RGB, ToTensor() * 2 - 1 -> f32 [3,H,W] in [-1,1]."""
    from PIL import Image
    arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous() * 2 - 1


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", type=str, required=True)
    parser.add_argument("--checkpoint_path", type=str, required=True)
    parser.add_argument("--output_path", type=str, required=False)
    parser.add_argument("--eval_image_path", type=str, required=False)
    parser.add_argument("--decoder_only", action="store_true")
    parser.add_argument("--decoder_bin_path", type=str, default="", required=False)
    parser.add_argument("--decoder_out_path", type=str, default="", required=False)
    parser.add_argument("--vae_path", type=str, default=None, required=False)
    parser.add_argument("--batch_size", type=int, default=16, required=False)
    cli = vars(parser.parse_args(argv))
    check_mode_args(parser, cli)
    with open(cli["config_path"]) as f:
        config = yaml.safe_load(f) or {}
    config.update(cli)                                   # CLI over file, like OmegaConf.merge(config, args)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != config.get("local_rank"):
        config["local_rank"] = env_local_rank
    config["unet_ckpt"] = None
    return config


class Evaluator:
    def __init__(self, args, model=None):
        """model: an object with the wrapper's surface (tests); default: built and loaded like inference.py:69-74."""
        self.args = args
        self.rank, self.world = rank_world()
        if model is None:
            from .model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
            rank = args.get("local_rank", -1)
            device = torch.device("cuda", rank if rank is not None and rank >= 0 else 0)
            model = SD15_1step_codec_stage1(args, device=device)
            model.load(args["checkpoint_path"], args.get("vae_path"))
            model.prepare()
            model.codec_model.update(force=True)
            model.codec_model.debug = False
        self.model = model

    @torch.no_grad()
    def evaluate(self):
        args = self.args
        self.model.eval()
        if not args["decoder_only"]:
            return self.evaluate_folder()
        src, dst = args["decoder_bin_path"], args["decoder_out_path"]
        if os.path.isdir(src):
            return self.decode_folder(src, dst)
        # clamp(-1,1)*0.5+0.5 and save_image's 8-bit quantisation (inference.py:106-107) happen in the output kernel;
        # a picture with NaN / Inf raises kernels.HalfRangeError there instead of being written
        recon = self.model.decode(fp=src, out="u8")
        save_image_u8(recon[0], dst)

    @torch.no_grad()
    def decode_folder(self, src, dst):
        return decode_folder_pipelined(self.model, src, dst, self.args["batch_size"], self.rank, self.world)

    @torch.no_grad()
    def evaluate_folder(self):
        """The encode + decode loop of the reference (inference.py:109-134), one picture at a time; this rank's share of the
        folder when several processes run it (module docstring)."""
        out = self.args["output_path"]
        recon_path, stream_path = os.path.join(out, "recon"), os.path.join(out, "stream")

        def one_picture(idx, name, path):
            image = load_image(path)[None].to(self.model.device)
            noise = picture_noise(idx, image.shape[2], image.shape[3], self.model.device)
            stream_file = os.path.join(stream_path, f"{name}.bin")
            _, bpp = self.model.encode(image, fp=stream_file, noise=noise)
            recon = self.model.decode(fp=stream_file, out="u8")
            save_image_u8(recon[0], os.path.join(recon_path, f"{name}.png"))
            bpp = dict(bpp)
            bpp["image_name"] = name
            return bpp

        return evaluate_sharded(out, self.args["eval_image_path"], self.rank, self.world, one_picture,
                                lambda frame, path: frame.to_csv(path))


def picture_noise(idx, H, W, device):
    """The draw of latent_dist.sample() for picture `idx` of the sorted folder.  inference.py:28 seeds the global generator
    once and lets the pictures draw from it one after the other; here the draw belongs to the PICTURE (a generator seeded
    with its index): the same stream whichever rank encodes it, however many ranks share the folder."""
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    gen = torch.Generator(device=device).manual_seed(idx)
    return torch.randn(1, Hp // 8, Wp // 8, 4, device=device, generator=gen)


def evaluate_sharded(out, src, rank, world, one_picture, write_table):
    """The folder loop shared by the two harnesses.  one_picture(idx, name, path) -> row dict (it writes the picture's
    files itself).  Rank 0 prepares `out` (main process only: inference.py:49-54) and the others wait for its marker file;
    rank r takes pictures r, r + world, ... of the sorted folder and appends every row to its part file at once (a failure
    at picture k keeps the rows of the pictures before it); rank 0 merges the parts into detail_bpp_caption.xlsx /
    summary_bpp.xlsx when every rank has left its done file."""
    import pandas as pd
    from . import dataparallel
    marker, token = os.path.join(out, ".onedc_run"), run_token()
    part_file = lambda r: os.path.join(out, f".detail_part{r}.csv")      # noqa: E731
    done_file = lambda r: os.path.join(out, f".done{r}")                 # noqa: E731
    failed_file = lambda r: os.path.join(out, f".failed{r}")             # noqa: E731

    def someone_failed():
        for r in range(world):
            if r != rank and _read(failed_file(r)) == token:
                return f"rank {r} failed (see its traceback)"
        return None
    if rank == 0:
        # whatever an earlier attempt left behind goes FIRST (a stale marker would let rank 1 start while this rank is still
        # deleting the directory; stale done / part files would be merged): the marker, then the rest with the directory
        if os.path.exists(marker):
            os.remove(marker)
        try:
            prepare_output_dir(out)
        except BaseException:                    # (a refused directory: tell the waiting ranks, if there is a place to say it)
            if os.path.isdir(out):
                with open(failed_file(0), "w") as f:
                    f.write(token)
            raise
        with open(marker, "w") as f:
            f.write(token)
    else:
        wait_for(lambda: _read(marker) == token, f"rank 0 to prepare {out!r}", abort=someone_failed)
    try:
        names = sorted(f for f in os.listdir(src) if f.endswith((".jpg", ".png")))
        for idx in dataparallel.shard(len(names), rank, world):
            name = os.path.splitext(names[idx])[0]
            row = pd.DataFrame(one_picture(idx, name, os.path.join(src, names[idx])), index=[idx])
            row.to_csv(part_file(rank), mode="a", header=not os.path.exists(part_file(rank)))
    except BaseException:
        try:                                     # the other ranks (rank 0's merge above all) stop waiting at once
            with open(failed_file(rank), "w") as f:
                f.write(token)
        except OSError:
            pass
        raise
    with open(done_file(rank), "w") as f:
        f.write(token)
    if rank != 0:
        return None
    wait_for(lambda: all(_read(done_file(r)) == token for r in range(world)), "the other ranks to finish", abort=someone_failed)
    parts = [pd.read_csv(part_file(r), index_col=0) for r in range(world) if os.path.exists(part_file(r))]
    detail = pd.concat(parts).sort_index() if parts else pd.DataFrame()
    write_table(detail, os.path.join(out, "detail_bpp_caption.xlsx"))
    avg = detail.mean(numeric_only=True).to_dict() if len(detail) else {}
    write_table(pd.DataFrame([avg]), os.path.join(out, "summary_bpp.xlsx"))
    for r in range(world):
        for f in (part_file(r), done_file(r)):
            if os.path.exists(f):
                os.remove(f)
    os.remove(marker)
    print(avg)
    return avg


def decode_folder_pipelined(model, src, dst, batch_size, rank=0, world=1, z_only=False, group=8):
    """Every *.bin of a directory -> <dst>/<name>.png in the throughput mode: batches of one padded size through the
    overlapped pipeline (`model.pipeline().map(..., out="u8", to_host=True)`: 8-bit pictures in page-locked host memory),
    `group` batches per map() call (bounds the host memory held), PNG writing on a small thread pool beside the next call.
    Batches are dealt to the ranks round-robin.  Same pictures as sequential `model.decode` calls, bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    from . import dataparallel
    from .entropy.utils import decode_i, read_from_file
    os.makedirs(dst, exist_ok=True)
    by_size = {}
    for f in sorted(glob.glob(os.path.join(src, "*.bin"))):
        d = decode_i(read_from_file(f), 14, 64)
        by_size.setdefault((d["pad_height"], d["pad_width"], d["height"], d["width"]), []).append(f)
    batches = [g[i:i + batch_size] for g in by_size.values() for i in range(0, len(g), batch_size)]
    mine = [batches[i] for i in dataparallel.shard(len(batches), rank, world)]
    pipe = model.pipeline(z_only=z_only)
    written = []
    try:
        with ThreadPoolExecutor(max_workers=4) as pool:
            pending = []
            for i in range(0, len(mine), group):
                chunk = mine[i:i + group]
                outs = pipe.map([[read_from_file(f) for f in b] for b in chunk], out="u8", to_host=True)
                for b, imgs in zip(chunk, outs):
                    for f, img in zip(b, imgs):
                        path = os.path.join(dst, os.path.splitext(os.path.basename(f))[0] + ".png")
                        pending.append(pool.submit(save_image_u8, img, path))
                        written.append(path)
            for fut in pending:
                fut.result()
    finally:
        pipe.close()
    return written


def main(argv=None):
    return Evaluator(parse_args(argv)).evaluate()


if __name__ == "__main__":
    main()
