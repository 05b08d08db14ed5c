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
    def __init__(self, args):
        from .model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
        self.args = args
        rank = args.get("local_rank", -1)
        device = torch.device("cuda", rank if rank is not None and rank >= 0 else 0)
        self.model = SD15_1step_codec_stage1(args, device=device)
        self.model.load(args["checkpoint_path"], args.get("vae_path"))
        self.model.prepare()
        self.model.codec_model.update(force=True)
        self.model.codec_model.debug = False

    @torch.no_grad()
    def evaluate(self):
        args = self.args
        self.model.eval()
        if not args["decoder_only"]:
            return self.evaluate_folder()
        src, dst = args["decoder_bin_path"], args["decoder_out_path"]
        if os.path.isdir(src):
            os.makedirs(dst, exist_ok=True)
            files = sorted(glob.glob(os.path.join(src, "*.bin")))
            by_size = {}
            from .entropy.utils import decode_i, read_from_file
            for f in files:
                d = decode_i(read_from_file(f), 14, 64)
                by_size.setdefault((d["pad_height"], d["pad_width"], d["height"], d["width"]), []).append(f)
            for group in by_size.values():
                for i in range(0, len(group), args["batch_size"]):
                    chunk = group[i:i + args["batch_size"]]
                    recon = self.model.decode(fp=chunk, out="u8")
                    for f, img in zip(chunk, recon):
                        save_image_u8(img, os.path.join(dst, os.path.splitext(os.path.basename(f))[0] + ".png"))
            return
        # clamp(-1,1)*0.5+0.5 and save_image's 8-bit quantisation (inference.py:106-107) happen in the output kernel;
        # a picture with NaN / Inf raises kernels.HalfRangeError there instead of being written
        recon = self.model.decode(fp=src, out="u8")
        save_image_u8(recon[0], dst)


    @torch.no_grad()
    def evaluate_folder(self):
        """The encode + decode loop of the reference (inference.py:109-134), one picture at a time."""
        import pandas as pd
        args = self.args
        out = args["output_path"]
        recon_path, stream_path = prepare_output_dir(out)
        src = args["eval_image_path"]
        names = sorted(f for f in os.listdir(src) if f.endswith((".jpg", ".png")))
        torch.manual_seed(0)                                   # inference.py:28: the draw of latent_dist.sample()
        rows = []
        for idx, fname in enumerate(names):
            name = os.path.splitext(fname)[0]
            image = load_image(os.path.join(src, fname))[None].to(self.model.device)
            _, _, H, W = image.shape
            Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
            noise = torch.randn(1, Hp // 8, Wp // 8, 4, device=self.model.device)
            stream_file = os.path.join(stream_path, f"{name}.bin")
            _, bpp = self.model.encode(image, fp=stream_file, noise=noise)
            recon = self.model.decode(fp=stream_file, out="u8")
            save_image_u8(recon[0], os.path.join(recon_path, f"{name}.png"))
            bpp = dict(bpp)
            bpp["image_name"] = name
            rows.append(pd.DataFrame(bpp, index=[idx]))
        detail = pd.concat(rows) if rows else pd.DataFrame()
        detail.to_csv(os.path.join(out, "detail_bpp_caption.xlsx"))
        avg = detail.mean(numeric_only=True).to_dict() if rows else {}
        pd.DataFrame([avg]).to_csv(os.path.join(out, "summary_bpp.xlsx"))
        print(avg)
        return avg


def main(argv=None):
    return Evaluator(parse_args(argv)).evaluate()


if __name__ == "__main__":
    main()
