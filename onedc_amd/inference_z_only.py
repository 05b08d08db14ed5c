"""Command-line harness of the ex-low-rate (0.0034 bpp, "z-only") model with the reference's surface
(src/models/sd15_onedc_codec_z_only/inference.py:93-161).

    python -m onedc_amd.inference_z_only --config_path CFG.yaml --checkpoint_path CKPT --output_path OUT \
        --eval_image_path PICTURES [--vae_path vae.safetensors]

Same flags and YAML handling as the reference (CLI over file, LOCAL_RANK override :157-159); per picture: reflect
padding to a multiple of 64 (:106-109), `forward` = analysis -> FSQ indices -> picture from the indices alone
(model_sd15_with_codec_stage1.py:198-218), `recon/<name>.png`, and the two tables `detail_bpp_caption.xlsx` /
`summary_bpp.xlsx` with the reference's columns: `bpp_z` = 14 bit per 64x64 patch of the PADDED picture over the
pixels of the original (:114-117) and `bpp_hard_y` (:126; identically 0 in eval mode: y_q = 0 and sigma = 0 there,
compression_model.py:142-150).  The reference writes the tables with `DataFrame.to_excel` (:138-142); openpyxl is not
installed offline, so the same names carry CSV content when it is missing (as the stage-1 harness's tables always do).

Additions (the reference never writes a bitstream for this model): `stream/<name>.bin` -- the container with an
empty y part, 16 B header + ceil(14 n / 8) B of indices = 268 B for a 768x768 picture (readme.md:24 "0.24 KB") -- and
`--decoder_only --decoder_bin_path X.bin|DIR --decoder_out_path Y.png|DIR` which decodes such files.
"""
import argparse
import os
import torch
import yaml

from .inference import (check_mode_args, decode_folder_pipelined, evaluate_sharded, load_image, picture_noise, rank_world,
                        save_image, save_image_u8)


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", type=str, required=True)
    parser.add_argument("--checkpoint_path", type=str, required=True)
    parser.add_argument("--output_path", type=str, required=False)
    parser.add_argument("--eval_image_path", type=str, required=False)
    parser.add_argument("--vae_path", type=str, default=None, required=False)
    parser.add_argument("--decoder_only", action="store_true")
    parser.add_argument("--decoder_bin_path", type=str, default="", required=False)
    parser.add_argument("--decoder_out_path", type=str, default="", required=False)
    parser.add_argument("--batch_size", type=int, default=16, required=False)
    cli = vars(parser.parse_args(argv))
    check_mode_args(parser, cli)
    with open(cli["config_path"]) as f:
        config = yaml.safe_load(f) or {}
    config.update(cli)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != config.get("local_rank"):
        config["local_rank"] = env_local_rank
    config["unet_ckpt"] = None
    return config


def write_table(frame, path):
    """`to_excel` like the reference where openpyxl exists, CSV content under the same name otherwise."""
    try:
        import openpyxl  # noqa: F401
        frame.to_excel(path)
    except ImportError:
        frame.to_csv(path)


class Evaluator:
    def __init__(self, args, model=None):
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
        """This rank's share of the folder (inference.evaluate_sharded: rank 0 alone prepares the output directory, the
        reference's `accelerator.is_main_process` guard :52-57, and merges the tables)."""
        args = self.args
        self.model.eval()
        if args["decoder_only"]:
            return self.decode_files()
        if getattr(self.model, "vae_encoder", None) is None:
            raise RuntimeError("the checkpoint carries no encoder weights (enc.*, hyper_enc.*, VAE encoder)")
        out = args["output_path"]
        recon_path, stream_path = os.path.join(out, "recon"), os.path.join(out, "stream")

        def one_picture(idx, name, path):
            image = load_image(path)[None].to(self.model.device)
            noise = picture_noise(idx, image.shape[2], image.shape[3], self.model.device)      # (:30: the draw of latent_dist.sample())
            enc, recon = self.model.forward_z_only(image, noise=noise)
            save_image(recon[0].clamp(-1.0, 1.0) * 0.5 + 0.5, os.path.join(recon_path, f"{name}.png"))
            with open(os.path.join(stream_path, f"{name}.bin"), "wb") as f:
                f.write(enc["stream"][0])
            return {"image_name": name, "bpp_hard_y": enc["bpp_hard_y"], "bpp_z": enc["bpp_z"]}

        return evaluate_sharded(out, args["eval_image_path"], self.rank, self.world, one_picture, write_table)

    def decode_files(self):
        args = self.args
        src, dst = args["decoder_bin_path"], args["decoder_out_path"]
        if not os.path.isdir(src):
            recon = self.model.decode_z_only_stream(fp=src, out="u8")
            save_image_u8(recon[0], dst)
            return
        return decode_folder_pipelined(self.model, src, dst, args["batch_size"], self.rank, self.world, z_only=True)


def main(argv=None):
    return Evaluator(parse_args(argv)).evaluate()


if __name__ == "__main__":
    main()
