"""Child process of tests/test_harness_ranks.py: the stage-1 CLI harness (onedc_amd.inference.Evaluator) on a stand-in model
that needs no GPU -- encode / decode are cheap deterministic functions of (picture, noise) -- so that the rank logic
(RANK / WORLD_SIZE from the environment, output directory, part files, merged tables) runs as separate processes on CPU."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import inference


class StandInModel:
    device = torch.device("cpu")

    def __init__(self, log):
        self.log = log

    def eval(self):
        return self

    def encode(self, image, fp=None, noise=None):
        h = hashlib.sha256(image.numpy().tobytes() + noise.numpy().tobytes()).digest()
        with open(fp, "wb") as f:
            f.write(h * 4)
        with open(self.log, "a") as f:
            f.write(os.path.basename(fp) + "\n")
        n = image.shape[2] * image.shape[3]
        return h, {"bpp": 8.0 * len(h) * 4 / n, "bpp_y": float(noise.sum()), "bpp_z": h[0] / 255.0}

    def decode(self, fp=None, out="u8"):
        with open(fp, "rb") as f:
            b = f.read()
        return torch.tensor(list(b[:48]), dtype=torch.uint8).view(1, 4, 4, 3)

    def pipeline(self, z_only=False):
        return StandInPipeline(self.log)


class StandInPipeline:
    """DecodePipeline's surface (map / close): a picture is a deterministic function of its container's bytes."""

    def __init__(self, log):
        self.log, self.closed = log, False

    def map(self, batches, out="f32", to_host=False):
        assert out == "u8" and to_host and not self.closed
        res = []
        for b in batches:
            with open(self.log, "a") as f:
                f.write(f"batch{len(b)}\n")
            res.append(torch.stack([torch.tensor(list(hashlib.sha256(blob).digest() * 2)[:48], dtype=torch.uint8).view(4, 4, 3)
                                    for blob in b]))
        return res

    def close(self):
        self.closed = True


if __name__ == "__main__":
    out, src = sys.argv[1], sys.argv[2]
    fail_at = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rank = int(os.environ.get("RANK", "0"))
    model = StandInModel(os.path.join(os.path.dirname(out), f"log_{os.path.basename(out)}_rank{rank}.txt"))
    if fail_at >= 0:
        enc = model.encode

        def failing(image, fp=None, noise=None, _n=[0]):
            if _n[0] == fail_at:
                raise RuntimeError("stand-in failure")
            _n[0] += 1
            return enc(image, fp=fp, noise=noise)
        model.encode = failing
    if os.environ.get("HARNESS_MODE") == "decode_folder":      # --decoder_only with a directory: the sharded pipelined decode
        args = {"decoder_only": True, "decoder_bin_path": src, "decoder_out_path": out, "batch_size": 3}
    else:
        args = {"decoder_only": False, "output_path": out, "eval_image_path": src, "batch_size": 4}
    inference.Evaluator(args, model=model).evaluate()
