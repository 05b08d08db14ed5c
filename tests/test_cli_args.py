"""Argument handling of the two command-line harnesses (host logic, no GPU): what each mode needs is checked while
parsing, and an existing --output_path is only removed when it looks like an earlier run's output."""
import os

import pytest

from onedc_amd import inference, inference_z_only


@pytest.fixture()
def cfg(tmp_path):
    p = tmp_path / "config.yaml"
    p.write_text("vae_attn_patch: 16\nlora_config:\n  lora_rank: 64\n  lora_alpha: 8.0\n")
    return str(p)


@pytest.mark.parametrize("mod", [inference, inference_z_only])
def test_modes_require_their_paths(mod, cfg, tmp_path):
    base = ["--config_path", cfg, "--checkpoint_path", "ckpt"]
    with pytest.raises(SystemExit):
        mod.parse_args(base)                                              # evaluate mode without output / pictures
    with pytest.raises(SystemExit):
        mod.parse_args(base + ["--output_path", str(tmp_path / "out")])
    with pytest.raises(SystemExit):
        mod.parse_args(base + ["--decoder_only"])
    with pytest.raises(SystemExit):
        mod.parse_args(base + ["--decoder_only", "--decoder_bin_path", "a.bin"])
    with pytest.raises(SystemExit):
        mod.parse_args(base + ["--output_path", str(tmp_path / "out"), "--eval_image_path", str(tmp_path / "missing")])
    a = mod.parse_args(base + ["--decoder_only", "--decoder_bin_path", "a.bin", "--decoder_out_path", "a.png"])
    assert a["decoder_only"] and a["vae_attn_patch"] == 16 and a["unet_ckpt"] is None
    pics = tmp_path / "pics"
    pics.mkdir()
    a = mod.parse_args(base + ["--output_path", str(tmp_path / "out"), "--eval_image_path", str(pics)])
    assert a["eval_image_path"] == str(pics) and not a["decoder_only"]


def test_output_dir_is_only_removed_when_it_is_an_earlier_run(tmp_path):
    out = tmp_path / "out"
    recon, stream = inference.prepare_output_dir(str(out))                # fresh
    assert os.path.isdir(recon) and os.path.isdir(stream)
    (out / "recon" / "a.png").write_bytes(b"x")
    inference.prepare_output_dir(str(out))                                # an earlier run: replaced, like the reference
    assert not (out / "recon" / "a.png").exists()
    other = tmp_path / "precious"
    other.mkdir()
    (other / "thesis.tex").write_text("...")
    with pytest.raises(SystemExit):
        inference.prepare_output_dir(str(other))
    assert (other / "thesis.tex").exists()
    empty = tmp_path / "empty"
    empty.mkdir()
    inference.prepare_output_dir(str(empty))
    f = tmp_path / "file"
    f.write_text("x")
    with pytest.raises(SystemExit):
        inference.prepare_output_dir(str(f))
