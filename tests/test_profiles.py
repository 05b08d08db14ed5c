"""The committed measurement records under profiles/ (copied there by tools/keep_profiles.py from one tools/profile_round.sh batch)
must say the same thing in every file: the bench line carries the contract's keys, its roofline object agrees with itself and with
the PMC traffic record it quotes, and the rocprofv3 kernel statistics name kernels that exist in this tree.  No GPU needed."""
import csv
import glob
import io
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
ROUND = "r4"


def _line(name):
    with open(os.path.join(PROF, f"{ROUND}_{name}.json")) as f:
        return json.load(f)


def _stats(name):
    with open(os.path.join(PROF, f"{ROUND}_{name}.csv")) as f:
        text = "".join(l for l in f if not l.startswith("#"))
    return list(csv.DictReader(io.StringIO(text)))


def test_bench_line_carries_the_contract():
    line = _line("bench_line")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    # value = images of all ranks / time of the K timed steps
    per_step = line["config"]["images_per_gpu_per_step"] * line["n_gpus"]
    assert line["value"] == pytest.approx(per_step / (line["ms_per_step"] * 1e-3), rel=2e-3)
    base = line["cpu_baseline"]
    assert base["kind"] in ("reference", "port") and base["cores"] >= 1 and base["value"] > 0 and base["sample"]


def test_roofline_object_is_consistent():
    line = _line("bench_line")
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=2e-4)
    # achieved = algorithmic flops of the family's launches of one step / their summed duration
    assert roof["achieved"] == pytest.approx(roof["algorithmic_gflop_per_step"] / (roof["launches_per_step"] * roof["avg_launch_us"]) * 1e3,
                                             rel=2e-3)
    # the dominant kernel cannot be slower alone than inside the pipeline, nor faster than the peak at the clock it held
    assert roof["isolated"]["achieved"] >= roof["achieved"]
    assert roof["clock_ghz"] > 0.5 and roof["achieved"] <= roof["peak_at_clock"]
    # whole-path rate <= the dominant kernel's own rate alone <= peak
    assert roof["whole_path_tflops"] <= roof["peak"]
    # traffic is the PMC record's figure for the same kernel (bench.py quotes the record that was committed when it ran, i.e. the
    # batch before: the counters repeat to well within 2 %), and is not below the algorithmic bytes
    pmc = _line("pmc_traffic")
    assert roof["traffic"] == pytest.approx(pmc[roof["kernel"]]["hbm_bytes_per_launch"], rel=2e-2)
    assert pmc[roof["kernel"]]["launches"] == roof["launches_per_step"]
    assert roof["traffic"] >= roof["algorithmic_bytes_per_launch"]
    assert roof["mfma_util_pmc"] == pytest.approx(_line("pmc_mfma")[roof["kernel"]]["mfma_util"], abs=2e-2)


def test_every_mode_has_a_line_from_the_same_round():
    for name in ("bench_line_bf16", "bench_line_sequential", "bench_line_zonly", "bench_line_encdec", "bench_line_under_profiler"):
        line = _line(name)
        assert line["value"] > 0 and line["n_gpus"] == 1
    assert _line("bench_line_bf16")["dtype"] == "bf16" and _line("bench_line")["dtype"] == "f16"
    # a sequential decode cannot beat the pipelined one
    assert _line("bench_line_sequential")["value"] <= _line("bench_line")["value"] * 1.01


def test_kernel_statistics_name_kernels_of_this_tree():
    sources = ""
    for path in glob.glob(os.path.join(ROOT, "onedc_amd", "csrc", "*.hip")):
        with open(path) as f:
            sources += f.read()
    for name in ("sequential_kernel_stats", "bench_default_kernel_stats"):
        rows = _stats(name)
        assert rows and abs(sum(float(r["Percentage"]) for r in rows) - 100.0) < 0.5
        ours = 0.0
        for r in rows:
            m = re.search(r"(\w+_kernel)\b", r["Name"])
            if m and "rocclr" not in r["Name"] and "at::native" not in r["Name"]:
                assert m.group(1) in sources, r["Name"]
                ours += float(r["Percentage"])
        assert ours > 95.0, ours                           # the device time is this package's own kernels
    # the record the bench line's roofline is about leads the sequential profile
    assert "conv_win4_kernel" in _stats("sequential_kernel_stats")[0]["Name"]


def _family_avg_us(rows):
    tot = calls = 0.0
    for r in rows:
        if "conv_win4_kernel<0" in r["Name"] or "conv_win4_kernel<1" in r["Name"]:      # plain / GroupNorm on load, with or without x2
            tot += float(r["TotalDurationNs"])
            calls += int(r["Calls"])
    return tot / calls / 1e3


def test_live_launch_durations_agree_with_the_profiler():
    """the average launch duration bench.py measures live with HIP events must agree with rocprofv3's for the same
    command -- the pipelined run under the profiler, and the kernel alone (the bench's isolated pass against the sequential profile)."""
    under = _line("bench_line_under_profiler")["roofline"]
    assert under["avg_launch_us"] == pytest.approx(_family_avg_us(_stats("bench_default_kernel_stats")), rel=5e-2)
    alone = _line("bench_line")["roofline"]["isolated"]["avg_launch_us"]
    assert alone == pytest.approx(_family_avg_us(_stats("sequential_kernel_stats")), rel=5e-2)
