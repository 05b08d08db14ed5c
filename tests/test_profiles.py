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
ROUND = "r6"


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
    assert base["cpu_model"] and base["host_threads"] >= base["cores"]            # the CPU baseline names its processor
    # ... and carries the reference harness's one-thread setting and an all-threads entry (a fresh CPU-only child process, bounded)
    assert base["one_thread"]["cores"] == 1 and base["one_thread"]["value"] > 0
    allt = base.get("all_threads")                          # (opt-in since round 6 measured it: it times out on the pool's 256-thread hosts)
    assert allt is None or (allt["kind"] == "port" and allt["sample"] and (allt["value"] is None or allt["value"] > 0))
    # round 6: `value` is the FIRST of three timed regions of K steps, the other two say how far one region can be trusted
    rep = line["repeatability"]
    assert rep["regions"] == 3 and rep["values"][0] == line["value"] and rep["min"] <= line["value_median_of_3"] <= rep["max"]
    assert rep["spread_pct"] == pytest.approx(100.0 * (rep["max"] - rep["min"]) / rep["max"], abs=0.02) and rep["spread_pct"] < 3.0
    assert line["config"]["rccl_ranks"] == 1 and line["config"]["launcher"] == "none"


def test_roofline_object_is_consistent():
    line = _line("bench_line")
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=2e-4)
    # achieved = algorithmic flops of the family's launches of one step / their summed duration -- since round 5 the launches of
    # the untimed sequential decode right after the timed region (the GPU to itself); the figure of the pipelined timed region
    # rides beside it and cannot be the better one
    assert roof["achieved"] == pytest.approx(roof["algorithmic_gflop_per_step"] / (roof["launches_per_step"] * roof["avg_launch_us"]) * 1e3,
                                             rel=2e-3)
    assert roof["frac_in_pipeline"] == pytest.approx(roof["achieved_in_pipeline"] / roof["peak"], abs=2e-4)
    assert roof["achieved_in_pipeline"] <= roof["achieved"] and roof["avg_launch_us_in_pipeline"] >= roof["avg_launch_us"]
    assert roof["launches_per_step_in_pipeline"] == roof["launches_per_step"]
    assert roof["isolated"]["achieved"] == roof["achieved"] and "GPU to itself" in roof["achieved_how"]
    # round 6: `achieved` is the MEDIAN of three warmed isolated passes, its spread rides beside it, and the
    # median pass is consistent with the event-bracketed stage S of one batch alone -- every conv / GEMM launch of a decode fits inside it
    assert roof["achieved_definition"].startswith("isolated-median")
    assert len(roof["achieved_passes"]) == 3 and roof["achieved_min"] <= roof["achieved"] <= roof["achieved_max"]
    assert roof["achieved"] == sorted(roof["achieved_passes"])[1]
    assert (roof["achieved_max"] - roof["achieved_min"]) / roof["achieved"] < 0.05
    assert roof["isolated_inconsistent"] is False
    assert roof["conv_gemm_ms_per_decode"] <= 1.05 * line["device"]["stage_s_ms"]
    assert sum(v["ms_per_step"] for v in roof["by_kernel"].values()) == pytest.approx(roof["conv_gemm_ms_per_decode"], rel=2e-2)
    # the power roof for the whole path and per stage
    assert roof["whole_path_frac_of_peak_at_clock"] == pytest.approx(roof["whole_path_tflops"] / roof["peak_at_clock"], abs=2e-4)
    stages = line["device"]["stages"]
    for name in ("codec_synthesis", "unet_step", "vae_decoder"):
        st = stages[name]
        assert st["frac"] == pytest.approx(st["tflops"] / roof["peak"], abs=2e-4)
        assert st["frac_of_peak_at_clock"] == pytest.approx(st["tflops"] / roof["peak_at_clock"], abs=2e-4)
    assert sum(stages[n]["ms"] for n in ("codec_synthesis", "unet_step", "vae_decoder")) + stages["image_out_ms"] == \
        pytest.approx(line["device"]["stage_s_ms"], rel=1e-2)
    assert roof["clock_ghz"] > 0.5 and roof["achieved"] <= roof["peak_at_clock"]
    # whole-path rate <= the dominant kernel's own rate alone <= peak
    assert roof["whole_path_tflops"] <= roof["achieved"] <= roof["peak"]
    # what the run did NOT measure says so, and names a committed record of this or an earlier round
    for key in ("traffic_from", "mfma_util_from"):
        m = re.match(r"profiles/(r\d)_pmc_(traffic|mfma)\.json", roof[key])
        assert m and "not measured by this run" in roof[key], roof[key]
        assert os.path.exists(os.path.join(PROF, os.path.basename(m.group(0))))
    assert "not measured by this run" in roof["sustained_on_random_data"]["from"]
    # traffic is that record's figure for the same kernel (the counters repeat to well within 2 % from batch to batch), and is
    # not below the algorithmic bytes
    pmc = _line("pmc_traffic")
    assert roof["traffic"] == pytest.approx(pmc[roof["kernel"]]["hbm_bytes_per_launch"], rel=3e-2)
    assert pmc[roof["kernel"]]["launches"] % roof["launches_per_step"] == 0          # (one PMC pass = the regions of a 1-step run)
    assert roof["traffic"] >= roof["algorithmic_bytes_per_launch"]
    assert roof["mfma_util_pmc"] == pytest.approx(_line("pmc_mfma")[roof["kernel"]]["mfma_util"], abs=3e-2)


def test_side_lines_of_the_default_run_agree_with_the_full_lines():
    """configs 4 and 5, the bf16 library and a batch of 32 ride in the line the driver's one command
    prints; each must agree with the builder-run line of its own mode from the same batch within box / run variance."""
    line = _line("bench_line")
    for key in ("random_init", "z_only", "batch32", "bf16", "vae_up3_bf16", "encode_decode"):
        side = line[key]
        assert side["value"] and side["value"] > 0 and side["steps"] >= 2 and side["how"], key
        assert side["value"] == pytest.approx(side["images_per_step"] * 1e3 / side["ms_per_step"], rel=2e-3)
    assert line["z_only"]["value"] == pytest.approx(_line("bench_line_zonly")["value"], rel=0.06)
    assert line["bf16"]["value"] == pytest.approx(_line("bench_line_bf16")["value"], rel=0.06)
    assert line["encode_decode"]["value"] == pytest.approx(_line("bench_line_encdec")["value"], rel=0.08)
    assert line["encode_decode"]["parity"]["z_indices_equal"] in (True, False)
    assert abs(line["encode_decode"]["parity"]["bpp_delta_rel"]) < 5e-3
    # twice the batch buys (almost) nothing: the small UNet levels are not what a step waits for (profiles/r5_negatives.txt, 3)
    assert line["batch32"]["value"] == pytest.approx(line["value"], rel=0.05)
    # the other weight preset ran the headline's K steps in the headline's bracket
    assert line["random_init"]["steps"] == line["steps"]


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
    # the roofline figure: the kernel alone (the bench's untimed sequential pass against the sequential profile)
    alone = _line("bench_line")["roofline"]["avg_launch_us"]
    assert alone == pytest.approx(_family_avg_us(_stats("sequential_kernel_stats")), rel=5e-2)
    # inside the pipeline the two clocks bracket different things: rocprofv3's duration runs from a kernel's first wave to its last,
    # the HIP events on the launch stream also contain the time the launch waits for CUs the other batch's kernel holds (and the
    # profile's row also holds the warm-up steps and the one isolated pass): the live figure cannot be the smaller one, and both
    # must tell the same story (round 5, two boxes: 1 406 live against 1 268 us)
    under = _line("bench_line_under_profiler")["roofline"]
    prof = _family_avg_us(_stats("bench_default_kernel_stats"))
    assert 0.97 * prof <= under["avg_launch_us_in_pipeline"] <= 1.3 * prof
    assert under["avg_launch_us"] <= under["avg_launch_us_in_pipeline"]


def test_design_quotes_the_committed_numbers():
    """no headline figure of the live DESIGN.md may contradict profiles/r6_*: the numbers it quotes for
    the round are looked up in the records, and the document stays a document (<= 48 KB; the history is HISTORY.md)."""
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        text = f.read()
    assert len(text.encode()) <= 48 * 1024 and os.path.exists(os.path.join(ROOT, "HISTORY.md"))
    assert not re.search(r"(?<![A-Z_])R[56]_[A-Z0-9_]+", text)    # (no placeholder of the template survived; -DEXP_R5_BASELINE is a flag)
    line = _line("bench_line")
    roof = line["roofline"]
    for value, what in ((line["value"], "images/s of the committed line"), (roof["achieved"], "TFLOP/s of the dominant kernel alone"),
                        (roof["achieved_in_pipeline"], "TFLOP/s inside the pipeline"), (roof["clock_ghz"], "clock"),
                        (line["z_only"]["value"], "z_only side line"), (line["encode_decode"]["value"], "encode_decode side line"),
                        (line["bf16"]["value"], "bf16 side line"), (line["batch32"]["value"], "batch32 side line")):
        shown = {f"{value:.1f}", f"{value:.2f}", f"{value:.3f}", f"{value:.0f}"}
        shown |= {f"{value:,.{n}f}".replace(",", " ") for n in (0, 1, 2)}              # (thousands written "1 144.87")
        assert any(re.search(r"(?<![\d.])" + re.escape(s) + r"(?![\d])", text) for s in shown), (what, value)
    with open(os.path.join(PROF, f"{ROUND}_sequential_kernel_stats.csv")) as f:
        head = f.readline() + f.readline()
    total = re.search(r"([\d.]+) ms of kernel time per batch", head).group(1)
    assert total in text, total
