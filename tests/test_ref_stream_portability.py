"""Can ANY fp32 implementation other than the one that wrote it decode a reference-made y stream?  A measurement.

The four-part prior truncates (ln sigma - ln 0.11) / step to a table index (reference src/modules/entropy/
entropy_models.py:355-362); a sigma that differs in its last bits next to a boundary picks the neighbouring CDF and rANS
desynchronises for the rest of the picture.  The golden streams g8_* / g9_* were written by the REFERENCE's own modules
(tests/golden/make_golden.py) under torch-CPU fp32 with 8 threads in the build container.  Here the same torch-CPU fp32
arithmetic (oracle/torch_ref.py, checked bit-exact against those planes at the writing configuration) is re-run with
OTHER thread counts -- the reference's own harness sets torch.set_num_threads(1), src/inference.py:29 -- and, in child
processes, with oneDNN restricted to another instruction set.  Every table-index decision that differs from the stream's is
counted, teacher-forced (the reference's symbols are fed back, so one flip cannot cascade).

Measured in the build container (Xeon, AVX-512; profiles/r3_ref_stream_portability.txt, with the GPU box's host beside it):
1 thread flips 95 of the 294 912 decisions of the random-init 768x768 stream and 7 of the trained-like one; 2, 4 and 8
threads flip none; oneDNN on AVX2 flips again.  So a reference-made stream is portable between two runs of the reference
itself only if thread count and CPU instruction set match -- the HIP prior (exact f32, fixed fma order) sits in the same
class: [4, 4, 21, 62] and [0, 2, 2, 2] flips on the same two streams (tests/test_gpu_fullsize.py).

The tests assert what must hold anywhere: the flip rate of a torch-CPU fp32 run stays below 1e-3 (it is rounding noise,
not an algorithmic difference), and the run that matches the writer's configuration reproduces the stream exactly where
that configuration is available (same container: 8 threads)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import golden
from onedc_amd import synth
from oracle import torch_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = [("g8_ref_stream_768.npz", None), ("g9_ref_stream_768_trained.npz", "trained_like")]


def teacher_forced_flips(name, preset, threads):
    g = golden(name)
    sd = synth.codec_state_dict(0, preset=preset)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        with torch.no_grad():
            fused, _ = R.z_indices_to_params(sd, torch.from_numpy(g["indices"]).long())
            flips = []

            def hop(k, plane):
                flips.append(int((plane.to(torch.int16).numpy() != g["index_planes"][k]).sum()))
                return torch.from_numpy(g["symbols"][k].astype(np.float32))
            R.decompress_four_part_prior(sd, fused, hop)
    finally:
        torch.set_num_threads(prev)
    return flips, int(g["index_planes"].size)


def report(tag):
    lines = [f"[{tag}] torch {torch.__version__}, {os.cpu_count()} host threads, "
             f"cpu: {cpu_model()}, oneDNN ISA cap: {os.environ.get('DNNL_MAX_CPU_ISA', 'none')}"]
    worst = 0.0
    for name, preset in STREAMS:
        for threads in (1, 2, 8, 16):
            if threads > (os.cpu_count() or 1) * 2:
                continue
            flips, total = teacher_forced_flips(name, preset, threads)
            worst = max(worst, sum(flips) / total)
            lines.append(f"[{tag}] {name}: torch-CPU fp32 prior, {threads:3d} threads: table-index flips per hop {flips} of "
                         f"{total} ({sum(flips) / total:.2e})")
    print("\n".join(lines))
    return worst


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def test_torch_cpu_prior_against_reference_made_streams_across_thread_counts():
    worst = report("host")
    assert worst <= 1e-3


def test_torch_cpu_prior_with_another_onednn_isa():
    """The same count in a child process whose oneDNN may use AVX2 only (the writer had AVX-512 where the CPU offers it)."""
    code = ("import sys; sys.path[:0] = [%r, %r]; import test_ref_stream_portability as t; "
            "w = t.report('host, DNNL_MAX_CPU_ISA=AVX2'); sys.exit(0 if w <= 1e-3 else 1)" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, DNNL_MAX_CPU_ISA="AVX2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_torch_cpu_prior_on_the_gpu_box_host():
    """Host-only (no kernel runs): the same measurement on the GPU box's CPU -- a third machine class besides the build
    container's Xeon and the MI355X."""
    worst = report("gpu-box host")
    assert worst <= 1e-3
