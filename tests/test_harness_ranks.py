"""The drop-in harness under a multi-process launch (reference src/inference.py:37,49-59,82,127: main-process-guarded
rmtree, pictures sharded over the processes): two CPU processes with RANK / WORLD_SIZE set like a launcher does, a stand-in
model (tests/_harness_runner.py).  Checked: the pictures are partitioned, the output directory is prepared once, and the
merged tables and files equal the single-process run's."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RUNNER = os.path.join(HERE, "_harness_runner.py")


def make_pictures(folder, n):
    from PIL import Image
    os.makedirs(folder)
    rng = np.random.default_rng(0)
    for i in range(n):
        h, w = 64 + 8 * (i % 3), 72 + 16 * (i % 2)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(folder, f"pic{i:02d}.png"))


def launch(out, src, rank=None, world=None, run_id="t", extra=()):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if rank is not None:
        env.update(RANK=str(rank), WORLD_SIZE=str(world), ONEDC_RUN_ID=run_id)
    return subprocess.Popen([sys.executable, RUNNER, out, src, *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def finish(procs, ok=True):
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert (p.returncode == 0) == ok, e.decode()[-2000:]


def test_two_ranks_partition_the_folder_and_merge_one_table(tmp_path):
    src = str(tmp_path / "pictures")
    make_pictures(src, 7)
    single, multi = str(tmp_path / "single"), str(tmp_path / "multi")
    finish([launch(single, src)])
    # an earlier run's output is in the way: only rank 0 may remove it, and rank 1 must not start before that
    os.makedirs(os.path.join(multi, "recon"))
    open(os.path.join(multi, "recon", "stale.png"), "w").close()
    finish([launch(multi, src, r, 2) for r in (1, 0)])
    logs = [open(tmp_path / f"log_multi_rank{r}.txt").read().split() for r in (0, 1)]
    assert sorted(logs[0] + logs[1]) == sorted(open(tmp_path / "log_single_rank0.txt").read().split())
    assert logs[0] == [f"pic{i:02d}.bin" for i in (0, 2, 4, 6)] and logs[1] == [f"pic{i:02d}.bin" for i in (1, 3, 5)]
    for name in ("detail_bpp_caption.xlsx", "summary_bpp.xlsx"):
        assert open(os.path.join(single, name)).read() == open(os.path.join(multi, name)).read(), name
    for sub in ("recon", "stream"):
        a, b = sorted(os.listdir(os.path.join(single, sub))), sorted(os.listdir(os.path.join(multi, sub)))
        assert a == b and len(a) == 7 and "stale.png" not in b
        assert not filecmp.cmpfiles(os.path.join(single, sub), os.path.join(multi, sub), a, shallow=False)[1]
    # no rendezvous files are left behind
    assert sorted(os.listdir(multi)) == ["detail_bpp_caption.xlsx", "recon", "stream", "summary_bpp.xlsx"]


def test_rows_of_finished_pictures_survive_a_failure(tmp_path):
    """The reference (and round 3 here) kept the rows in memory until the end of the folder: an exception at picture k lost
    pictures 0 .. k - 1.  Now every row is appended to the rank's part file at once."""
    src = str(tmp_path / "pictures")
    make_pictures(src, 5)
    out = str(tmp_path / "out")
    finish([launch(out, src, extra=("3",))], ok=False)
    import pandas as pd
    part = pd.read_csv(os.path.join(out, ".detail_part0.csv"), index_col=0)
    assert list(part["image_name"]) == ["pic00", "pic01", "pic02"]
    assert sorted(os.listdir(os.path.join(out, "recon"))) == ["pic00.png", "pic01.png", "pic02.png"]
