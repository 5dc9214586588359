"""The profile post-processing tools on a synthetic rocprofv3 kernel table: since round 6 the Adam kernel runs in chunks on the
second stream beside the NEXT step's forward, so steps are delimited by the gradient-norm pass (one launch per update on the
data-path stream) -- ``tools/prof_summary.py <db> N last N`` must again cover exactly N steps."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, steps=6, fwd_bwd=10, chunks=4):
    db = sqlite3.connect(path)
    db.execute("create table kernels(name text, start int, end int, queue_id int)")
    t = 0
    for step in range(steps):
        for k in range(fwd_bwd):
            db.execute("insert into kernels values(?,?,?,?)", ("gemm_kernel(args)", t, t + 1000, 1))
            t += 1200
            if step > 0 and k < chunks:  # the previous step's update, beside this step's first kernels, on the second queue
                db.execute("insert into kernels values(?,?,?,?)", ("adam_kernel(args)", t - 1100, t - 300, 2))
        db.execute("insert into kernels values(?,?,?,?)", ("sumsq_kernel(args)", t, t + 500, 1))
        t += 700
    db.commit()
    db.close()


def test_prof_summary_counts_whole_steps_with_a_chunked_update(tmp_path):
    p = str(tmp_path / "t.db")
    _trace(p)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), p, "3", "last", "3"],
                         capture_output=True, text=True, check=True).stdout
    rows = {ln.split("(")[0].strip(): ln.split() for ln in out.splitlines() if "_kernel" in ln}
    assert "the last 3 steps" in out.splitlines()[0]
    assert int(rows["gemm_kernel"][1]) == 30 and int(rows["adam_kernel"][1]) == 12 and int(rows["sumsq_kernel"][1]) == 3


def test_prof_queues_takes_one_step_and_the_norm_pass_queue_as_the_data_path(tmp_path):
    p = str(tmp_path / "t.db")
    _trace(p)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_queues.py"), p], capture_output=True, text=True,
                         check=True).stdout
    assert "kernels 15" in out.splitlines()[0]  # 10 data-path kernels + 4 update chunks + the norm pass
    assert "queue 1 (main) n 11" in out and "queue 2 n 4" in out
