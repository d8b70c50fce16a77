"""The cpu_baseline leg of bench.py: the real reference's CPU-thread actor path (oracle/_ref/rela*.so, built
from the reference's sources where they exist) driven by oracle/ref_actor_bench.py.  Skipped where the
prebuilt module is absent."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not glob.glob(os.path.join(ROOT, "oracle", "_ref", "rela*.so")),
                    reason="oracle/_ref is built only where /root/reference exists")
def test_reference_actor_path_runs_and_reports_a_rate():
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_actor_bench.py"), "--threads", "2", "--games",
                          "4", "--seconds", "2", "--warmup", "1", "--num_action", "18"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["threads"] == 2 and rec["games"] == 4 and rec["env_steps_per_s"] > 0
    assert rec["buffer_size"] > 0  # the reference's actors really inserted into the reference's replay
