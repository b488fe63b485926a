"""Soak of the captured-graph paths: scripts/soak_graphs.py as a CHILD process -- a crash inside hipGraphLaunch is a
segmentation fault of the process, which must fail this test, not kill the test run.  >= 300 captures and >= 3 000 replays through the
split-graph executor (the default: the runtime never sees a multi-branch graph), a shorter run through the runtime's own executor (the
fallback, held to three streams per capture)."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, caps, reps, timeout):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    env = dict(os.environ)
    if mode == "runtime":
        for k in ("DEBUG_HIP_DYNAMIC_QUEUES", "GPU_MAX_HW_QUEUES", "DEBUG_HIP_FORCE_GRAPH_QUEUES"):      # the runtime's defaults for its own executor
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_graphs.py"), mode, str(caps), str(reps)], env=env, capture_output=True,
                       text=True, timeout=timeout)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, f"soak ({mode}) died with exit code {p.returncode}:\n{tail}"
    m = re.search(r"soak ok: .*?(\d+) captures, (\d+) trainer replays, (\d+) decode turns.*?widest capture (\d+) streams", p.stdout)
    assert m, tail
    return [int(g) for g in m.groups()], p.stdout


def test_split_executor_soak_300_captures_3000_replays():
    (caps, reps, turns, widest), out = _run("split", 300, 3000, 900)
    assert caps >= 300 and reps + 12 * turns >= 3000, out
    assert "split usable {0: True}" in out, out


def test_runtime_executor_fallback_soak_stays_within_three_streams():
    (caps, reps, turns, widest), out = _run("runtime", 60, 600, 600)
    assert caps >= 60 and widest <= 3, out
