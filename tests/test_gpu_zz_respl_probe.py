"""Runs LAST of the GPU files (alphabetical order): a probe of an opt-in data flow in its own process -- whatever it does cannot affect the
tests before it, and it is xfail(strict=False): it reports, it never fails the suite."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TOL_MX = 1e-3


@pytest.mark.xfail(strict=False, reason="EV_MX_RESPL=1 is opt-in: its kernels are validated (tests/test_gpu_ops.py::test_mx_residual_from_planes), the engine "
                                          "flow had not run on a GPU when round 3 ended -- an XPASS here means it can become the default")
def test_residual_from_planes_flow_probe():
    """The opt-in data flow of DESIGN.md "Next" 0c (ResBlock residuals rebuilt from the plane sets, no fp32 residual streams) on every reference
    fixture, in its own process (a fault there cannot take this suite down): waveform, DC-free waveform and mel inside the 1e-3 contract."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, EV_MX_RESPL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_respl_probe.py")], env=env, capture_output=True, text=True, timeout=300)
    line = [x for x in r.stdout.splitlines() if x.startswith("RESPL_PROBE ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    res = json.loads(line[-1][len("RESPL_PROBE "):])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "respl_probe.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    assert res and all(v["dur_ok"] and v["mel"] < TOL_MX and v["wav"] < TOL_MX and v["wav_ac"] < TOL_MX for v in res.values()), res
