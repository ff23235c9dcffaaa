"""Build-container-only tests against the REAL reference (skipped where /root/reference does not exist, i.e. on the GPU box):

* ``oracle/reference_cpu_path.py`` - what bench.py's cpu_baseline times - is the reference's CPU path, call for call;
* plug-in mode: importing this package after ``optimum.quanto`` overrides the ROCm kernels of
  the reference's operators, adds the fused ops and registers ``quanto_hip`` in the reference's extension registry
  (library/extensions/hip/__init__.py:18-36, library/extensions/extension.py:58-86, tests/library/test_extensions.py:23-24).

Each check runs in its own interpreter: both packages define ``quanto::`` operators, and the order of the two imports is
exactly what is under test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("QUANTO_REFERENCE") or ("/root/reference" if os.path.isdir("/root/reference") else os.path.join(ROOT, ".refcopy"))
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optimum", "quanto")), reason="reference checkout not present")


def _run(mode):
    env = dict(os.environ, PYTHONPATH="")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "reference_subprocess.py"), mode], capture_output=True, text=True,
                          timeout=900, env=env, cwd=ROOT)
    assert proc.returncode == 0 and "ALL-OK" in proc.stdout, f"stdout:\n{proc.stdout[-3000:]}\nstderr:\n{proc.stderr[-3000:]}"
    return proc.stdout


@needs_reference
def test_reference_cpu_path_restatement_equals_the_reference():
    out = _run("cpu_path")
    assert "torch.float32: ok" in out and "torch.bfloat16: ok" in out


@needs_reference
def test_plugin_mode_installs_into_an_imported_reference():
    out = _run("plugin")
    assert "CUDA kernels overridden" in out and "get_extension('quanto_hip') resolves" in out


@pytest.mark.gpu
@needs_reference
def test_plugin_mode_runs_reference_tensors_on_the_fast_kernels():
    """Needs a ROCm device AND a reference checkout (scripts/run_reference_tests_gpu.sh ships a scratch copy as .refcopy/): F.linear on
    a reference WeightQBitsTensor living on the device ends in this library's GEMV / streaming / fused kernels."""
    out = _run("plugin_gpu")
    assert "reference QLinear(qint4).forward -> gemv" in out
