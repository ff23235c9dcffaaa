"""pytest plugin (``-p quanto_amd_plugin``) for gate G4: the reference's OWN test-suite, unchanged, with this backend
plugged into the unmodified reference - ``import optimum.quanto`` first, then ``import optimum_quanto_amd``."""
import optimum.quanto  # noqa: F401  the unmodified reference (first on sys.path)

import optimum_quanto_amd  # noqa: F401,E402  plug-in mode: CUDA (= ROCm) kernels of quanto::*, registry entry quanto_hip, F.linear routing
from optimum_quanto_amd.library import plugin  # noqa: E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

assert plugin.installed(), "plug-in mode did not engage"

_kernels = {}


def pytest_runtest_teardown(item):
    try:
        k = quanto_hip.lib.last_kernel()
    except Exception:  # library not loaded (no device): nothing to record
        return
    if k:
        _kernels[k] = _kernels.get(k, 0) + 1


def pytest_terminal_summary(terminalreporter):
    terminalreporter.write_line(f"quanto_amd plug-in: last libquanto_hip kernel seen after a test, by count: {dict(sorted(_kernels.items()))}")
