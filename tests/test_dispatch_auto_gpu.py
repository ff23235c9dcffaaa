"""AUTO must stay close to the best forced kernel OFF the grid its thresholds were fitted on (K, N in {1024, 4096, 14336}).

`scripts/auto_vs_best.py` times AUTO and every kernel that accepts a shape in hipGraphs inside one process (decode shapes rotate over
> 256 MB of weights, best of five replays each); this test runs its quick list and fails when AUTO is more than 15 % (+ 0.5 us of
timer noise) behind the best choice.  The full table is committed under profiles/ (r04_auto_vs_best.jsonl)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

# A TIMING assertion: marked `perf`, not `gpu`, so that the driver's `-m gpu -x` parity run cannot be cut short by clock noise
# (run it with `pytest -m perf` on the GPU box; the visit scripts do)
pytestmark = [pytest.mark.perf, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a ROCm device")]


@pytest.mark.parametrize("fmt", ["int4", "int8"])
def test_auto_is_within_15_percent_of_the_best_forced_kernel(fmt):
    import optimum_quanto_amd  # noqa: F401
    from auto_vs_best import QUICK, sweep

    def behind(rows):
        return [r for r in rows if r["auto_us"] > 1.15 * r["best_us"] + 0.5]

    bad = behind(sweep(QUICK, formats=(fmt,)))
    # a shape that misses the gate is measured again, twice: inside a long test session the clock state moves a few-microsecond kernel by
    # more than the margin; a real dispatch hole misses every time
    for _ in range(2):
        if not bad:
            break
        bad = behind(sweep([(r["M"], r["K"], r["N"]) for r in bad], formats=(fmt,)))
    assert not bad, "\n".join(f"{r['fmt']} ({r['M']},{r['K']},{r['N']}): AUTO={r['auto_kernel']} {r['auto_us']} us, best={r['best']} {r['best_us']} us"
                               for r in bad)
