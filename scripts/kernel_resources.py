#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a hipcc -S listing: python scripts/kernel_resources.py file.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt not in dem:
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", b).group(1)  # noqa: E731
    print(f"{dem[:110]:110s} agpr {b.splitlines()[0].strip():>3s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s}")
