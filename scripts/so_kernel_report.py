#!/usr/bin/env python3
"""Every kernel that ships in libquanto_hip.so with its register / spill / scratch / LDS figures, read from the code objects' metadata notes.

    python scripts/so_kernel_report.py [path/to/lib.so] [--spills] [--json]

The shared library carries one clang offload bundle per translation unit (section .hip_fatbin); each bundle holds a gfx950 ELF whose
NT_AMDGPU_METADATA note lists the kernels.  No ROCm perl tooling needed (roc-obj-ls is not usable in this image): the bundle format is parsed
here and llvm-readelf prints the notes.  Used by tests/test_build_invariants.py: no kernel the product dispatch can reach may spill.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, cur)
            triple = data[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos = cur


def kernels(path):
    out = []
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for b in notes.split("  - .agpr_count:")[1:]:
            g = lambda k, b=b: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))  # noqa: E731
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            out.append({"name": name, "agpr": int(b.splitlines()[0].strip()), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                        "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                        "lds": g("group_segment_fixed_size")})
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(out, names):
        k["demangled"] = d
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = args[0] if args else os.path.join(here, "optimum_quanto_amd", "lib", "libquanto_hip.so")
    ks = kernels(path)
    if "--spills" in sys.argv:
        ks = [k for k in ks if k["vgpr_spill"] or k["scratch"]]
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
        return
    for k in sorted(ks, key=lambda k: k["demangled"]):
        print(f"{k['demangled'][:150]:150s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} spill {k['vgpr_spill']:3d} scratch {k['scratch']:4d} lds {k['lds']:6d}")
    print(f"{len(ks)} kernels, {os.path.getsize(path) / 1e6:.1f} MB", file=sys.stderr)


if __name__ == "__main__":
    main()
