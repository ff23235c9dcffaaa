#!/usr/bin/env python3
"""Audit of the generated code of the four-wave AGPR layouts of qmm_mfma_large.hip (cdna_hip_programming.md 5.7 item 4: hipcc neither
sees nor pads asm MFMAs).  For every kernel whose MFMAs are inline asm, between the first and the last v_mfma:

  * no scratch access, and
  * every compiler-generated v_accvgpr_write / v_accvgpr_mov is followed - before any un-padded MFMA - by an MFMA whose asm string
    opens with ``s_nop 4`` (VALU write of an AGPR -> MFMA reading it as C), and
  * every compiler-generated v_accvgpr_read / _mov is preceded - after the last un-padded MFMA - by an MFMA that ends with
    ``s_nop 15`` (MFMA result -> other reader: 12 states).

    python scripts/audit_agpr_isa.py [path/to/qmm_mfma_large.s]      (compiles the file with hipcc -S when no path is given)
Exit status 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_asm():
    out = os.path.join(tempfile.mkdtemp(prefix="qh_audit_"), "lt.s")
    src = os.path.join(ROOT, "optimum_quanto_amd", "csrc", "qmm_mfma_large.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return out


def audit(path):
    lines = open(path).read().split("\n")
    problems, audited = [], 0
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_ZN2qh2lt24qbytes_mfma_large_kernel\w+:", ln)]
    for st in starts:
        end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[st:end]
        mf = [i for i, ln in enumerate(body) if "v_mfma" in ln]
        if not mf or "#ASMSTART" not in body[mf[0] - 1] and "s_nop" not in body[mf[0] - 1]:
            continue  # builtin MFMAs: hipcc's own hazard recognizer covers them
        audited += 1
        name = body[0].split(":")[0]
        pending_write = None   # compiler wrote an AGPR, no padded MFMA since
        since_plain_mfma = None  # an un-padded-at-the-end MFMA was the last MFMA seen
        for i in range(mf[0], mf[-1] + 1):
            ln = body[i]
            if "scratch_" in ln:
                problems.append(f"{name}: scratch access inside the MFMA region (line {i})")
            if "v_mfma" in ln:
                head = "s_nop 4" in body[i - 1]
                tail = "s_nop 15" in body[i + 1]
                if pending_write is not None and not head:
                    problems.append(f"{name}: MFMA at line {i} follows a compiler accumulator write (line {pending_write}) without wait states")
                pending_write = None
                since_plain_mfma = None if tail else i
            elif "v_accvgpr_write" in ln or "v_accvgpr_mov" in ln or "v_accvgpr_read" in ln:
                if "v_accvgpr_read" not in ln:
                    pending_write = i
                if since_plain_mfma is not None and ("v_accvgpr_read" in ln or "v_accvgpr_mov" in ln):
                    problems.append(f"{name}: compiler accumulator read at line {i} follows an un-padded MFMA (line {since_plain_mfma})")
                    since_plain_mfma = None
    return audited, problems


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    audited, problems = audit(path)
    print(f"audited {audited} asm-MFMA kernels, {len(problems)} problem(s)")
    for p in problems[:40]:
        print("  " + p)
    sys.exit(1 if problems or audited == 0 else 0)
