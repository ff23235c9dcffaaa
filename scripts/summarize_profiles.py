#!/usr/bin/env python3
"""Copy the judged summaries of `scripts/collect_profiles.sh <round> ...` from gpurun_out/profiles_<round>/ into profiles/.

    python scripts/summarize_profiles.py r01

Per workload: rNN_<w>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), rNN_<w>_bench_under_rocprof.json (the bench line of
that profiled run) and pmc_<w>.json (HBM-side bytes per bench step from the separate FETCH_SIZE / WRITE_SIZE passes, with the
gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads -> x2).
A step that launches two qh:: kernels (dequantize + dense GEMM) sums both.
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", f"profiles_{rnd}")
    dst = os.path.join(ROOT, "profiles")
    for stats in sorted(glob.glob(os.path.join(src, "*_kernel_stats.csv"))):
        w = os.path.basename(stats)[: -len("_kernel_stats.csv")]
        shutil.copy(stats, os.path.join(dst, f"{rnd}_{w}_kernel_stats.csv"))
        bench = os.path.join(src, f"trace_{w}.bench.json")
        if os.path.exists(bench):
            lines = [ln for ln in open(bench).read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(dst, f"{rnd}_{w}_bench_under_rocprof.json"), "w").write(lines[-1] + "\n")
        counters = os.path.join(src, f"{w}_hbm_counters.json")
        if os.path.exists(counters):
            c = json.load(open(counters))
            fetch = sum(v["mean_per_launch"] for v in c.get("FETCH_SIZE", {}).values())
            write = sum(v["mean_per_launch"] for v in c.get("WRITE_SIZE", {}).values())
            if fetch or write:
                out = {
                    "workload": w, "round": rnd, "kernels": sorted(set(c.get("FETCH_SIZE", {})) | set(c.get("WRITE_SIZE", {}))),
                    "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
                    "correction": "gfx950 rocprofv3 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 "
                                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as reported",
                    "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
                    "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --eager --steps 8 --warmup 2",
                }
                json.dump(out, open(os.path.join(dst, f"pmc_{w}.json"), "w"), indent=1)
        print("copied", w)


if __name__ == "__main__":
    main()
