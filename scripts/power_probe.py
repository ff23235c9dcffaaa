#!/usr/bin/env python3
"""Board power and shader clock while a GEMM runs in its steady state: turns "power-bound" from an inference (cycles / wall time)
into a reading.  For each variant a hipGraph of the cfg2 call (or torch.matmul bf16 = hipBLASLt) is replayed for --seconds while a
thread samples the amdgpu hwmon files (power1_average / power1_input in uW, freq1_input = sclk in Hz; `rocm-smi --json` as a
fallback) every 50 ms; the first second is dropped.  One JSON line per variant: us per launch, mean / max watts, mean sclk.

    python scripts/power_probe.py --cfgs 0 1 5 --matmul [--seconds 4] [--const]
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)), ("mclk_Hz", ("freq2_input",)),
                           ("temp_mC", ("temp1_input",))):
            for n in names:
                p = os.path.join(d, n)
                if key not in out and os.path.exists(p):
                    try:
                        int(open(p).read())
                        out[key] = p
                    except (OSError, ValueError):
                        pass
        if out:
            break
    return out


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter()}
            if self.files:
                for k, p in self.files.items():
                    try:
                        row[k] = int(open(p).read())
                    except (OSError, ValueError):
                        pass
            else:
                try:
                    js = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
                    card = next(iter(js.values()))
                    for k, v in card.items():
                        if "Power" in k and "W" in k:
                            row["power_uW"] = int(float(v) * 1e6)
                        if "sclk" in k.lower() and "(" in str(v):
                            row["sclk_Hz"] = int(float(str(v).split("(")[1].split("Mhz")[0]) * 1e6)
                except Exception as e:  # noqa: BLE001
                    row["err"] = str(e)[:80]
            self.rows.append(row)
            time.sleep(0.05)


def measure(label, graph, steps, seconds, files, flops):
    s = Sampler(files)
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        graph.replay()
        torch.cuda.synchronize()
        n += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    s.stop = True
    s.join()
    us = e0.elapsed_time(e1) * 1e3 / (5 * steps)
    rows = [r for r in s.rows if r["t"] - t0 > 1.0]
    mean = lambda k: (sum(r[k] for r in rows if k in r) / max(1, sum(1 for r in rows if k in r)))  # noqa: E731
    out = {"variant": label, "us": round(us, 2), "tflops": round(flops / us / 1e6, 1), "samples": len(rows), "watts_mean": round(mean("power_uW") / 1e6, 1),
           "watts_max": round(max([r.get("power_uW", 0) for r in rows] or [0]) / 1e6, 1), "sclk_mhz_mean": round(mean("sclk_Hz") / 1e6, 1),
           "sclk_mhz_min": round(min([r["sclk_Hz"] for r in rows if "sclk_Hz" in r] or [0]) / 1e6, 1), "temp_c": round(mean("temp_mC") / 1e3, 1),
           "source": "hwmon" if files else "rocm-smi"}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", nargs="*", default=["0"])
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--matmul", action="store_true")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--const", action="store_true", help="constant operands (the data-independent power floor)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import optimum_quanto_amd  # noqa: F401

    files = hwmon_files()
    print(json.dumps({"hwmon": files}), flush=True)
    kind, M, K, N, _ = bench.WORKLOADS[args.workload]
    flops, _ = bench.algorithmic_work(kind, M, K, N)
    x, sets = bench.build_inputs(kind, M, K, N, dev, 1, seed=1)
    if args.const:
        x.fill_(1.0)
        sets[0][0].fill_(1)
    steps = 20
    for cfg in args.cfgs:
        os.environ["QUANTO_HIP_LARGE_CFG"] = cfg
        step = bench.make_step(kind, x, sets, K, N)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(steps):
                step()
        measure(f"{args.workload} QUANTO_HIP_LARGE_CFG={cfg}" + (" const" if args.const else ""), g, steps, args.seconds, files, flops)
    os.environ.pop("QUANTO_HIP_LARGE_CFG", None)
    if args.matmul:
        a = x if x.dtype == torch.bfloat16 else torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        if args.const:
            w.fill_(1.0)
        for _ in range(3):
            torch.matmul(a, w.t())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(steps):
                torch.matmul(a, w.t())
        measure("torch.matmul bf16 (hipBLASLt), dense weights" + (" const" if args.const else ""), g, steps, args.seconds, files, flops)


if __name__ == "__main__":
    main()
