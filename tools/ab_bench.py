#!/usr/bin/env python3
"""A/B of two builds of liblavie_hip.so on the SAME GPU box (devices differ by >10 %: never compare across calls).
Runs bench.py once per library (child processes, LAVIE_HIP_LIB override) in alternation and prints video-latents/s
plus the per-class device time of one instrumented forward.
Usage: python tools/ab_bench.py [--rounds 2] [--steps 1] libA.so libB.so ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, steps):
    # "path.so@N" runs that library with LAVIE_FORCE_TILE=N (GEMM kernel selection override, see igemm.h)
    lib, _, mode = lib.partition("@")
    env = dict(os.environ, LAVIE_HIP_LIB=os.path.abspath(lib))
    if mode:
        env["LAVIE_FORCE_TILE"] = mode
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "1",
                          "--cpu-steps", "0"], env=env, capture_output=True, text=True)
    if out.returncode != 0:
        print(out.stderr[-2000:])
        raise SystemExit(f"bench.py failed with {lib}")
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    for r in range(a.rounds):
        for lib in a.libs:
            d = run(lib, a.steps)
            cls = " ".join(f"{c['name'].split('_')[0]}={c['ms']:.2f}" for c in d["kernel_breakdown"]["classes"] if c["launches"])
            print(f"round {r} {os.path.basename(lib):30s} {d['value']:.4f} lat/s  {d['ms_per_step'] / 50:.2f} ms/fwd | {cls}", flush=True)


if __name__ == "__main__":
    main()
