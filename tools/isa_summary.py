#!/usr/bin/env python3
"""Per-kernel resource summary of a hipcc -S dump (gfx950): VGPRs, SGPRs, spills, MFMA / ds_read / LDS-DMA / wait counts.
Usage: python tools/isa_summary.py file.s [substring-of-kernel-name [--dump]]
Build the dump with: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ilavie_amd/csrc -Iinclude \
    -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only -o /tmp/x.s lavie_amd/csrc/<file>.hip"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    dump = "--dump" in sys.argv
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", body).group(1))
        meta[m.group(1)] = (g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"))
    lines = text.split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s", l) or re.match(r"^[a-z_]\w*:\s+; @", l)]
    for j, (i, name) in enumerate(starts):
        end = next((k for k in range(i, len(lines)) if lines[k].startswith("\t.section") or ".end_amdhsa_kernel" in lines[k]), len(lines))
        body = lines[i:end]
        dn = demangle(name)
        if want and want not in dn:
            continue
        cnt = lambda pat: sum(1 for l in body if re.search(pat, l))
        v, s, sp, scr = meta.get(name, (-1, -1, -1, -1))
        pats = {"mfma": r"v_mfma", "ds_read": r"ds_read", "glds": r"global_load_lds|buffer_load.* lds", "vmcnt0": r"vmcnt\(0\)",
                "lgkm0": r"lgkmcnt\(0\)", "barrier": r"s_barrier", "scratch_ops": r"scratch_"}
        counts = " ".join(f"{k} {cnt(p)}" for k, p in pats.items())
        print(f"{dn[:110]}\n    vgpr {v} sgpr {s} spill {sp} scratch {scr} | {counts}")
        if dump:
            print("\n".join(body))


if __name__ == "__main__":
    main()
