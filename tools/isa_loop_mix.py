#!/usr/bin/env python3
"""Static instruction mix of the innermost MFMA loop of a kernel, from the gfx950 ISA hipcc emits (no GPU needed).
usage: tools/isa_loop_mix.py geo4d_amd/csrc/attention.hip 'flash_attn_kernelI6bf16_tLi1ELi2ELi2E' [more mangled-name substrings ...]
Prints per kernel: instructions per loop iteration by class, and a VALU : MFMA issue-cycle estimate (wave64: 4 cycles per VALU
instruction, transcendentals 5/3 of that - MI355X_MICROARCH.md; v_mfma_f32_32x32x16 32 cycles, 16x16x32 16 cycles per SIMD)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)", op): return "valu transcendental"
    if op.startswith("v_pk_"): return "valu packed"
    if op.startswith("v_cvt"): return "valu convert"
    if "permlane" in op or "dpp" in op or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "valu cross-lane"
    if op.startswith("v_"): return "valu other"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds read"
    if op.startswith("ds_"): return "lds other"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_"): return "salu"
    return "other"


def innermost_mfma_loop(lines):
    """The shortest backward-branch loop of the listing that contains an MFMA: [header label .. back branch]."""
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            if any("v_mfma" in b for b in body) and (best is None or len(body) < len(best)):
                best = body
    return best


def loop_mix(lines):
    best = innermost_mfma_loop(lines)
    mix, forms = {}, {}
    for b in best or []:
        t = b.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        c = classify(t[0])
        mix[c] = mix.get(c, 0) + 1
        if c == "mfma":
            forms[t[0]] = forms.get(t[0], 0) + 1
    return mix, forms


def main():
    src, pats = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/geo4d_amd/csrc",
                        "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S", "-o", asm, src], check=True,
                       capture_output=True)
        lines = open(asm).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for pat in pats:
        for k, (i, name) in enumerate(starts):
            if pat not in name:
                continue
            end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
            mix, forms = loop_mix(lines[i:end])
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\((?!.*>).*$", "", dem.replace("(anonymous namespace)::", "").replace("void ", ""))
            valu = sum(n for c, n in mix.items() if c.startswith("valu")) + mix.get("valu transcendental", 0) * 2 / 3
            mf = sum(n * (32 if "32x32" in f else 16) for f, n in forms.items())
            print(f"{dem}: {sum(mix.values())} instructions per iteration")
            print("   " + "  ".join(f"{c} {n}" for c, n in sorted(mix.items(), key=lambda x: -x[1])))
            print(f"   MFMA {forms} = {mf} pipe cycles | VALU ~{valu * 4:.0f} issue cycles | VALU : MFMA = {valu * 4 / max(mf, 1):.2f}")


if __name__ == "__main__":
    main()
