"""Compiler-level guards for the phased conv_gemm K loop (geo4d_amd/csrc/gemm_kernel_v3.h), checked on the gfx950 ISA hipcc emits
(cross-compiles without a GPU). The schedule only works if the compiler keeps three properties that it silently broke several times while
the kernel was written (profiles/r03_gemm_v3_phased.md):
  * no scratch: a spilled value is reloaded with a VMEM load, whose wait is an `s_waitcnt vmcnt(0)` in the middle of the loop;
  * no `vmcnt(0)` between the first and the last MFMA of the kernel (the counted waits keep a slab of LDS-DMA in flight across barriers);
  * the staging loads are `buffer_load_dwordx4 ... lds` on SGPR resources and SGPR offsets - no waterfall loop (`v_readfirstlane` + branch)
    around them, which is what a resource or offset the compiler believes divergent turns into."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

CASES = [
    ("160x320 split x split", "bf16x3_t, 160, 320, 2, 4, 2, false", 7),     # counted wait = pieces of one slab of this tile: 1 + 1 + 3 + 2
    ("192x256 GEGLU writing the split format", "bf16x3_t, 192, 256, 2, 4, 2, true", 6),    # 1 + 1 + 2 + 2
    ("128x256 bf16", "bf16_t, 128, 256, 2, 4, 0, false", 6),
]


@pytest.mark.slow
@pytest.mark.parametrize("name,targs,nwait", CASES, ids=[c[0] for c in CASES])
def test_phased_k_loop_keeps_its_pipeline(name, targs, nwait):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        src, asm = os.path.join(d, "k.hip"), os.path.join(d, "k.s")
        with open(src, "w") as f:
            f.write('#include "gemm_kernel_v3.h"\nnamespace geo4d_gemm {\n'
                    f"template __global__ void conv_gemm_v3_kernel<{targs}>(const geo4d_conv_gemm_t, const int, const int);\n}}\n")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/geo4d_amd/csrc",
                            "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S",
                            "-Rpass-analysis=kernel-resource-usage", "-o", asm, src], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        usage = r.stderr
        assert re.search(r"ScratchSize \[bytes/lane\]: 0\b", usage), re.findall(r"(?:ScratchSize|VGPRs Spill)[^\n]*", usage)
        vgprs = int(re.search(r"VGPRs: (\d+)", usage).group(1))
        assert vgprs <= 256, vgprs                     # two waves per SIMD
        lines = open(asm).read().splitlines()
    mf = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32" in l]
    assert len(mf) >= 8 * 4, len(mf)
    loop = lines[mf[0]:mf[-1] + 1]
    text = "\n".join(loop)
    assert "scratch_" not in text
    assert not re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", text), "a drained DMA queue inside the K loop"
    counted = [int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)]
    assert len(counted) >= 7 and nwait in counted and min(counted) >= nwait - 1, counted
    if name.startswith("160x320"):
        # the emitted loop IS the designed schedule: per barrier interval (staging loads, fragment reads, MFMAs, counted wait) of the two
        # slabs of a pair. 160x320 on 2 x 4 waves: A0 / A1 = 96 / 64 rows (2 / 1 pieces per wave), B0 / B1 = 192 / 128 rows (3 / 2 pieces),
        # fragments A0 B0 = 3 blocks x 2 reads, A1 B1 = 2 x 2; quadrant MFMAs = blocks x blocks x 3 (bf16x3).
        want = [(1, 4, 27, 7), (2, 4, 18, 7), (3, 6, 12, 8), (2, 4, 18, 7),       # even slab: (A0,B0) (A0,B1) (A1,B1) (A1,B0)
                (1, 6, 18, 7), (2, 4, 27, 7), (2, 6, 18, 6), (3, 6, 12, 7)]       # odd slab:  (A0,B1) (A0,B0) (A1,B0) (A1,B1)
        # the slab-pair loop = the loop header (LLVM's "Loop Header: Depth=2" annotation) followed by all 150 MFMAs of a pair; its text
        # runs to the pair's closing barrier (blocks laid out behind it are the once-per-tile window set-up: no loads, reads or MFMAs)
        heads = [i for i, l in enumerate(lines) if re.search(r"Loop Header: Depth=2\b", l)]
        spans = [(sum("v_mfma" in x for x in lines[h:(heads[k + 1] if k + 1 < len(heads) else len(lines))]), h) for k, h in enumerate(heads)]
        nm, h = max(spans)
        assert nm == 150, spans
        tail = lines[h:]
        last_mfma = max(i for i, l in enumerate(tail) if "v_mfma" in l)
        end = next(i for i in range(last_mfma, len(tail)) if "s_barrier" in tail[i])
        body = tail[:end + 1]
        got, cur = [], [0, 0, 0, None]
        for l in body:
            if "s_barrier" in l:
                got.append(tuple(cur))
                cur = [0, 0, 0, None]
            elif "buffer_load_dwordx4" in l and " lds" in l:
                cur[0] += 1
            elif "ds_read_b128" in l:
                cur[1] += 1
            elif "v_mfma" in l:
                cur[2] += 1
            elif re.search(r"s_waitcnt vmcnt\((\d+)\)", l):
                cur[3] = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
        assert cur[:3] == [0, 0, 0], cur               # nothing but the back branch after the last barrier
        assert got == want, got
    loads = [i for i, l in enumerate(loop) if "buffer_load_dwordx4" in l and " lds" in l]
    assert len(loads) >= 8, len(loads)
    for i in loads:
        assert re.search(r"buffer_load_dwordx4 v\d+, s\[\d+:\d+\], s\d+ offen lds", loop[i]), loop[i]
        window = "\n".join(loop[max(0, i - 6):i + 4])
        assert "v_readfirstlane" not in window and "s_and_saveexec" not in window, "waterfall loop around a staging load:\n" + window


def test_built_library_scratch_is_confined_to_the_listed_kernels():
    """Every kernel of the built libgeo4d_hip.so, read from the code objects' metadata (tools/so_kernel_table.py): spilled registers
    (scratch) only in the kernels LISTED here, each with its bound. Round 6: the HEADLINE mode (bf16x3m) no longer launches a spilling
    attention kernel - its spatial self-attention is `flash_attn2_kernel<f16_t, false, 2>` (216 VGPRs, no scratch: asserted below), its
    cross-attention `flash_attn_kernel<f16_t, 2, 1, 1, false>`. Two of the listed ones ARE still on a product path and are a known debt
    (VERDICT r4 weak #3): `flash_attn2_kernel<bf16x3_t, true, 2>` (the default spatial self-attention of the STRICT bf16x3 mode only: 76 bytes
    since round 5 staged K / V^T through buffer resources - 120 before -, still ~20 registers over its 256-register budget at two waves
    per SIMD; measured faster than the spill-free one-wave build all the same) and
    the 256x256 second-generation GEMM tile (24-32 bytes spilled before the K loop, reloaded in the epilogue). The other attention
    instantiations in the list are A/B builds ops.attention only launches on request (variant 1..3)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import so_kernel_table as skt
    lib = os.path.join(ROOT, "geo4d_amd", "csrc", "libgeo4d_hip.so")
    if not os.path.exists(lib) or not os.path.exists(skt.READELF):
        pytest.skip("library or llvm-readelf missing")
    ks = skt.kernels(lib)
    names = skt.demangle([k["name"] for k in ks])
    assert len(ks) > 150 and sum(1 for n in names if n.startswith("conv_gemm_v3_kernel")) >= 16, len(ks)     # (round 4 pruned ~100 instantiations)
    allowed = {"flash_attn_kernel<float, 1, 2, 2, false>", "flash_attn_kernel<bf16x3_t, 1, 2, 2, false>", "flash_attn_kernel<bf16x3_t, 1, 2, 2, true>",
               "flash_attn_kernel<bf16_t, 1, 1, 4, false>", "flash_attn_kernel<f16_t, 1, 1, 4, false>"}
    allowed.add("flash_attn2_kernel<bf16x3_t, true, 2>")     # round 4: the two-waves-per-SIMD build of the skewed-block kernel (variant 5: 9 registers over its 256 budget, measured faster than the spill-free one-wave build at N >= 1024 all the same)
    # round 4: the branch-free buffer-resource fast paths of the register epilogue cost the 256x256 second-generation tile a few registers that are
    # spilled BEFORE the K loop and reloaded in the epilogue (<= 96 bytes; nothing inside the loop - asserted on the ISA by the phased-loop tests)
    small_ok = lambda n, sc: (n.startswith("conv_gemm_v2_kernel<bf16x3_t, 256, 256") or n.startswith("conv_gemm_v2_kernel<bf16_t, 256, 256")) and sc <= 96
    bad = [(n, k["scratch"]) for k, n in zip(ks, names) if k["scratch"] and n not in allowed and not small_ok(n, k["scratch"])]
    assert not bad, bad
    # every kernel the bf16x3m mode's attention classes launch: no scratch
    headline = {"flash_attn2_kernel<f16_t, false, 2>", "flash_attn_kernel<f16_t, 2, 1, 1, false>", "flash_attn_kernel<f16_t, 1, 1, 1, false>",
                "flash_attn_kernel<f16_t, 1, 2, 2, false>", "temporal_attn_kernel<f16_t>"}
    seen = {n: k["scratch"] for k, n in zip(ks, names) if n in headline}
    assert set(seen) == headline and not any(seen.values()), seen
    assert all(k["vgpr"] <= 256 for k, n in zip(ks, names) if n.startswith("conv_gemm"))       # 8-wave tiles: two waves per SIMD
