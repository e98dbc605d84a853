#!/usr/bin/env python3
"""One line per kernel of the BUILT libgeo4d_hip.so: VGPRs, SGPRs, scratch bytes per lane, static LDS - read from the AMDGPU metadata
notes of the code objects inside the library's clang offload bundles (no recompilation, no GPU).
usage: tools/so_kernel_table.py [path/to/libgeo4d_hip.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    """Yield (target id, bytes) of every device code object bundled into the file (one bundle per translation unit)."""
    blob = open(path, "rb").read()
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, idlen = struct.unpack_from("<QQQ", blob, off)
            tid = blob[off + 24:off + 24 + idlen].decode()
            off += 24 + idlen
            if "amdgcn" in tid and size:
                yield tid, blob[pos + o:pos + o + size]
        pos = blob.find(MAGIC, pos + len(MAGIC))


def kernels(path):
    out = []
    for tid, co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
            name = get("name").group(1).strip("'\"")
            out.append(dict(name=name, vgpr=int(get("vgpr_count").group(1)), sgpr=int(get("sgpr_count").group(1)),
                            scratch=int(get("private_segment_fixed_size").group(1)), lds=int(get("group_segment_fixed_size").group(1))))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    r = [re.sub(r"^void ", "", n.replace("(anonymous namespace)::", "").replace("geo4d_gemm::", "")) for n in r]
    return [re.sub(r"\((?!.*>).*$", "", n) for n in r]        # drop the argument list (the last parenthesis group after the template arguments)


if __name__ == "__main__":
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "geo4d_amd", "csrc", "libgeo4d_hip.so"))
    for k, n in sorted(zip(ks, demangle([k["name"] for k in ks])), key=lambda kn: (-kn[0]["scratch"], kn[1])):
        print(f"{n[:100]:100s} VGPR {k['vgpr']:3d}  SGPR {k['sgpr']:3d}  scratch {k['scratch']:5d}  static LDS {k['lds']:6d}")
    print(f"{len(ks)} kernels, {sum(1 for k in ks if k['scratch'])} with scratch")
