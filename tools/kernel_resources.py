#!/usr/bin/env python
"""Registers, scratch, spills and LDS of every kernel in the SHIPPED library (gnuradio4_amd/libgr4hip.so): walks the clang offload bundles of the .so, takes each gfx950
code object and reads its AMDGPU metadata note with llvm-readelf.  No GPU needed.  Used by tests/test_abi_host.py (no kernel of the library may spill) and to regenerate
the per-kernel table of DESIGN.md:   python tools/kernel_resources.py [--md] [filter]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield data[pos + off: pos + off + size]
        pos += len(MAGIC)


def kernels(path=None):
    """-> list of dicts {name, vgpr, agpr, sgpr, scratch, vspill, sspill, lds, wg} for every kernel of the library"""
    path = path or os.path.join(ROOT, "gnuradio4_amd", "libgr4hip.so")
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k, d="0": (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, d])[1]
            out.append(dict(name=g("name", "?"), vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                            vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")), lds=int(g("group_segment_fixed_size")), wg=int(g("max_flat_workgroup_size"))))
    names = subprocess.run([CXXFILT], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, nm in zip(out, names):
        k["demangled"] = re.sub(r"^void ", "", nm)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = args[0] if args else ""
    ks = [k for k in kernels() if flt in k["demangled"]]
    md = "--md" in sys.argv
    if md:
        print("| kernel | VGPR | AGPR | SGPR | LDS B | scratch B | VGPR spills | SGPR spills |\n|---|---|---|---|---|---|---|---|")
    for k in sorted(ks, key=lambda k: k["demangled"]):
        nm = re.sub(r"\(.*", "", k["demangled"])
        if md:
            print(f"| `{nm}` | {k['vgpr']} | {k['agpr']} | {k['sgpr']} | {k['lds']} | {k['scratch']} | {k['vspill']} | {k['sspill']} |")
        else:
            print(f"{nm[:100]:100s} vgpr {k['vgpr']:4d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} lds {k['lds']:6d} scratch {k['scratch']:5d} vspill {k['vspill']:3d} sspill {k['sspill']:3d}")
    bad = [k for k in ks if k["vspill"] or k["scratch"]]
    print(f"# {len(ks)} kernels, {len(bad)} with scratch / VGPR spills", file=sys.stderr)
