#!/usr/bin/env python3
"""usage: tools/kernel_resources.py [lib.so] [name filter]
Registers, scratch and LDS of every kernel in a built library, read from the code objects' metadata notes (what
-Rpass-analysis=kernel-resource-usage prints at compile time, but from the shipped binary): unbundles .hip_fatbin
(clang offload bundle), runs llvm-readelf --notes on each gfx950 code object."""
import os, re, struct, subprocess, sys, tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
DEMANGLE = "c++filt"


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = 0
    while True:
        at = data.find(magic, at)
        if at < 0:
            return
        n = struct.unpack_from("<Q", data, at + 24)[0]
        q = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24 : q + 24 + tl].decode()
            q += 24 + tl
            if "gfx" in triple and size:
                yield triple, data[at + off : at + off + size]
        at += len(magic)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stratum_amd", "libstratum_hip.so")
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for triple, blob in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in notes.split("  - .agpr_count:")[1:]:
            get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, block) or [None, "?"])[1]
            rows.append((get("name"), get("vgpr_count"), get("sgpr_count"), block.split()[0], get("private_segment_fixed_size"), get("group_segment_fixed_size"), get("vgpr_spill_count"), get("sgpr_spill_count")))
    names = subprocess.run([DEMANGLE], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("%-72s %5s %5s %5s %8s %7s %6s %6s" % ("kernel", "vgpr", "sgpr", "agpr", "scratch", "lds", "vspill", "sspill"))
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))
        if filt in n and "rocprim" not in n and "hipcub" not in n:  # (the library's sort / scan kernels are not ours to tune)
            print("%-72s %5s %5s %5s %8s %7s %6s %6s" % ((n[:72],) + r[1:]))


if __name__ == "__main__":
    main()
