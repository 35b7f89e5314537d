"""Kernels of a built library straight from its code objects (no compiler, no GPU):
    python tools/so_kernels.py [libreid_hip.so]      -> name, VGPRs, AGPRs, scratch bytes per lane, LDS bytes of every kernel
The .so carries one clang offload bundle per translation unit in its .hip_fatbin section; each bundle holds a gfx950 ELF whose
AMDGPU metadata note lists the kernels with their register / scratch / LDS sizes (llvm-readelf --notes prints it)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, data[pos + off:pos + off + size]
        pos += 24


def kernels(path):
    rows = {}
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for part in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
            part = ".agpr_count:" + part
            get = lambda k, part=part: re.search(r"\." + k + r":\s*(\S+)", part)
            name = get("name")
            if not name:
                continue
            sym = name.group(1).strip("'\"")
            rows[sym] = {"vgprs": int(get("vgpr_count").group(1)), "agprs": int(get("agpr_count").group(1)),
                         "scratch": int(get("private_segment_fixed_size").group(1)), "lds": int(get("group_segment_fixed_size").group(1))}
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", o)) for o in out[:len(names)]]


def short(dem):
    """template kernels: name<args>; plain kernels: name"""
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", dem)
    return m.group(1) if m else dem


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "real-time-reid-tracking_amd", "libreid_hip.so")
    rows = kernels(lib)
    names = sorted(rows)
    print("kernel,vgprs,agprs,scratch_bytes_per_lane,lds_bytes")
    for n, d in zip(names, demangle(names)):
        r = rows[n]
        print('"%s",%d,%d,%d,%d' % (short(d), r["vgprs"], r["agprs"], r["scratch"], r["lds"]))
    print("# %d kernels, %d with scratch, %d bytes" % (len(rows), sum(1 for r in rows.values() if r["scratch"]), os.path.getsize(lib)), file=sys.stderr)
