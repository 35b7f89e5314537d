"""Registers, scratch and occupancy of every kernel of the library, from hipcc's own remarks (no GPU needed):
    python tools/kernel_resources.py [out.csv]        # recompiles csrc/*.hip with -Rpass-analysis=kernel-resource-usage into /tmp
Prints name, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes, source file - one line per kernel."""
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-reid-tracking_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage"]


def product_sources():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    return re.search(r"^SRCS = (.*)$", mk, re.M).group(1).split(), re.search(r"^DBG_SRCS = (.*)$", mk, re.M).group(1).split()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", re.sub(r"\((Gemm|const|void|float|unsigned|int|_Float16|reid).*$", "", o))) for o in out]


def analyse(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", "/tmp/kr_%s.o" % src], capture_output=True, text=True, cwd=CSRC)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip(), "file": src}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def table(which="product"):
    prod, dbg = product_sources()
    srcs = prod if which == "product" else dbg
    with concurrent.futures.ThreadPoolExecutor(6) as ex:
        rows = [r for rs in ex.map(analyse, srcs) for r in rs]
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["kernel"] = n
    return rows


if __name__ == "__main__":
    rows = table()
    lines = ["kernel,vgprs,agprs,scratch_bytes_per_lane,occupancy_waves_per_simd,lds_bytes,file"]
    for r in sorted(rows, key=lambda r: (r["file"], r["kernel"])):
        lines.append('"%s",%s,%s,%s,%s,%s,%s' % (r["kernel"], r.get("VGPRs", ""), r.get("AGPRs", ""), r.get("ScratchSize [bytes/lane]", ""),
                                                 r.get("Occupancy [waves/SIMD]", ""), r.get("LDS Size [bytes/block]", ""), r["file"]))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    sys.stdout.write(text)
    print("# %d kernels, %d with scratch" % (len(rows), sum(1 for r in rows if int(r.get("ScratchSize [bytes/lane]", "0")) > 0)), file=sys.stderr)
