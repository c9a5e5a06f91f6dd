#!/usr/bin/env python3
"""What every kernel of libgmsplat.so asks of a CU, from the compiler's own remarks (no GPU needed).

Compiles each csrc/*.hip to a scratch object with the Makefile's flags plus `-Rpass-analysis=kernel-resource-usage` and prints, per
kernel: VGPRs / AGPRs / SGPRs, scratch bytes per lane, static LDS bytes per block, the compiler's occupancy estimate (waves per
SIMD; it accounts for registers AND the static LDS of the kernel's launch bounds) and floor(160 KB / static LDS) blocks per CU
(160 KB of LDS per CU: MI355X_MICROARCH.md).  A 256-thread block puts one wave on each of the CU's four SIMDs, so for such
kernels "waves per SIMD" is "blocks per CU" (31 876 B of LDS -> 5 blocks: the occupancy step DESIGN.md section 7 measures).

    python tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussian-mesh-splatting_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import srchash  # noqa: E402

LDS_PER_CU = 160 * 1024
FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vspill", "SGPRs Spill": "sspill"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*$", "", o).replace("void ", "").replace("gms::", "") for o in out[:len(names)]]


def remarks(src, tmp):
    # the Makefile's per-file flags: the files it lists under NOSLP are compiled without the SLP vectoriser
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^NOSLP\s*:=\s*(.*)$", mk, re.M)
    noslp = ["-fno-slp-vectorize"] if m and os.path.basename(src) in m.group(1).split() else []
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + noslp + [
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, "o.o")]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    kernels, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            kernels.append(cur)
        elif key in FIELDS and cur is not None:
            cur[FIELDS[key]] = int(val)
    return kernels


def main():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "test_hooks.hip")
    print("kernel resources of libgmsplat.so (gfx950, -O3; kernel source hash %s)" % srchash.kernel_source_hash())
    print("occ = the compiler's occupancy estimate in waves per SIMD (registers and static LDS); blk/CU LDS = floor(160 KB / static LDS of")
    print("a block).  A block of 256 threads puts one wave on each SIMD: occ is then the number of resident blocks per CU.")
    print()
    print("%-20s %-58s %5s %5s %5s %8s %8s %4s %10s" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS B", "occ", "blk/CU LDS"))
    with tempfile.TemporaryDirectory() as tmp:
        for f in srcs:
            ks = remarks(f, tmp)
            names = demangle([k["name"] for k in ks])
            for k, n in sorted(zip(ks, names), key=lambda t: t[1]):
                lds = k.get("lds", 0)
                blk = "-" if lds == 0 else str(LDS_PER_CU // lds)
                spill = "" if not (k.get("vspill") or k.get("sspill")) else "  SPILLS v%d s%d" % (k.get("vspill", 0), k.get("sspill", 0))
                print("%-20s %-58s %5d %5d %5d %8d %8d %4d %10s%s" % (f, n[:58], k.get("vgpr", 0), k.get("agpr", 0), k.get("sgpr", 0), k.get("scratch", 0), lds,
                                                                     k.get("occ", 0), blk, spill))


if __name__ == "__main__":
    main()
