#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch of every kernel of libhip_ad_rgb.so, from the code object's metadata notes (no GPU needed).
Usage: python tools/kernel_resources.py [path to .so] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mitsuba3_amd", "libhip_ad_rgb.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + so], stdout=subprocess.DEVNULL) if False else None
    # the fat binary sits in the .hip_fatbin section: extract and unbundle the gfx950 code object
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
    targets = subprocess.check_output([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fat]).decode().split()
    tgt = next(t for t in targets if "gfx950" in t)
    co = os.path.join(d, "gfx950.co")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + tgt, "--input=" + fat, "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
rows = []
for blk in notes.split(".agpr_count:")[1:]:                 # one kernel descriptor per .agpr_count key (keys are sorted)
    m = re.search(r"\n\s+\.name:\s+(\S+)", blk[blk.find(".group_segment_fixed_size"):])
    if not m or ".vgpr_count" not in blk:
        continue
    name = subprocess.check_output(["c++filt", m.group(1)]).decode().strip()
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    rows.append((name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print("kernel resources of this build (hipcc -O3 --offload-arch=gfx950; from the code object's metadata)")
print("%-90s %5s %6s %10s %8s" % ("kernel", "VGPR", "SGPR", "LDS bytes", "scratch"))
for r in sorted(rows):
    print("%-90s %5d %6d %10d %8d" % r)
