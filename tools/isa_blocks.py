#!/usr/bin/env python3
"""Static ISA audit of the kernels in libhip_ad_rgb.so (no GPU needed): instruction counts per kernel, and per basic block of one kernel -- the tool
behind DESIGN.md section 0 items 6h / 6i (register copies of the structurised control flow show up as blocks that are mostly v_mov).
Usage:  python tools/isa_blocks.py [--so PATH] [--kernels REGEX]            # one line per matching kernel
        python tools/isa_blocks.py --blocks 'k_trace_closest<false, false>'  # basic blocks of one kernel: index:instructions/VALU/v_mov
        python tools/isa_blocks.py --blocks NAME --show 27,34                # ... and the instructions of the listed blocks"""
import argparse
import collections
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(so):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so])
        targets = subprocess.check_output([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fat]).decode().split()
        tgt = next(t for t in targets if "gfx950" in t)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + tgt, "--input=" + fat, "--output=" + co])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", co]).decode()


def kernels(dis):
    out, cur = collections.OrderedDict(), None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        t = line.split("//")[0].split()
        if cur is not None and t and re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", t[0]):
            cur.append(" ".join(t))
    return out


def summary(ins):
    c = collections.Counter()
    for l in ins:
        op = l.split()[0]
        c["total"] += 1
        c["valu"] += op.startswith("v_")
        c["v_mov"] += op.startswith("v_mov") or op.startswith("v_accvgpr")
        c["salu"] += op.startswith("s_")
        c["branch"] += op.startswith("s_cbranch")
        c["vmem"] += op.startswith(("global_", "buffer_", "flat_"))
        c["lds"] += op.startswith("ds_")
    return c


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "mitsuba3_amd", "libhip_ad_rgb.so"))
    ap.add_argument("--kernels", default="k_trace_closest|k_resolve<|k_shade<0, 1u")
    ap.add_argument("--blocks", default=None, help="substring of one kernel's demangled name")
    ap.add_argument("--show", default="", help="comma-separated block indices to print")
    a = ap.parse_args()
    ks = kernels(disassemble(a.so))
    if a.blocks is None:
        for name, ins in ks.items():
            if re.search(a.kernels, name):
                c = summary(ins)
                print("%-72s total %5d valu %5d v_mov %4d salu %4d branch %3d vmem %3d lds %3d" % (re.sub(r"\(.*", "", name)[:72], c["total"], c["valu"], c["v_mov"], c["salu"], c["branch"], c["vmem"], c["lds"]))
    else:
        name = next(n for n in ks if a.blocks in n)
        blocks, cur = [], []
        for l in ks[name]:
            cur.append(l)
            op = l.split()[0]
            if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm":
                blocks.append(cur); cur = []
        if cur:
            blocks.append(cur)
        print(re.sub(r"\(.*", "", name))
        print(" ".join("%d:%d/%d/%d" % (i, len(b), sum(x.startswith("v_") for x in b), sum(x.startswith("v_mov") for x in b)) for i, b in enumerate(blocks)))
        for i in [int(x) for x in a.show.split(",") if x]:
            print("--- block", i)
            for l in blocks[i]:
                print("   ", l)
