"""Static instruction mix of the hot loops of a kernel source, from the gfx950 assembly hipcc emits (no GPU needed).
usage: python scripts/isa_mix.py learningbycheating_amd/csrc/conv_hdma.hip [kernel-name-substring] [min-mfma-per-block]
Prints, per kernel and per basic block with at least that many MFMAs, the count of MFMA / LDS reads by width / LDS-DMA /
global loads / VALU / SALU / s_waitcnt / s_barrier, the register and scratch figures, and the order of MFMAs (M), LDS reads (r)
and lgkmcnt waits (wN) inside the block -- how far ahead of its MFMA a fragment read is issued."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cat(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return op
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ds_write"
    if op.startswith("global_load") and "lds" in ins:
        return "lds_dma"
    if op.startswith(("global_load", "buffer_load")):
        return "global_load"
    if op.startswith(("global_store", "buffer_store")):
        return "global_store"
    if op.startswith("scratch"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    return "valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else "other")


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "learningbycheating_amd", "csrc"), "-x", "hip", "--cuda-device-only", "-S", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    kern, body = None, {}
    for l in lines:
        m = re.match(r"^(_Z\S+?):\s*; @", l)
        if m:
            kern = m.group(1)
            body[kern] = []
        elif kern:
            body[kern].append(l)
    for k, ls in body.items():
        if want not in k:
            continue
        res = {m.group(1): m.group(2) for l in ls for m in [re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", l)] if m}
        print("%s\n  %s" % (k, res))
        blocks, cur = {"entry": []}, "entry"
        for l in ls:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = m.group(1)
                blocks[cur] = []
                continue
            t = l.strip()
            if t and not t.startswith((";", ".")):
                blocks[cur].append(t)
        for name, ins in blocks.items():
            c = collections.Counter(cat(i) for i in ins)
            if c["mfma"] < min_mfma:
                continue
            seq = []
            for i in ins:
                ct = cat(i)
                if ct == "mfma":
                    seq.append("M")
                elif ct.startswith("ds_read"):
                    seq.append("r")
                elif ct == "s_waitcnt":
                    m = re.search(r"lgkmcnt\((\d+)\)", i)
                    seq.append("w" + m.group(1) if m else "v")
                elif ct == "s_barrier":
                    seq.append("B")
                elif ct == "lds_dma":
                    seq.append("d")
            print("  %s: %d instructions %s\n    %s" % (name, len(ins), dict(c), "".join(seq)))


if __name__ == "__main__":
    main()
