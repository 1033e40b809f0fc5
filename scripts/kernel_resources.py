"""Registers / spills / scratch / LDS per kernel of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py learningbycheating_amd/csrc/conv_hdmap_256x128_320.hip [name filter]"""
import re, subprocess, sys, os

def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + root + "/include", "-I" + root + "/learningbycheating_amd/csrc",
           "-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("void ", "")
            cur = {"name": re.sub(r"\(IgemmArgs.*|\(WgradArgs.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("%-64s %5s %5s %6s %6s %7s %7s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "scratch", "LDS"))
    for r in rows:
        if flt in r["name"]:
            print("%-64s %5d %5d %6d %6d %7d %7d" % (r["name"][:64], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
                                                      r.get("ScratchSize", -1), r.get("LDS Size", -1)))

if __name__ == "__main__":
    main()
