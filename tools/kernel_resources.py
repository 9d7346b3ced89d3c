"""Registers / scratch / occupancy of every kernel in one .hip file, as the compiler reports them (no GPU needed).

    python tools/kernel_resources.py scan_amd/csrc/conv_fwd.hip [filter]
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(.*", "", l) for l in out.stdout.splitlines()]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    recs, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|"
                      r"VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"name": v}
            recs.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    names = demangle([r["name"] for r in recs])
    print("%-78s %5s %5s %7s %5s %5s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "SGPR"))
    for r, n in zip(recs, names):
        n = n.replace("void ", "")
        if flt in n:
            print("%-78s %5s %5s %7s %5s %5s" % (n[:78], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                                  r.get("Occupancy"), r.get("SGPRs")))


if __name__ == "__main__":
    main()
