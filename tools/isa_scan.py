"""Static look at one kernel's gfx950 ISA (no GPU): instruction mix, scratch traffic, where the MFMAs sit.

    python tools/isa_scan.py scan_amd/csrc/conv_fwd.hip 'conv_split_kernel<3, 256, 16, 512, 3, 1, true>'
"""
import re
import subprocess
import sys


def main():
    src, want = sys.argv[1], sys.argv[2].replace(" ", "")
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value",
                          "--cuda-device-only", "-S", src, "-o", "-"] + sys.argv[3:], capture_output=True, text=True).stdout
    blocks = re.split(r"\n(?=_Z[^\n]*:\s+; @)", asm)
    for b in blocks:
        mangled = b.split(":")[0]
        if not mangled.startswith("_Z"):
            continue
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace(" ", "")
        if name != want:
            continue
        lines = [l.strip() for l in b.split("\n")]
        ins = [l for l in lines if l and not l.startswith((";", ".", "_Z")) and not l.endswith(":")]
        def cnt(p):
            return sum(1 for l in ins if re.match(p, l))
        print(name, "instructions:", len(ins))
        for label, pat in (("v_mfma", r"v_mfma"), ("ds_read", r"ds_read|ds_load"), ("ds_write", r"ds_write|ds_store"),
                           ("buffer_load", r"buffer_load"), ("global/flat", r"global_|flat_"),
                           ("scratch", r"scratch_"), ("s_barrier", r"s_barrier"), ("s_waitcnt", r"s_waitcnt"),
                           ("v_cvt_pk_bf16", r"v_cvt_pk_bf16"), ("v_readfirstlane", r"v_readfirstlane"),
                           ("s_cbranch", r"s_cbranch")):
            print("  %-16s %5d" % (label, cnt(pat)))
        idx = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        sc = [i for i, l in enumerate(ins) if l.startswith("scratch_")]
        if idx:
            print("  mfma span: instruction %d .. %d; scratch inside the span: %d" %
                  (idx[0], idx[-1], sum(1 for i in sc if idx[0] <= i <= idx[-1])))
        return
    print("kernel not found:", want)


if __name__ == "__main__":
    main()
