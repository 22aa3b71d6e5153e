#!/usr/bin/env python3
"""Where the loads, the waits and the MFMAs of a kernel sit, in program order: is the software prefetch in the ISA?

For every kernel instance of the built objects whose name contains one of the arguments (default: all with a loop that
loads), the hot loop (tools/isa_mix.py's definition; --whole: the whole kernel, for fully unrolled ones such as k_rfb_tail)
as a string of
    LDn   n vector-memory loads in a row        Mn   n MFMAs in a row        (k)  s_waitcnt vmcnt(k)
    BAR   s_barrier                             br   a conditional branch
A queue of depth D shows as waits of (D - 1) x (loads per step) while loads are issued; `(0)` right behind `LD` is a load that is
waited for where it is requested (docs/EXPERIMENTS.md, round 5, "Prefetch queues hipcc had collapsed").

    python tools/ab/r5_queue_audit.py k_pw_mfma k_conv3x3_rows
    python tools/ab/r5_queue_audit.py --whole k_rfb_tail
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402


def listing(ins):
    seq = []
    for _, op, tx in ins:
        if op.startswith(("buffer_load", "global_load")):
            seq.append("LD")
        elif op.startswith("v_mfma"):
            seq.append("M")
        elif op == "s_waitcnt" and "vmcnt" in tx:
            seq.append("(" + re.search(r"vmcnt\((\d+)\)", tx).group(1) + ")")
        elif op == "s_barrier":
            seq.append("BAR")
        elif op.startswith("s_cbranch"):
            seq.append("br")
    out = " ".join(seq) + " "
    out = re.sub(r"(M )+", lambda m: "M%d " % (len(m.group(0)) // 2), out)
    out = re.sub(r"(LD )+", lambda m: "LD%d " % (len(m.group(0)) // 3), out)
    return out.strip()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    whole = "--whole" in sys.argv
    objs = sorted(glob.glob(os.path.join(ROOT, "infercam_onnx_amd", "csrc", "build", "*_kernels.o")))
    if not objs:
        sys.exit("no object files: build the library first")
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            for co in isa_mix.device_code_objects(obj, tmp):
                dis = subprocess.run([os.path.join(isa_mix.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
                kernels = isa_mix.parse(dis)
                names = isa_mix.demangle(list(kernels))
                for mangled, ins in kernels.items():
                    name = names[mangled]
                    if args and not any(a in name for a in args):
                        continue
                    if whole:
                        print("%s (whole kernel, %d instr)\n  %s" % (name, len(ins), listing(ins)))
                        continue
                    hl = isa_mix.hot_loop(ins, isa_mix.loops_of(ins))
                    if not hl:
                        continue
                    text = listing(ins[hl[0]:hl[1] + 1])
                    if "LD" in text or args:
                        print("%s (hot loop, %d instr)\n  %s" % (name, hl[1] - hl[0] + 1, text))


if __name__ == "__main__":
    main()
