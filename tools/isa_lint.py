"""Scan the device code of libazhip.so for packed-fp32 VALU instructions in the operand form that returns wrong
values on gfx950 while a wave of another kernel issues MFMAs on the same SIMD (profiles/r03_pkfma_corun.md,
tools/probes/pkfma_corun.hip): v_pk_{fma,mul,add}_f32 with op_sel set for src1 (the HIGH register of the pair feeds
the low half).  src0 high selection (op_sel:[1,0,0]), a swapped src2 (op_sel:[0,0,1]) and all op_sel_hi forms were
measured clean.

    python tools/isa_lint.py [--src2] [path/to/libazhip.so]   -> prints offending kernels, exit code 1 if any
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def device_isa(lib_path):
    """yield (kernel_name, instruction_text) for every instruction of every gfx950 code object in the library"""
    with tempfile.TemporaryDirectory(prefix="azlint.") as d:
        local = os.path.join(d, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
        for name in sorted(os.listdir(d)):
            if "gfx950" not in name:
                continue
            out = subprocess.run([OBJDUMP, "-d", os.path.join(d, name)], check=True, capture_output=True, text=True).stdout
            kernel = None
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel = m.group(1)
                elif kernel and "\t" in line:
                    yield kernel, line.strip()


def risky_packed_ops(lib_path, src2_too=False):
    hits = []
    for kernel, ins in device_isa(lib_path):
        if not PACKED.search(ins):
            continue
        m = OPSEL.search(ins)
        if not m:
            continue
        bits = m.group(1).split(",")
        if bits[1] == "1" or (src2_too and len(bits) > 2 and bits[2] == "1"):
            hits.append((kernel, ins.split("//")[0].strip()))
    return hits


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    args = [a for a in sys.argv[1:] if a != "--src2"]
    lib = args[0] if args else os.path.join(here, "..", "activezero_amd", "lib", "libazhip.so")
    bad = risky_packed_ops(lib, src2_too="--src2" in sys.argv)
    for k, i in bad:
        print(k, "|", i)
    print(len(bad), "packed-fp32 instructions with a high-register selection on src1" + ("/src2" if "--src2" in sys.argv else ""))
    sys.exit(1 if bad else 0)
