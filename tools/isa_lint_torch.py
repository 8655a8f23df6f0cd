#!/usr/bin/env python3
"""The packed-fp32 operand form of profiles/r03_pkfma_corun.md in the OTHER libraries whose kernels run in the step
(beside the side stream's matrix kernels in the backward pass): libtorch_hip.so (ATen) and the fp32 hipBLASLt kernel
libraries (the SPP upsampling's dense products).  Their gfx950 code objects are zstd-compressed offload bundles ("CCOB"):
  libtorch_hip.so : llvm-objcopy --dump-section .hip_fatbin, split at the CCOB headers, clang-offload-bundler --unbundle
  hipblaslt/*.co  : clang-offload-bundler --unbundle directly
then llvm-objdump -d and the same scan as tools/isa_lint.py.  Mangled names of the kernels that contain the form are
written to <out>/risky_kernels.txt; with a rocprofv3 kernel trace (--trace t_kernel_trace.csv) the ATen kernels of the
traced run are checked against them by their distinctive name fragments.  CPU only, ~5 minutes, ~1 GB of temp space.

    python tools/isa_lint_torch.py --out /tmp/torchlint [--trace gpurun_out/r03d/trace/t_kernel_trace.csv]
"""
import argparse, csv, glob, os, re, struct, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
RISKY = re.compile(r"v_pk_(fma|mul|add)_f32.*op_sel:\[[01],1")
PACKED = re.compile(r"v_pk_(fma|mul|add)_f32")


def scan_bundle(path):
    co = path + ".co"
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}", f"--targets={TARGET}",
                        f"--output={co}"], capture_output=True)
    if r.returncode != 0 or not os.path.exists(co):
        return 0, []
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
    os.remove(co)
    kernel, total, hits = None, 0, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
        elif PACKED.search(line):
            total += 1
            if RISKY.search(line):
                hits.append(kernel)
    return total, hits


def match_trace(trace, risky):
    if trace.endswith(".txt"):  # one kernel name per line (tools/trace_kernels.sh keeps this instead of the 50 MB csv)
        names = sorted({l.strip() for l in open(trace) if "at::native" in l})
    else:
        names = sorted({r["Kernel_Name"] for r in csv.DictReader(open(trace)) if "at::native" in r["Kernel_Name"]})
    # mangled names spell functor / op names literally: compare on those fragments, and on the scalar type (a
    # float / double kernel of the trace is not its complex / Half / BFloat16 namesake)
    skip = ("anonymous", "namespace", "TensorIteratorBase", "operator")
    other = ("7complex", "4Half", "8BFloat16", "13Float8")
    bad = []
    for n in names:
        fr = [w for w in re.findall(r"[A-Za-z_][A-Za-z0-9_]{7,}", n) if w not in skip][:6]
        plain = "complex" not in n and "Half" not in n and "BFloat16" not in n
        cands = [k for k in risky if all(w in k for w in fr) and not (plain and any(o in k for o in other))]
        if cands:
            bad.append((n, cands[0]))
    print(f"{len(names)} ATen kernels in the trace; candidates among the kernels with the form (same op names, same scalar class): {len(bad)}")
    for n, k in bad:
        print("   ", n[:170], "\n        ~", k[:170])
    # exact comparison of the full demangled names (the fragment matcher above over-reports: `tanh` matches `atanh`, a
    # float multi_tensor_apply<multiplies> matches its complex / power_functor namesakes)
    dem = subprocess.run(["c++filt"], input="\n".join(risky), capture_output=True, text=True).stdout.splitlines()
    norm = lambda t: re.sub(r"\s+", "", t.replace(" [clone .kd]", "")).replace("(anonymousnamespace)", "{anon}")
    rset = {norm(d) for d in dem}
    allnames = [l.strip() for l in open(trace)] if trace.endswith(".txt") else names
    exact = [n for n in allnames if n and norm(n) in rset]
    print(f"exact demangled-name matches among ALL {len(allnames)} kernels of the trace: {len(exact)}")
    for n in exact:
        print("    EXACT", n[:200])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--trace")
    ap.add_argument("--reuse", action="store_true", help="take <out>/risky_kernels.txt from an earlier run")
    ap.add_argument("--lib", help="scan this shared library's gfx950 code objects instead of libtorch_hip.so (e.g. librccl.so)")
    a = ap.parse_args()
    if a.reuse:
        risky = [l.strip() for l in open(os.path.join(a.out, "risky_kernels.txt")) if l.strip()]
        for t in a.trace.split(","):
            print("==", t)
            match_trace(t, risky)
        return
    import torch
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    os.makedirs(a.out, exist_ok=True)
    fat = os.path.join(a.out, "fatbin.bin")
    target_lib = a.lib or os.path.join(lib, "libtorch_hip.so")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", target_lib,
                    os.path.join(a.out, "discard.so")], check=True)
    os.remove(os.path.join(a.out, "discard.so"))
    blob = open(fat, "rb").read()
    pieces = []
    for m in re.finditer(b"CCOB", blob):
        ver, _method, fsize, _usize = struct.unpack_from("<HHII", blob, m.start() + 4)
        if ver in (2, 3) and 0 < fsize <= len(blob) - m.start():
            p = os.path.join(a.out, f"torch_{len(pieces):03d}.ccob")
            open(p, "wb").write(blob[m.start():m.start() + fsize])
            pieces.append(p)
    os.remove(fat)
    blas = [] if a.lib else [f for f in glob.glob(os.path.join(lib, "hipblaslt", "library", "TensileLibrary_SS_SS_*gfx950.co"))]
    print(f"{len(pieces)} bundles in {os.path.basename(target_lib)}, {len(blas)} fp32 hipBLASLt libraries")
    with ThreadPoolExecutor(6) as ex:
        res_t = list(ex.map(scan_bundle, pieces))
        res_b = list(ex.map(scan_bundle, blas))
    for p in pieces:
        os.remove(p)
    risky = sorted({k for _, hits in res_t for k in hits})
    open(os.path.join(a.out, "risky_kernels.txt"), "w").write("\n".join(risky) + "\n")
    print(f"{os.path.basename(target_lib)}: {sum(t for t, _ in res_t)} packed-fp32 instructions, {sum(len(h) for _, h in res_t)} in the form, "
          f"in {len(risky)} kernels")
    print(f"hipBLASLt fp32 : {sum(t for t, _ in res_b)} packed-fp32 instructions, {sum(len(h) for _, h in res_b)} in the form")
    if a.trace:
        for t in a.trace.split(","):
            print("==", t)
            match_trace(t, risky)

if __name__ == "__main__":
    main()
