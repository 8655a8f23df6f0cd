"""Static check of one kernel's ISA (hipcc -S output): every VALU / store source register must not be the
destination of an LDS read or global load that the preceding s_waitcnt instructions have not retired
(in-order counters: lgkmcnt for ds_*, vmcnt for global/buffer loads).  Straight-line approximation: the
outstanding queues are carried across labels and branches in program order.
usage: isa_wait_check.py file.s kernel_name_substring"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and sys.argv[2] in l and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m: return {int(m.group(1))}
    return set()
lgkm, vm = [], []   # lists of (dest regs, line)
bad = 0
for i in range(start + 1, end):
    l = txt[i].split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."): continue
    op, _, rest = l.partition(" ")
    toks = [t.strip() for t in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", rest)
        if m: lgkm = lgkm[len(lgkm) - int(m.group(1)):] if int(m.group(1)) else []
        m = re.search(r"vmcnt\((\d+)\)", rest)
        if m: vm = vm[len(vm) - int(m.group(1)):] if int(m.group(1)) else []
        continue
    srcs = set()
    dst = set()
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        dst = regs(toks[0]); 
        for t in toks[1:]: srcs |= regs(t.split()[0])
    elif op.startswith("ds_write") or op.startswith("ds_add") or op.startswith("ds_store"):
        for t in toks: srcs |= regs(t.split()[0])
    elif op.startswith("global_load") or op.startswith("buffer_load"):
        dst = regs(toks[0])
        for t in toks[1:]: srcs |= regs(t.split()[0])
    elif op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("buffer_store"):
        for t in toks: srcs |= regs(t.split()[0])
    elif op.startswith("v_"):
        dst = regs(toks[0]) if toks else set()
        for t in toks[1:]: srcs |= regs(t.split()[0])
        if op.startswith("v_cmp") or op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
            srcs |= regs(toks[0]) if toks else set(); 
    for q, nm in ((lgkm, "lgkm"), (vm, "vm")):
        for d, ln in q:
            if d & (srcs | dst):
                print(f"line {i - start}: {l}\n    touches v{sorted(d & (srcs | dst))} still outstanding ({nm}) from line {ln - start}: {txt[ln].strip()}")
                bad += 1
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        lgkm.append((dst, i))
    elif op.startswith("ds_"):
        lgkm.append((set(), i))
    elif op.startswith("global_load") or op.startswith("buffer_load"):
        vm.append((dst, i))
    elif op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("buffer_store"):
        vm.append((set(), i))
    elif op.startswith("s_load") or op.startswith("s_buffer_load"):
        lgkm.append((set(), i))
print("suspicious:", bad)
