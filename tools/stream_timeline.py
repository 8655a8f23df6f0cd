"""Per-stream busy time, idle gaps and the longest kernels of each stream from a rocprofv3 kernel trace
(t_kernel_trace.csv), for the LAST step of a bench run.  usage: stream_timeline.py trace.csv [steps_in_trace]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id"))) for r in rows)
# steps are delimited by the Adam kernels: find the last multi_tensor kernel of each step
adam = [i for i, k in enumerate(ks) if "multi_tensor" in k[2] or "adam" in k[2].lower()]
ends = []
for i in adam:
    if not ends or ks[i][0] - ks[ends[-1]][1] > 20e6:
        ends.append(i)
    else:
        ends[-1] = i
print("optimizer groups found:", len(ends))
if len(ends) < 2:
    sys.exit(0)
lo, hi = ends[-2] + 1, ends[-1] + 1
step = ks[lo:hi]
t0, t1 = step[0][0], max(k[1] for k in step)
print("last step: %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
by = collections.defaultdict(list)
for k in step:
    by[k[3]].append(k)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
for sid, kk in by.items():
    busy = union([(k[0], k[1]) for k in kk])
    print("stream %s: %d kernels, busy %.2f ms, first +%.2f ms, last end +%.2f ms" % (sid, len(kk), busy / 1e6, (kk[0][0] - t0) / 1e6, (max(k[1] for k in kk) - t0) / 1e6))
# main stream = the one with most kernels; phases: before / during side activity
main = max(by, key=lambda s: len(by[s]))
side = [s for s in by if s != main]
if side:
    sk = sorted(k for s in side for k in by[s])
    s0, s1 = sk[0][0], max(k[1] for k in sk)
    print("side streams active from +%.2f to +%.2f ms (%.2f ms), busy %.2f ms" % ((s0 - t0) / 1e6, (s1 - t0) / 1e6, (s1 - s0) / 1e6, union([(k[0], k[1]) for k in sk]) / 1e6))
    mk = [k for k in by[main] if k[1] > s0 and k[0] < s1]
    print("main stream inside that window: busy %.2f ms; kernel time by family:" % (union([(k[0], k[1]) for k in mk]) / 1e6))
    fam = collections.Counter()
    for k in mk:
        fam[k[2].split("<")[0].split("(")[0][:40]] += k[1] - k[0]
    for n, v in fam.most_common(14):
        print("    %-42s %.2f ms" % (n, v / 1e6))
    gaps = 0; last = None
    for k in sorted(by[main]):
        if last is not None and k[0] > last: gaps += k[0] - last
        last = max(last or 0, k[1])
    print("main stream idle gaps over the step: %.2f ms" % (gaps / 1e6))

# ---- the step in 3-ms buckets: busy time of the main and the side stream, and who is running -----------------------------
if side:
    w = 3e6
    print("\n  t ms   main  side   main stream: top families (ms)  |  side stream")
    sk_all = [k for s in side for k in by[s]]
    t = t0
    while t < t1:
        def inside(kk):
            c = collections.Counter()
            for k in kk:
                o = min(k[1], t + w) - max(k[0], t)
                if o > 0:
                    c[k[2].replace("void ", "").split("(")[0][:34]] += o / 1e6
            return c
        fm, fs = inside(by[main]), inside(sk_all)
        print("%6.1f  %5.2f %5.2f   %s  |  %s" % ((t - t0) / 1e6, sum(fm.values()), sum(fs.values()),
              ", ".join("%s %.2f" % nv for nv in fm.most_common(3)), ", ".join("%s %.2f" % nv for nv in fs.most_common(2))))
        t += w
