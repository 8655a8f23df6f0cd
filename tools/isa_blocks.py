import re,sys,subprocess
src=sys.argv[1]; kern=sys.argv[2]; extra=sys.argv[3:]
subprocess.run(["/opt/rocm/bin/hipcc","-O3","--offload-arch=gfx950","-std=c++17","-S","--cuda-device-only","-o","/tmp/k.s",src]+extra,check=True,stderr=subprocess.DEVNULL)
txt=open('/tmp/k.s').read().split('\n')
on=False; lines=[]
for l in txt:
    if l.startswith(kern+':'): on=True
    if on and lines and re.match(r'^[_a-zA-Z].*:', l) and not l.startswith(kern): break
    if on: lines.append(l)
open('/tmp/kern.s','w').write('\n'.join(lines))
blk='entry'; stats={}; order=[]
stats[blk]=dict(start=0,mfma=0,scr=0,ds=0,gl=0,st=0,n=0); order.append(blk)
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: blk=m.group(1); order.append(blk); stats[blk]=dict(start=i,mfma=0,scr=0,ds=0,gl=0,st=0,n=0)
    s=stats[blk]; t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    s['n']+=1
    if 'v_mfma' in t: s['mfma']+=1
    if 'scratch_' in t: s['scr']+=1
    if t.startswith('ds_'): s['ds']+=1
    if t.startswith('global_load'): s['gl']+=1
    if t.startswith('global_store'): s['st']+=1
for b in order:
    s=stats[b]
    if s['mfma'] or s['scr'] or s['st'] or s['gl']: print(b,s)
print(len(lines),'lines')
