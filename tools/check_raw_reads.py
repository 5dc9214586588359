import re,sys
s=open('/tmp/gemm_bf16-hip-amdgcn-amd-amdhsa-gfx950.s').read()
lines=s.splitlines()
name='?'; in_asm=False; pend={}; bad=0; total=0
for idx,raw in enumerate(lines):
    t=raw.strip()
    m=re.match(r'^(_ZN\S+):', t)
    if m: name=m.group(1); pend={}; continue
    if 'ASMSTART' in t: in_asm=True; continue
    if 'ASMEND' in t: in_asm=False; continue
    l=t.split(';')[0].strip()
    if not l or l.endswith(':') or l.startswith('.'): continue
    m=re.match(r'ds_read_b64_tr_b16\s+v\[(\d+):(\d+)\]', l)
    if m and in_asm:
        total+=1
        for r in range(int(m.group(1)), int(m.group(2))+1): pend[r]=idx
        continue
    if l.startswith('s_waitcnt') and 'lgkmcnt(0)' in l:
        pend.clear(); continue
    if not pend: continue
    regs=set()
    for a,b in re.findall(r'v\[(\d+):(\d+)\]', l): regs.update(range(int(a), int(b)+1))
    for a in re.findall(r'\bv(\d+)\b', l): regs.add(int(a))
    hit=[r for r in regs if r in pend]
    if hit:
        bad+=1
        print(name[-80:], 'line', idx, ':', l[:90], ' <- read at line', pend[hit[0]])
        for r in hit: pend.pop(r,None)
print('hand-issued tr reads', total, '; uses of their destinations before an lgkmcnt(0):', bad)
