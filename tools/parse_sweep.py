import re, collections, sys
rows=[]
for l in open(sys.argv[1]):
    m=re.match(r'(\S+)\s+(.*?)\s+calls=\s*(\d+) avg=\s*([\d.]+)us min=\s*([\d.]+)us\s+(\d+) GB/s', l)
    if not m or 'finalize' in l: continue
    tag,k,calls,avg,mn,gbs=m.groups()
    rows.append((tag,k.replace(' ',''),float(avg),float(mn),int(gbs)))
by=collections.OrderedDict()
for tag,k,avg,mn,gbs in rows:
    op=re.search(r'<(\d)',k).group(1)
    by.setdefault(tag,{})[op]=(k,avg,gbs)
merged=collections.OrderedDict()
for tag,d in by.items():
    base=re.sub(r'_[ab]$','',tag)
    merged.setdefault(base,[]).append(d)
for tag,ds in merged.items():
    out=[]
    for op in ('0','1','2'):
        vals=[d[op] for d in ds if op in d]
        if not vals: continue
        out.append(f"K{ {'0':1,'1':2,'2':4}[op]}:{vals[0][0][:27]:27s} "+"/".join(f"{v[2]:4d}" for v in vals))
    print(f"{tag:14s}", "   ".join(out))
