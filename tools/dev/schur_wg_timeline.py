import re,sys,statistics as S
t=open(sys.argv[1]).read().split('SCHUR stamps')[0]
w=[(int(a),float(b),float(c),int(d)) for a,b,c,d in re.findall(r'\[(\d+) ([\d.]+) ([\d.]+) (\d+)\]',t)]
print(len(w), t.splitlines()[0][:60])
for s in (1,2,3,4):
    q=[x for x in w if x[3]==s]
    if q: print(' segments',s,'n',len(q),'exit med %.1f min %.1f max %.1f'%(S.median([x[2] for x in q]),min(x[2] for x in q),max(x[2] for x in q)))
