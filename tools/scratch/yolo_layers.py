import csv, sys, re
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
from yolo_graph import Builder
b=Builder(64); hw_log=[]; orig=b.conv
def conv(x,cout,k=1,s=1,act=True,groups=1):
    y=orig(x,cout,k,s,act,groups); hw_log.append((x.c,cout,k,s,groups,x.hw,y.hw)); return y
b.conv=conv; b.build()
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[(r['Kernel_Name'],int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows]
isconv=lambda n: (('Conv' in n and 'gemm' in n) or 'depthwise' in n or 'conv3x3' in n or 'conv_window' in n or 'conv1x1' in n) and 'ConvTEpi' not in n and 'wperm' not in n and 'wfrag' not in n
seq=[(n,d) for n,d in names if isconv(n)]
per=len(hw_log); print(len(seq)/per)
last=seq[-per:]
out=[]
for (cin,cout,k,s,g,hin,hout),(n,d) in zip(hw_log,last):
    macs=64*cout*(cin//g)*k*k*hout*hout; byts=4*64*(cin*hin*hin+cout*hout*hout)
    bound=max(2*macs/157.3e12*1e6, byts/6e12*1e6)
    kind='window' if 'conv_window' in n else ('direct' if 'conv3x3_direct' in n else ('dw' if 'depthwise' in n else ('small' if 'small' in n else ('thin' if 'thin' in n else 'gemm'))))
    out.append((d/1e3,bound,cin,cout,k,s,g,hout,kind))
print("total conv us", sum(o[0] for o in out), "sum bounds", sum(o[1] for o in out))
key=(lambda r:-(r[0]-r[1])) if len(sys.argv)<3 else (lambda r:(r[4],-r[7],r[2],r[3]))
out.sort(key=key)
for r in out[:int(sys.argv[3]) if len(sys.argv)>3 else 30]: print("%7.1f %6.1f gap %6.1f | %3d->%3d k%d s%d g%3d @%3d %s"%(r[0],r[1],r[0]-r[1],*r[2:]))
