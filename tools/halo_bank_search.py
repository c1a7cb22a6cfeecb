import numpy as np, itertools, collections
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS+= [[l+32 for l in g] for g in GROUPS]
def evalf(KS,CINB,MF,f,pitch_pad,TD=4,TH=8,TW=8):
    NVV=CINB//16
    HH,HW=TH+KS-1,TW+KS-1
    PW=((HW+pitch_pad-1)//pitch_pad)*pitch_pad
    G=max(1,NVV//2 if MF==32 else NVV//4)
    tot=0;n=0;worst=0
    for kd,kh,kw in itertools.product(range(KS),repeat=3):
        for i in range(64//MF):
            for g in range(G):
                for grp in GROUPS:
                    slots=collections.Counter()
                    for lane in grp:
                        lvb=(lane>>5) if MF==32 else (lane>>4)
                        lv=lvb+(2*g if MF==32 else 4*g)
                        r=i*MF+(lane&(MF-1)); tw=r%TW; th=(r//TW)%TH; td=r//(TW*TH)
                        hd,hh,hw=td+kd,th+kh,tw+kw
                        hv=(hd*HH+hh)*PW+hw
                        pv=lv^(f(hd,hh,hw,hv)%NVV)
                        slots[((hv*CINB+pv*16)//16)%16]+=1
                    c=max(slots.values()); tot+=c;n+=1;worst=max(worst,c)
    return tot/n,worst,PW
import sys
for (KS,CINB,MF) in [(3,64,32),(7,64,16),(3,128,32),(3,32,32),(3,64,32)]:
    best=[]
    for pad in (1,2,4):
        for a,b,c,sh in itertools.product(range(4),range(4),range(4),(0,1,2,3)):
            f=lambda hd,hh,hw,hv,a=a,b=b,c=c,sh=sh: a*hh+b*hd+((hw+c*hh)>>sh)
            avg,w,PW=evalf(KS,CINB,MF,f,pad)
            best.append((avg,w,pad,PW,a,b,c,sh))
        for sh in (0,1,2,3,4):
            f=lambda hd,hh,hw,hv,sh=sh: hv>>sh
            avg,w,PW=evalf(KS,CINB,MF,f,pad); best.append((avg,w,pad,PW,'hv>>',sh,0,0))
    best.sort(key=lambda t:(t[0],t[3]))
    print((KS,CINB,MF),best[:4])
