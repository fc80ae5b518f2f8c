#!/usr/bin/env python3
"""Build container only: random closed-GOP option sets; the GOPs of a sequence coded one by one on the CPU harness (the product frame loop), joined, against the reference
application single run over the whole sequence.  usage: fuzz_concat.py [count] [seed]"""
import os, sys, subprocess, tempfile, random
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import _enc
from _e2e import make_yuv
from _libs import REF_APP
from xeve_amd import gop
rnd=random.Random(int(sys.argv[2]) if len(sys.argv)>2 else 7)
bad=0; n=int(sys.argv[1]) if len(sys.argv)>1 else 30
with tempfile.TemporaryDirectory() as d:
    for it in range(n):
        w,h=rnd.choice([(128,64),(136,72),(128,136)])
        bf=rnd.choice([0,1,3,7,15])
        keyint=rnd.choice([2,4,6,8,12,16]) if bf==0 else rnd.choice([2,3,4,5,8,9,12,16])
        total=rnd.choice([keyint+1, 2*keyint, 2*keyint+3, 3*keyint-1, 20])
        threads=rnd.choice([1,2,3])
        cli=["--preset",rnd.choice(["fast","medium"]),"-b",str(bf),"--closed-gop","-I",str(keyint),"-q",str(rnd.choice([27,32,40]))]
        yuv,evc=os.path.join(d,"a.yuv"),os.path.join(d,"a.evc")
        make_yuv(yuv,w,h,total,rnd.choice([13,5013]))
        data=open(yuv,'rb').read(); fb=w*h*3//2
        p=subprocess.run([REF_APP,"-i",yuv,"-w",str(w),"-h",str(h),"-z","30","--frames",str(total),"-m",str(threads),"-v","0","-o",evc]+cli,capture_output=True,text=True)
        if p.returncode!=0: print("ref refuses",total,cli); continue
        ref=open(evc,'rb').read()
        shards=gop.plan(total,keyint)
        by={}
        for s in shards: by.setdefault(s.frames,[]).append(s)
        outs={}
        try:
            cfg=_enc.config(w,h,cli,threads)
            for frames,group in by.items():
                res=_enc.encode_cpu(cfg,[data[s.seek*fb:(s.seek+frames)*fb] for s in group],frames)
                for s,o in zip(group,res): outs[s.gop]=o
        except Exception as e:
            print("refused",total,cli,str(e)[:80]); continue
        got=b"".join(outs[g] for g in range(len(shards)))
        unaligned = bf and keyint % (bf+1) != 0
        if got!=ref:
            bad+= 0 if unaligned else 1; print("DIFF (unaligned keyint: the reference does not join either)" if unaligned else "DIFF",w,h,total,threads,cli,len(got),len(ref))
print("done",n,"cases",bad,"different")
