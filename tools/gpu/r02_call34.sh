#!/bin/bash
mkdir -p gpurun_out
cd /tmp && mkdir -p cnt && python - <<'PY' > /root/repo/gpurun_out/r02_call34.log 2>&1
import sys,subprocess,os,time,hashlib,json
sys.path.insert(0,'/root/repo/tests')
from _e2e import *
from _libs import ORACLE_SO
G=json.load(open('/root/repo/tests/golden/e2e_v1.json'))
name="cfg3_1080p_ra_medium"
w,h,n,seed,extra=REAL_CASES[name]
make_yuv('/tmp/cnt/b.yuv',w,h,n,seed)
cmd=[REF_APP,"-i",'/tmp/cnt/b.yuv',"-w",str(w),"-h",str(h),"-z","30","--frames",str(n),"-m","1","-v","0","-o",'/tmp/cnt/b.evc']+list(extra)
env=dict(os.environ,LD_PRELOAD=SHIM,XEVE_HIP_LIB=HIP_LIB,XEVE_HIP_SHIM_TABLES="0",XEVE_HIP_SHIM_TREE="2",XEVE_SHIM_TREE_CHECK=ORACLE_SO)
t=time.time()
p=subprocess.run(cmd,env=env,capture_output=True,text=True)
d=open('/tmp/cnt/b.evc','rb').read()
print(name,p.returncode,hashlib.md5(d).hexdigest()==G[name]["md5"],round(time.time()-t,1))
print(p.stderr[-4000:])
PY
cat /root/repo/gpurun_out/r02_call34.log | tail -30
