#!/usr/bin/env python3
"""Regenerates tests/golden/config_golden.json: BASELINE configs 1/2 at full size -- the 1 M-read synthetic Illumina set
(dsrc_amd/synth.py) compressed by the UNMODIFIED reference CLI (oracle/_ref/dsrc_ref c ... -t1): archive size + md5
at -d0 -q0 (configs 1/2), -d3 -q2 (config 3's level on the same set) and -d3 -q2 -c.  Build container only; data only."""
import sys, hashlib, subprocess, json, time
sys.path.insert(0,'/root/repo')
from dsrc_amd import synth
t=time.time()
with open('/tmp/cfg/ill1m.fastq','wb') as f:
    for lo in range(1, 1000001, 50000):
        f.write(synth.illumina_fastq(50000, first=lo))
print('gen', time.time()-t)
res={}
for name, flags in (('d0q0',['-d0','-q0']),('d3q2',['-d3','-q2']),('d3q2c',['-d3','-q2','-c'])):
    t=time.time()
    subprocess.check_call(['/root/repo/oracle/_ref/dsrc_ref','c']+flags+['-t1','/tmp/cfg/ill1m.fastq','/tmp/cfg/%s.dsrc'%name])
    b=open('/tmp/cfg/%s.dsrc'%name,'rb').read()
    res[name]={'flags':flags,'size':len(b),'md5':hashlib.md5(b).hexdigest()}
    print(name, time.time()-t, res[name])
h=hashlib.md5(); n=0
with open('/tmp/cfg/ill1m.fastq','rb') as f:
    while True:
        b=f.read(1<<24)
        if not b: break
        h.update(b); n+=len(b)
res['input']={'reads':1000000,'size':n,'md5':h.hexdigest()}
import os
json.dump(res,open(os.path.join(os.path.dirname(os.path.abspath(__file__)),'config_golden.json'),'w'),indent=1)
