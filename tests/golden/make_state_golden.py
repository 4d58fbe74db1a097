#!/usr/bin/env python3
"""Regenerates tests/golden/state_golden.json: md5 of the archives the UNMODIFIED reference CLI (`dsrc c -b1 -t1`) writes
for tests/cases.py::state_dependent_fastq -- a file whose blocks depend on the state one BlockCompressor carries from
block to block -- used by the multi-rank / multi-device tests (a sharded run must still write THIS archive).
Build container only; data only."""
import sys, hashlib, subprocess, json, os
sys.path.insert(0,'/root/repo')
from tests.cases import state_dependent_fastq
from tests._oracle import Oracle, Config, REF_BIN
data=state_dependent_fastq()
print(len(data))
open('/tmp/state.fastq','wb').write(data)
res={'in_sha256':hashlib.sha256(data).hexdigest(),'size_in':len(data),'archives':[]}
o=Oracle()
for flags in (['-d0','-q0'],['-d1','-q1','-c']):
    subprocess.check_call([REF_BIN,'c']+flags+['-b1','-t1','/tmp/state.fastq','/tmp/state.dsrc'])
    b=open('/tmp/state.dsrc','rb').read()
    res['archives'].append({'flags':flags,'buf_mb':1,'size':len(b),'md5':hashlib.md5(b).hexdigest()})
    print(flags,len(b))
cuts=o.cut_chunks(data,1<<20); print(len(cuts))
cfg=Config.from_levels(0,0)
chunks=[data[s:s+n] for s,n in cuts]
a=[b for b,_,_ in o.compress_blocks_state(cfg,chunks)]
fresh=[o.compress_block(cfg,c)[0] for c in chunks]
print('blocks differing when state is not carried:', sum(x!=y for x,y in zip(a,fresh)), 'of', len(a))
# a smaller variant (regions of ~0.5 MB, three 1 MiB chunks) for the emulator-backed CPU test
small=state_dependent_fastq(4500)
open('/tmp/state_s.fastq','wb').write(small)
subprocess.check_call([REF_BIN,'c','-d0','-q0','-b1','-t1','/tmp/state_s.fastq','/tmp/state_s.dsrc'])
b=open('/tmp/state_s.dsrc','rb').read()
cs=[small[s:s+n] for s,n in o.cut_chunks(small,1<<20)]
a=[x for x,_,_ in o.compress_blocks_state(cfg,cs)]; fresh=[o.compress_block(cfg,c)[0] for c in cs]
print('small:',len(small),len(cs),'chunks;',sum(x!=y for x,y in zip(a,fresh)),'state-dependent')
res['small']={'n_per_region':4500,'in_sha256':hashlib.sha256(small).hexdigest(),'flags':['-d0','-q0'],'buf_mb':1,'size':len(b),'md5':hashlib.md5(b).hexdigest()}
json.dump(res,open('/root/repo/tests/golden/state_golden.json','w'),indent=1)
