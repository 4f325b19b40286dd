#!/usr/bin/env python3
"""How well-determined are the golden numbers?  Runs the oracle's two training steps in float64 (or float32) and
prints the deviation from the float32 reference fixture: after an Adam update (sign-like first steps) rounding
noise is amplified, and the step-1 generator gradient norm of the reference itself is only determined to a few
percent.  The tolerances of tests/test_step_gpu.py for step 1 come from here (profiles/r02_conditioning.txt).

    python tools/conditioning.py {model|bert|mix} {f64|f32}
"""
import sys, torch, numpy as np
sys.path[:0]=['/root/repo/tests','/root/repo','/root/repo/sba-gan_amd']
from helpers import *
from oracle import fill, sbagan_oracle as O
torch.set_num_threads(8)
variant=sys.argv[1]; dt=torch.float64 if sys.argv[2]=='f64' else torch.float32
if dt==torch.float64: torch.set_default_dtype(torch.float64)
Gs=load_golden('/root/repo/tests/golden','step_full_%s_b4.npz'%variant)
d=FULL; Bs=4; tag=500
x=make_inputs(d,Bs,18,lmax=18,tag=tag)
v='model' if variant=='model' else 'bert'
cv=lambda t: t.to(dt) if t.is_floating_point() else t
PG={k:cv(t) for k,t in fill.fill_state_dict(g_shapes(d,3,v)).items()}
PDs=[{k:cv(t) for k,t in fill.fill_state_dict(d_shapes(d,i),salt=i).items()} for i in range(3)]
st=O.OracleState(PG,PDs)
enc=fill.StandInImageEncoder(d['nef'],dtype=dt)
if dt==torch.float64:
    for a in ('wr','wc','bc','w_region','w_code','b_code'):
        if hasattr(enc,a): setattr(enc,a,getattr(enc,a).double())
for step in range(2):
    noise=fill.unit((2,Bs,d['nz']) if variant=='mix' else (Bs,d['nz']),tag+50+step).to(dt)
    eps=torch.from_numpy(Gs['step%d/eps'%step]).to(dt)
    o=O.train_step(st,[i.to(dt) for i in x['imgs']],x['sent'].to(dt),x['words'].to(dt),x['mask'],x['cap_lens'],x['class_ids'],noise,eps,enc,SMOOTH,variant=variant)
    for k in ['errD0','errD1','errD2','errG_total','gnormD2','gnormG']:
        ref=float(Gs['step%d/%s'%(step,k)]); print(step,k,'%.6g'%o[k],'%.6g'%ref,'rel %.2e'%(abs(o[k]-ref)/abs(ref)))
