"""Model-level repeat test (GPU box): the same sample() call N times on identical noise must return identical bits, and match the
oracle on four probe rows -- at several batch sizes.  This is the test that exposed the pair-split hand-off (MDT_TF256_PAIR=1):
launches with DIFFERENT data alternate here, unlike in an op-level repeat.   python tools/repeat_determinism_probe.py [repeats]"""
import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import DEV, make_model
from helpers import oracle_cfg, synth_sd
from moleculediffusiontransformer_amd import NoiseSource
from moleculediffusiontransformer_amd.synth import synth_normal
from oracle import unet_oracle as O
reps=int(sys.argv[1]) if len(sys.argv)>1 else 25
m=make_model("cfg1"); m.kernel_choice="narrow"
tot_bad=0
for B,T in ((8,3),(64,3),(256,3),(1024,2),(8,6)):
    seq=synth_normal("d/seq",(B,12)); init=synth_normal("d/init",(B,16,64)); nz=[synth_normal(f"d/s{i}",(B,16,64)) for i in range(T-1)]
    rows=torch.tensor(sorted({0,1,B//2,B-1}))
    want=O.sample(synth_sd("cfg1"), oracle_cfg("cfg1"), seq[rows], init[rows], lambda i,x: nz[i][rows], T, 1.0, False)
    first=None; bad=0; worst=0.0
    for r in range(reps):
        out=m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
        e=float((out.cpu()[rows]-want).abs().max()); worst=max(worst,e)
        if first is None: first=out
        elif not torch.equal(out,first): bad+=1
    tot_bad+=bad
    print(f"B={B} T={T}: {bad} of {reps-1} repeats differ from the first, worst err vs oracle {worst:.2e}, status {m._engine.handoff_status()}", flush=True)
print("TOTAL_BAD", tot_bad)
