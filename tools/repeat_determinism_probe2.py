"""As repeat_determinism_probe.py for the other model families / call forms that reach the pair-split launches: guidance (doubled
batch), the notebook model, the AnalogDiffusionSparse-shaped U-Net.  python tools/repeat_determinism_probe2.py [repeats]"""
import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import DEV, make_model
from helpers import oracle_cfg, synth_sd
from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal, MODEL_CASES
from oracle import unet_oracle as O
reps=int(sys.argv[1]) if len(sys.argv)>1 else 12
tot=0
for case,B,T,cs in (("cfg1",256,3,7.5),("cfg1",512,2,2.0),("nb",64,3,1.0),("nb",256,2,2.0),("sparse",64,3,1.0),("sparse",512,2,1.0)):
    kw=MODEL_CASES[case][1]
    m=make_model(case); m.kernel_choice="narrow"
    L,pd,n=kw["max_length"],kw["pred_dim"],kw["context_embedding_max_length"]
    seq=synth_normal("d2/seq",(B,n)); init=synth_normal("d2/init",(B,pd,L)); nz=[synth_normal(f"d2/s{i}",(B,pd,L)) for i in range(T-1)]
    rows=torch.tensor(sorted({0,B-1}))
    want=O.sample(synth_sd(case), oracle_cfg(case), seq[rows], init[rows], lambda i,x: nz[i][rows], T, cs, False)
    first=None; bad=0; worst=0.0
    for r in range(reps):
        out=m.sample(seq, DEV, cond_scale=cs, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
        worst=max(worst,float((out.cpu()[rows]-want).abs().max()))
        if first is None: first=out
        elif not torch.equal(out,first): bad+=1
    split={op.i[rt.F_NSPLIT] for op in m._engine.c.programs["eval"] if op.kind==rt.OP_TF256}
    tot+=bad
    print(f"{case} B={B} T={T} cond_scale={cs}: pair-split ops {split}, dual {m._engine.has_dual}: {bad} of {reps-1} repeats differ, worst err vs oracle {worst:.2e}, status {m._engine.handoff_status()}", flush=True)
print("TOTAL_BAD",tot)
