import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import DEV, make_model
from helpers import oracle_cfg, synth_sd
from moleculediffusiontransformer_amd import NoiseSource, runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal
from oracle import unet_oracle as O
def run(B,T,tag,choice="narrow"):
    seq=synth_normal("d/seq",(B,12)); init=synth_normal("d/init",(B,16,64)); nz=[synth_normal(f"d/s{i}",(B,16,64)) for i in range(T-1)]
    rows=torch.tensor(sorted({0,1,B//2,B-1}))
    want=O.sample(synth_sd("cfg1"), oracle_cfg("cfg1"), seq[rows], init[rows], lambda i,x: nz[i][rows], T, 1.0, False)
    m=make_model("cfg1"); m.kernel_choice=choice
    out=m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    out2=m.sample(seq, DEV, cond_scale=1.0, timesteps=T, noise=NoiseSource(init=init, steps=lambda i: nz[i]))
    print(f"{tag}: B={B} T={T} err {float((out.cpu()[rows]-want).abs().max()):.2e} repeat-diff {float((out-out2).abs().max()):.2e} status {m._engine.handoff_status()}", flush=True)
mode=sys.argv[1]
if mode=="sizes":
    for B in (8,16,32,64,256): run(B,3,"default")
    run(256,2,"default")
elif mode=="one":
    run(256,3,os.environ.get("TAG","x"))
