import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import make_model
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
m=make_model("cfg1"); m.kernel_choice="narrow"
eng=m.engine("cuda:0",12,B); eng.reserve(B)
emb=m._embed(synth_normal("prof/seq",(B,12)),"cuda:0")
eng.prepare_context(emb); eng.prepare_times(torch.tensor([0.1])); eng.select_time(0)
torch.manual_seed(0); eng.xin.normal_()
ops=eng.c.programs["eval"]; prog=eng.programs["eval"]
bind=eng._bind(xin=eng.xin,out=eng.pred)
tf=[i for i,op in enumerate(ops) if op.kind==rt.OP_TF256]
print("TF256 ops at", tf)
# state in front of every TF256 op from one clean pass
prog.run(bind,B,0,0,tf[0]); torch.cuda.synchronize()
for which in tf:
    op=ops[which]
    T,C=op.i[rt.F_T],256
    act0=eng.act.clone()
    outs=[]
    for rep in range(30):
        eng.act.copy_(act0)
        prog.run(bind,B,0,which,1); torch.cuda.synchronize()
        outs.append(eng.act[op.out.off*B: op.out.off*B + B*T*C].clone().view(B*T, C))
    ref=outs[0]
    nbad=0
    for r,o in enumerate(outs[1:],1):
        d=(o-ref).abs()
        if float(d.max())>0:
            nbad+=1
            rows=(d.max(dim=1).values>0).nonzero().flatten()
            cols=(d.max(dim=0).values>0).nonzero().flatten()
            if nbad<=3: print(f"  op {which} rep {r}: max diff {float(d.max()):.3e}; {len(rows)} rows differ (first {rows[:6].tolist()}, row blocks {sorted(set((rows//32).tolist()))[:8]}); {len(cols)} channels differ (min {int(cols.min())} max {int(cols.max())})")
    print(f"op {which}: blocks {op.i[rt.F_NBLOCKS]} cross {op.i[rt.F_CROSS]}: {nbad} of 29 repeats differ", flush=True)
    # continue the clean pass up to the next TF256 op
    eng.act.copy_(act0); prog.run(bind,B,0,which,1)
    nxt=[t for t in tf if t>which]
    if nxt: prog.run(bind,B,0,which+1,nxt[0]-which-1)
    torch.cuda.synchronize()
print("status",eng.handoff_status())
