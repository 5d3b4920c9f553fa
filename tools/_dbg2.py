import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import make_model
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal
B=int(sys.argv[1]) if len(sys.argv)>1 else 8
m=make_model("cfg1"); m.kernel_choice="narrow"
eng=m.engine("cuda:0",12,B); eng.reserve(B)
emb=m._embed(synth_normal("prof/seq",(B,12)),"cuda:0")
eng.prepare_context(emb); eng.prepare_times(torch.tensor([0.1])); eng.select_time(0)
torch.manual_seed(0); eng.xin.normal_()
ops=eng.c.programs["eval"]; prog=eng.programs["eval"]
bind=eng._bind(xin=eng.xin,out=eng.pred)
act0=eng.act.clone()
def run_all():
    eng.act.copy_(act0); eng.pred.zero_()
    snaps=[]
    for i in range(len(ops)):
        prog.run(bind,B,0,i,1)
        torch.cuda.synchronize()
        snaps.append((eng.act.clone(), eng.pred.clone()))
    return snaps
a=run_all(); b=run_all(); c=run_all()
for i,op in enumerate(ops):
    d1=max(float((a[i][0]-b[i][0]).abs().max()), float((a[i][1]-b[i][1]).abs().max()))
    d2=max(float((a[i][0]-c[i][0]).abs().max()), float((a[i][1]-c[i][1]).abs().max()))
    nan=bool(torch.isnan(a[i][0]).any())
    tag=rt.OP_NAMES[op.kind]+(f"/split{op.i[rt.F_NSPLIT]} nblk{op.i[rt.F_NBLOCKS]} cross{op.i[rt.F_CROSS]}" if op.kind==rt.OP_TF256 else "")
    if d1>0 or d2>0 or op.kind==rt.OP_TF256: print(i,tag,"run1-run2",d1,"run1-run3",d2,"nan-in-arena",nan,flush=True)
print("status",eng.handoff_status())
