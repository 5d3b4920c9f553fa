import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from gpu_util import make_model
from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.synth import synth_normal
from oracle.program_interp import Buffers, run_program
B=int(sys.argv[1]) if len(sys.argv)>1 else 8
which=int(sys.argv[2]) if len(sys.argv)>2 else 6
m=make_model("cfg1"); m.kernel_choice="narrow"
eng=m.engine("cuda:0",12,B); eng.reserve(B)
emb=m._embed(synth_normal("prof/seq",(B,12)),"cuda:0")
eng.prepare_context(emb); eng.prepare_times(torch.tensor([0.1])); eng.select_time(0)
torch.manual_seed(0); eng.xin.normal_()
ops=eng.c.programs["eval"]; prog=eng.programs["eval"]
bind=eng._bind(xin=eng.xin,out=eng.pred)
for i in range(which): prog.run(bind,B,0,i,1)
torch.cuda.synchronize()
act0=eng.act.clone()
op=ops[which]
print("op",which,rt.OP_NAMES[op.kind],"nsplit",op.i[rt.F_NSPLIT],"blocks",op.i[rt.F_NBLOCKS],"cross",op.i[rt.F_CROSS])
cpu=Buffers(eng.c.weights.clone(), act0.cpu().clone(), eng.shr.cpu().clone(), {0:eng.xin.cpu().view(-1).clone(),2:eng.pred.cpu().view(-1).clone()})
run_program([op],cpu,B,0)
lib=rt.load_library()
x_ref=op.a
bad=0
for rep in range(16):
    lib.mdt_set_tuning(b"pair_stride", 1 if (rep//2)%2 else 8)
    act=act0.clone()
    xin=cpu0=None
    off=x_ref.off*B; n=B*4*256
    act[off:off+n]*= (1.0+0.1*rep)            # a different input every launch
    cpu=Buffers(eng.c.weights.clone(), act.cpu().clone(), eng.shr.cpu().clone(), {0:eng.xin.cpu().view(-1).clone(),2:eng.pred.cpu().view(-1).clone()})
    run_program([op],cpu,B,0)
    eng.act.copy_(act)
    if os.environ.get("ZERO") and rep in (2,3,9): eng.xbuf.zero_()          # plain stores into the hand-off blocks between launches
    if os.environ.get("ZERO") and rep == 11: eng.xflags[64:].add_(0)        # plain read-modify-write of the flag words (same values)
    prog.run(bind,B,0,which,1); torch.cuda.synchronize()
    err=float((eng.act.cpu()-cpu.act).abs().max()); bad+= err>1e-3
    print("rep",rep,"stride",1 if (rep//2)%2 else 8,"vs cpu",err,"flags",eng.xflags[64:64+64:32].tolist(),flush=True)
lib.mdt_set_tuning(b"pair_stride",0)
print("BAD",bad)
