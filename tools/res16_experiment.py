"""CPU experiment (build container, oracle): what a bf16 RESIDUAL STREAM would cost the plain-bf16 mode of configs[4] in accuracy -- the
residual stream of the transformer blocks ("tf") or of transformer and ResNet blocks ("both") rounded to bf16 after every residual
add, everything else fp32 -- max-abs deviation of the final sample from the fp32 oracle on identical noise.  DESIGN.md section 9 (#3)
quotes the result.   python tools/res16_experiment.py"""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
from helpers import oracle_cfg, synth_sd
from moleculediffusiontransformer_amd.synth import synth_normal
from oracle import unet_oracle as O
torch.set_num_threads(8)
r = lambda t: t.bfloat16().float()
orig_tf, orig_rs = O._transformer1d, O._resnet
def tf16(sd, p, x, context, heads):
    x = F.group_norm(x, 32, sd[p + "to_in.0.weight"], sd[p + "to_in.0.bias"], eps=1e-6)
    x = F.conv1d(x, sd[p + "to_in.1.weight"], sd[p + "to_in.1.bias"])
    x = r(x.transpose(1, 2))
    i = 0
    while (p + f"blocks.{i}.attention.to_q.weight") in sd:
        bp = p + f"blocks.{i}."
        x = r(O._attention(sd, bp + "attention.", x, None, heads) + x)
        if (bp + "cross_attention.to_q.weight") in sd:
            x = r(O._attention(sd, bp + "cross_attention.", x, context, heads) + x)
        h = F.gelu(F.linear(x, sd[bp + "feed_forward.0.weight"], sd[bp + "feed_forward.0.bias"]))
        x = r(F.linear(h, sd[bp + "feed_forward.2.weight"], sd[bp + "feed_forward.2.bias"]) + x)
        i += 1
    x = x.transpose(1, 2)
    return F.conv1d(x, sd[p + "to_out.1.weight"], sd[p + "to_out.1.bias"])
def rs16(sd, p, x, mapping, groups):
    return r(orig_rs(sd, p, x, mapping, groups))
def run(case, B, T, shape, patched, which):
    sd, cfg = synth_sd(case), oracle_cfg(case)
    seq = synth_normal("e/seq", (B, 12)); init = synth_normal("e/init", (B,)+shape)
    nz = [synth_normal(f"e/s{i}", (B,)+shape) for i in range(T-1)]
    O._transformer1d = tf16 if (patched and which in ("tf","both")) else orig_tf
    O._resnet = rs16 if (patched and which in ("rs","both")) else orig_rs
    out = O.sample(sd, cfg, seq, init, lambda i, x: nz[i], T, 1.0, False)
    O._transformer1d, O._resnet = orig_tf, orig_rs
    return out
for case, B, T, shape in (("cfg1", 2, 16, (16,64)), ("cfg1", 2, 64, (16,64)), ("cfg5", 2, 16, (32,128))):
    t0=time.time(); a = run(case,B,T,shape,False,None)
    for which in ("tf","both"):
        b = run(case,B,T,shape,True,which)
        print(case, T, which, "max abs dev of a bf16 residual stream vs fp32:", float((a-b).abs().max()), "token agreement", float((a.argmax(1)==b.argmax(1)).float().mean()), f"{time.time()-t0:.0f}s", flush=True)
