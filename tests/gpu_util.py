"""Helpers for the -m gpu tests: build single ops, run them through the C ABI, compare with the CPU interpreter."""
import torch

from moleculediffusiontransformer_amd import runtime as rt
from moleculediffusiontransformer_amd.synth import synth_state_dict
from oracle.program_interp import Buffers, run_program

DEV = "cuda:0"


def ref(space, off=0):
    return rt.MdtRef(space, 0, off)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def run_both(ops, weights, act, shr, ext, B, n_shr=0):
    """Runs `ops` on the GPU (libmdt_hip) and on the CPU interpreter from identical buffers.
    Returns ((act, shr, ext) gpu->cpu, (act, shr, ext) cpu)."""
    cpu = Buffers(weights.clone(), act.clone(), shr.clone(), {k: v.clone().view(-1) for k, v in ext.items()})
    run_program(ops, cpu, B, n_shr)
    gw, ga, gs = weights.to(DEV), act.to(DEV), shr.to(DEV)
    ge = {k: v.to(DEV).contiguous() for k, v in ext.items()}
    b = rt.MdtBindings()
    b.weights, b.act, b.shr = rt.ptr(gw), rt.ptr(ga), rt.ptr(gs)
    for k, v in ge.items():
        b.ext[k] = rt.ptr(v)
    with torch.cuda.device(DEV):
        rt.Program(ops).run(b, B, n_shr)
        torch.cuda.synchronize()
    return (ga.cpu(), gs.cpu(), {k: v.cpu().view(-1) for k, v in ge.items()}), (cpu.act, cpu.shr, cpu.ext)


def make_model(case, cls_kw=None):
    from moleculediffusiontransformer_amd.synth import make_synth_model
    return make_synth_model(case, DEV)
