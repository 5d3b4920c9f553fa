import torch, sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import ref, rnd
from moleculediffusiontransformer_amd import runtime as rt
W, A = rt.SP_WEIGHT, rt.SP_ACT
for (M, N, K, ln, res) in [(4096, 1024, 256, 1, 0), (4096, 256, 256, 1, 1), (16384, 1024, 128, 1, 0), (16384, 128, 128, 0, 0), (65536, 1024, 128, 1, 0)]:
    weights = torch.zeros(N * K + N + 2 * K, device="cuda")
    act = torch.zeros(M * K + M * N, device="cuda")
    op = rt.MdtOp(); op.kind = rt.OP_GEMM
    op.a, op.w, op.bias, op.out = ref(A, 0), ref(W, 0), ref(W, N * K), ref(A, K)
    i = op.i
    i[rt.G_T_STRIDE], i[rt.G_O_STRIDE] = 1, 1
    i[rt.G_R_OUT], i[rt.G_R_IN], i[rt.G_LDA], i[rt.G_CIN], i[rt.G_TAPS], i[rt.G_N], i[rt.G_LDC], i[rt.G_O_ROWS] = 1, 1, K, K, 1, N, N, 1
    if ln: i[rt.G_PRO] = rt.PRO_LAYERNORM
    if res:
        op.res = ref(A, K); i[rt.G_LDR] = N
    i[rt.G_WFMT] = 16
    op.f[0] = 1e-5
    b = rt.MdtBindings(); b.weights, b.act = rt.ptr(weights), rt.ptr(act)
    prog = rt.Program([op])
    for _ in range(5): prog.run(b, M)
    torch.cuda.synchronize()
    t = rt.EventTimer(1); t.start()
    for _ in range(50): prog.run(b, M)
    t.stop()
    print(M, N, K, ln, res, "%.2f us" % (t.collect()[0] * 1e3 / 50), flush=True)
