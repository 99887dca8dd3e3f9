import os, sys
sys.path.insert(0, "m3f.pytorch_amd")
import torch
from m3t import ops, _lib
dev = torch.device("cuda:0"); lib = ops.lib()
torch.manual_seed(1)
for (m, n, k) in [(1536, 1024, 9600), (1536, 768, 9600), (1536, 512, 9600), (1536, 1024, 960)]:
    A = torch.randn(k, m, device=dev); Bm = torch.randn(k, n, device=dev) * 0.05
    sl = ops.amax_slots(2, dev); ops.measure_amax([(A, sl.data_ptr()), (Bm, sl.data_ptr() + 8)])
    ws = ops.workspace(dev)
    Cr = torch.empty(m, n, device=dev); Cn = torch.empty(m, n, device=dev); C1 = torch.empty(m, n, device=dev)
    ops.sgemm(1, 0, m, n, k, A, 0, m, Bm, 0, n, Cr, 0, n, amax=(sl.data_ptr(), sl.data_ptr() + 8)); torch.cuda.synchronize()
    ops.sgemm(1, 0, m, n, k, A, 0, m, Bm, 0, n, C1, 0, n, use_ws=False, amax=(sl.data_ptr(), sl.data_ptr() + 8)); torch.cuda.synchronize()
    ref = A.double().t() @ Bm.double()
    print(m, n, k, "plan", ops.sgemm_plan(1, m, n, k, ws_bytes=ws.numel() * 4), "ref err %.3e  ref(no ws) err %.3e" % (float((Cr.double() - ref).abs().max()), float((C1.double() - ref).abs().max())))
    for sp in (1, 10):
        _lib.check(lib.m3t_sgemm_ring(1, 0, m, n, k, ops._p(A), m, ops._p(Bm), n, ops._p(Cn), n, None, 0, 0, 0, 0, 0, 0, ops._p(ws), ws.numel() * 4, sp,
                                      sl.data_ptr(), sl.data_ptr() + 8, 0, ops._stream()), "ring")
        torch.cuda.synchronize()
        print("  ring splits", sp, "err %.3e" % float((Cn.double() - ref).abs().max()), "== ref(ws):", torch.equal(Cr, Cn), " == ref(no ws):", torch.equal(C1, Cn))
