"""Diagnostic (DPENV_WS_SELFCHECK build): the real network evaluation in waves 4-7 beside a synthetic packed-fp32 recurrence with
a scalar shadow in waves 0-3 (pk_probe_kernel in dpenv_policy.hip).  Usage: DPENV_LIB=build/wsdiag/selfcheck_noslp.so python tools/ws_pk_probe.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from tests import helpers as H
from tests.test_gpu_policy import make_ac
from ml4ca_amd import _lib

lib = _lib.load()
env, _ = H.make_pair('final_cont', 1024)
ac = make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env)
lib.dpenv_debug_pk_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
blocks, iters = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sink = torch.zeros(blocks * 256, dtype=torch.float32, device='cuda:0')
for variant, vname in ((0, 'A = v_pk_fma_f32 op_sel:[0,1,0] (lo <- src1.hi)'), (1, 'A = v_pk_fma_f32 op_sel:[1,0,0] (lo <- src0.hi)'), (3, 'A = v_pk_fma_f32 op_sel:[0,0,1] copy (lo <- src2.hi) + plain v_pk_fma_f32')):
    for partner, pname in ((0, 'VALU loop'), (1, 'mlp_eval (MFMA + LDS + packing)'), (2, 'LDS reads only'), (3, 'MFMAs only'), (4, 'f16 packing VALU only (v_cvt_pk_f16_f32, v_pk_mul_f16, v_pk_max_f16)'), (5, 'v_permlane32_swap only'),
                           (6, 'MFMAs fed from LDS, results untouched'), (7, 'MFMA -> packing VALU -> MFMA, registers only'), (8, 'MFMA with srcC = 0 constant -> packing VALU'),
                           (9, 'MFMAs interleaved with VALU that reads no MFMA result'), (10, 'MFMA result read by v_fma_f32')):
        if variant > 0 and partner not in (0, 1, 7):
            continue
        out = torch.zeros(68, dtype=torch.int32, device='cuda:0')
        for rep in range(5):
            rc = lib.dpenv_debug_pk_probe(env._h, out.data_ptr(), sink.data_ptr(), blocks, iters, variant, partner)
            assert rc == 0, rc
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        lanes = {l: int(c) for l, c in enumerate(o[1:65]) if c}
        print('%s | partner: %s | events %d (acc.lo only %d, acc.hi only %d, other %d) lanes %s' % (vname, pname, o[0], o[65], o[66], o[67], lanes))
