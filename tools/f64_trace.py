"""Phase trace of the fp64 dynamic-attention kernel (csrc/f64.hip built with -DF64_TRACE: tools/ab_build.sh f64 trace -DF64_TRACE):
shader-clock cycles (s_memtime) between the phase boundaries of one workgroup in the middle of the grid, waves 0 and 3, and inside
the row select of wave 0 / 3 (load + reductions, start estimate, then per level: clear + atomics, scans, bin; exit).
    MDGAT_HIP_LIB=$PWD/ab/lib_trace.so python tools/f64_trace.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = 512
lib = _lib.load()
qkv = torch.randn(B, 2 * N, 3, 4, 32, dtype=torch.float64, device='cuda:0') * 1.3
names = ['pass A', 'barrier', 'select', 'barrier', 'pass B', 'write partials + barrier(s) + ties + combine']
for k in (128, 64):
    for _ in range(3):
        ops.attention_f64(qkv, N, N, False, topk=k)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    lib.mdgat_f64_trace_read.restype = C.c_int
    assert lib.mdgat_f64_trace_read(buf, 64) == 0
    for w in (0, 1):
        t = [buf[32 * w + i] for i in range(32)]
        print(f'k={k} wave {0 if w == 0 else 3}: ' + ', '.join(f'{names[i]} {t[i + 1] - t[i]}' for i in range(6)) + f' | total {t[6] - t[0]} cycles')
        if t[8]:
            sel = [('load 32 values + min / max / moments', t[2], t[8]), ('row reductions + start', t[8], t[9])]
            for lv in range(3):
                a, b_, c = t[10 + 3 * lv], t[11 + 3 * lv], t[12 + 3 * lv]
                if a >= t[2] and c >= a:
                    sel += [(f'level {lv + 1}: clear + atomics', a, b_), (f'level {lv + 1}: scans + bin', b_, c)]
            sel.append(('exit (last value / ties) + result', max(x for x in t[10:19] if x <= t[20]) if t[20] else t[3], t[20] or t[3]))
            print('      select: ' + ', '.join(f'{n} {e - s}' for n, s, e in sel))
