"""Writes mdgat_matcher_amd/csrc/exp2_tab256.hpp: 2^(j / 256) for j = 0 .. 255, correctly rounded to double (60-digit decimal
arithmetic, then Python's correctly rounded Decimal -> float conversion), as hexadecimal floating-point literals."""
import os
from decimal import Decimal, getcontext

getcontext().prec = 60
vals = [float(Decimal(2) ** (Decimal(j) / Decimal(256))) for j in range(256)]
lines = ['    ' + ', '.join(float.hex(v) for v in vals[i:i + 4]) + ',' for i in range(0, 256, 4)]
text = ('// 2^(j / 256), j = 0 .. 255, correctly rounded (generated with 60-digit decimal arithmetic: tools/gen_exp2_table.py).\n'
        '// The fp64 attention kernels (f64.hip: exp_fast) copy it to LDS; exp(x) = 2^q T[j] (1 + r + r^2/2 + r^3/6 + r^4/24), '
        "x = (256 q + j + r') ln2 / 256.\n#pragma once\n__device__ const double MDGAT_EXP2_TAB256[256] = {\n" + '\n'.join(lines) + '\n};\n')
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mdgat_matcher_amd', 'csrc', 'exp2_tab256.hpp')
open(path, 'w').write(text)
print('wrote', path)
