#!/usr/bin/env python3
"""usage: tools/loop_spills.py <file.hip> <mangled-name substring> -> scratch loads / stores inside every loop (backward branch) of the
kernel's ISA (device-only compile to assembly): where a kernel's spills sit - in its iteration loop or around it."""
import re
import subprocess
import sys
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
flags = ['-fno-slp-vectorize'] if 'attention.hip' in src else []
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', *flags, '--cuda-device-only', '-S',
                       os.path.join(R, 'mdgat_matcher_amd', 'csrc', src), '-o', '/tmp/ls.s'], stderr=subprocess.DEVNULL)
lines = open('/tmp/ls.s').read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
loops = set()
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.add((labels[m.group(1)], i))
print(f'{lines[start][:100]}  ({end - start} lines; scratch loads {sum("scratch_load" in x for x in body)}, stores {sum("scratch_store" in x for x in body)})')
for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:6]:
    seg = body[a:b + 1]
    print(f'  loop lines {a}-{b} ({b - a}): scratch loads {sum("scratch_load" in x for x in seg)}, stores {sum("scratch_store" in x for x in seg)}, '
          f'barriers {sum("s_barrier" in x for x in seg)}, global loads {sum("global_load" in x for x in seg)}')
