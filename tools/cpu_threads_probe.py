import os, sys, time, torch
sys.path.insert(0, '.')
from mdgat_matcher_amd import synth
from oracle import mdgat_oracle as O
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cgroup cpu.max', e)
sd = synth.make_state_dict(L=9, seed=0); cfg = synth.default_config(L=9)
data = synth.make_batch(1, 512, 512)
for t in (4, 8, 16, 32, 64):
    torch.set_num_threads(t)
    with torch.no_grad():
        O.mdgat_forward(sd, cfg, data)
        t0 = time.perf_counter(); O.mdgat_forward(sd, cfg, data); dt = time.perf_counter() - t0
    print('threads', t, 'sec/pair', round(dt, 3))
