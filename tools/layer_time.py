import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth
dev = torch.device('cuda', 0)
cfg = synth.default_config(L=9, sinkhorn_iterations=100)
net = MDGAT(cfg).eval(); net.load_state_dict(synth.make_state_dict(L=9, seed=0, dtype=torch.float32))
d = synth.make_batch(64, 512, 512, dtype=torch.float32, device=dev)
inp = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
with torch.no_grad():
    for _ in range(3): net._run(*inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net._run(*inp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(os.environ.get('MDGAT_HIP_LIB', 'default'), 'ms/step %.3f' % (dt * 1e3), 'pairs/s %.0f' % (64 / dt))
