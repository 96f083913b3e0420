import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from mdgat_matcher_amd import MDGAT, synth
dev = torch.device('cuda', 0)
L, S, n, B = 9, 100, 512, 64
cfg = synth.default_config(L=L, sinkhorn_iterations=S)
net = MDGAT(cfg).eval(); net.load_state_dict(synth.make_state_dict(L=L, seed=0, dtype=torch.float32))
for mode in ('expand', 'distinct'):
    if mode == 'expand':
        one = synth.make_batch(1, n, n, dtype=torch.float32, device=dev)
        inp = tuple(one[k].expand(B, *one[k].shape[1:]).contiguous() for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'))
    else:
        d = synth.make_batch(B, n, n, dtype=torch.float32, device=dev)
        inp = tuple(d[k] for k in ('keypoints0', 'scores0', 'descriptors0', 'keypoints1', 'scores1', 'descriptors1'))
    ts = []
    with torch.no_grad():
        for i in range(30):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net._run(*inp)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, ' '.join('%.2f' % t for t in ts))
