"""Sinkhorn kernel class time inside the forward (no Z, fused arg-maxes) at 0 / 100 / 200 iterations (GPU box):
the fixed part (load + absorb + epilogue arg-maxes + extract) versus the per-iteration part."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = synth.make_batch(B, 512, 512, dtype=torch.float32, device=dev)
inp = (d['keypoints0'], d['scores0'], d['descriptors0'], d['keypoints1'], d['scores1'], d['descriptors1'])
for S in (0, 1, 100, 200):
    net = MDGAT(synth.default_config(L=9, sinkhorn_iterations=S)).eval()
    net.load_state_dict(synth.make_state_dict(L=9, seed=0, dtype=torch.float32))
    with torch.no_grad():
        for _ in range(5): net._run(*inp)
        net.profile(dev, True)
        for _ in range(10): net._run(*inp)
        p = net.profile(dev, False)
    print(f'S={S}: sinkhorn class {p["sinkhorn"][0] / p["sinkhorn"][1] * 1e3:.1f} us per launch (memset + kernel + extract + fixup)')
