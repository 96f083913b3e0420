"""Several forwards in flight on several streams at shapes whose Sinkhorn pair spans many CUs (GPU box):
    python tools/stress_streams.py [streams] [rounds] [N] [B]          (MDGAT_STRESS_ARITH=fp64: the exact mode)
N = 2048: 64 workgroups per pair, spread over all XCDs - the case in which concurrent cluster launches can leave every CU
with a workgroup whose partners cannot be dispatched.  Every result must equal the serial result (bit for bit, or - when a
launch fell back to the streaming kernel - to 1e-4 on Z with at most a handful of near-tie arg-maxes apart)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import MDGAT, synth
ns, rounds, N, B = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 4), (2, 3), (3, 2048), (4, 2)))
dev = 'cuda:0'
L = 9          # (synthetic weights are scaled for L = 9: shallower nets produce scores beyond the Sinkhorn range guard)
arith = os.environ.get('MDGAT_STRESS_ARITH', 'fp32')      # fp64: the reference-exact mode
net = MDGAT(synth.default_config(L=L, sinkhorn_iterations=50, arithmetic=arith))
if arith == 'fp64':
    net = net.double()
net.load_state_dict(synth.make_state_dict(L=L, seed=0))
net = net.eval().to(dev)
batches = [synth.make_batch(B, N, N, first_pair=10 * i, device=dev, dtype=torch.float64 if arith == 'fp64' else torch.float32) for i in range(ns)]
args = [(d['keypoints0'], d['descriptors0'], d['keypoints1'], d['descriptors1'], d['scores0'], d['scores1']) for d in batches]
serial = [net.match(*a, return_scores=True) for a in args]
torch.cuda.synchronize()
assert not net.check(dev)['sinkhorn_fallback']
streams = [torch.cuda.Stream(dev) for _ in range(ns)]
fb = 0
for r in range(rounds):
    t0 = time.time()
    outs = [None] * ns
    for rep in range(3):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i] = net.match(*args[i], return_scores=True)
    torch.cuda.synchronize()
    status = net.check(dev)
    fb += status['sinkhorn_fallback']
    for i in range(ns):
        assert torch.isfinite(outs[i][4]).all()
        if status['sinkhorn_fallback']:
            assert (outs[i][4] - serial[i][4]).abs().max() < 1e-4
            assert int((outs[i][0] != serial[i][0]).sum()) <= 4
        else:
            assert all(torch.equal(a, b) for a, b in zip(outs[i], serial[i]))
    print(f'round {r}: {ns} streams x 3 forwards (B={B}, N={N}) in {time.time() - t0:.2f} s, fallback taken: {status["sinkhorn_fallback"]}', flush=True)
print('OK, rounds with a fallback:', fb)
