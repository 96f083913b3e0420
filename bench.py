#!/usr/bin/env python3
"""Throughput bench of the MDGAT inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one forward of the hot path (encoders -> 2L attention layers -> Sinkhorn -> match extraction)
over one batch of synthetic frame pairs already resident in HBM.  Workload at every N: BASELINE.json
configs[1] per GPU (batch=64 pairs, 512 keypoints per frame, L=9, 100 Sinkhorn iterations, fp32); with
--gpus N every rank owns its own 64 pairs (weak scaling; pairs are independent, no data-path collective -
the only collective is the one-time RCCL broadcast of the packed weights from rank 0).
`--config {0,2,3,4}` runs another BASELINE.json configuration instead (never the headline; configs[3] = the
8-way sharded 4096 pairs is `--gpus 8 --config 3`, 512 pairs per GPU).

The timed window (--steps steps between barrier + synchronize, max over ranks) is repeated --windows times (default 5) and
the MEDIAN window is the reported value (`timing` lists them all).  Rank 0 prints ONE JSON line: metric keypoint-pairs/sec
(whole job), plus
  roofline     - the dominant kernel class of the step against the f16 MFMA roofline (or HBM for Sinkhorn): algorithmic
                 FLOPs per launch over its average launch duration, measured live with HIP events on the launch stream
                 (mdgat_profile); `frac_executed` = the 3x split-f16 MFMA work the matrix cores actually run.  The library runs a
                 batch of more than 32 768 keypoints as two half-batches in flight on two streams (csrc/api.hip: forward_batched;
                 that is what the timed windows measure); a kernel's own worth is measured with the launches NOT overlapping -
                 `roofline` / `kernels` come from extra forwards with mdgat_set_lanes(1) (one launch per class and layer over the
                 whole batch; agrees with a kernel trace of `MDGAT_FORWARD_LANES=1 python bench.py`), `roofline_two_lanes` /
                 `kernels_two_lanes` from the timed configuration (half the work per launch, intervals include waiting for CUs);
  roofline_qk  - the Q K^T contraction of full attention (the north star's "QK^T roofline"): the phase in isolation
                 (same kernel, softmax and P.V knocked out; mdgat_attention_qk_probe) timed with HIP events on the
                 launch stream, as algorithmic (`frac`) and executed (`frac_executed`) fraction of the dense f16 MFMA peak;
  parity       - measured in this run, outside the timed windows: the reference-held pairs of the workload's shape
                 (tests/golden/cfg_*.npz: inputs by seed, the imported reference's own fp64 outputs) through both arithmetic
                 modes - pairs within the literal 1e-4 on Z, max|dZ|, matches identical;
  roofline_sinkhorn - the Sinkhorn class against the fp32 vector peak (2 S (n+1)^2 FMA visits per pair);
  cpu_baseline - the CPU oracle (fp64 PyTorch restatement of the reference, "port") timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# RCCL on this pool's hosts shares device memory between the ranks of a node through dmabuf only: without this variable
# communicator set-up fails with `hipIpcGetMemHandle: invalid argument`.  It has to be in the environment before the HIP
# runtime is loaded, and a driver that launches this file with torch.distributed.run may not have exported it.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mdgat_matcher_amd import MDGAT, ops, shard, synth  # noqa: E402

# BASELINE.json configs: pairs per GPU, keypoints per frame, L, Sinkhorn iterations, attention dtype
CONFIGS = {
    0: dict(B=1, n=256, L=4, S=20, att='fp32', name='configs[0]: one frame pair, 256 keypoints, L=4, 20 Sinkhorn iterations'),
    1: dict(B=64, n=512, L=9, S=100, att='fp32', name='configs[1]'),
    2: dict(B=512, n=512, L=9, S=100, att='f16', name='configs[2]: f16 attention / fp32 Sinkhorn'),
    3: dict(B=512, n=512, L=9, S=100, att='fp32', name='configs[3]: 4096 pairs sharded 8-way = 512 per GPU'),
    4: dict(B=8, n=2048, L=9, S=200, att='fp32', name='configs[4]: 2048 dense keypoints, 200 Sinkhorn iterations'),
}
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
SPLIT_FACTOR = 3.0               # f16 MFMAs executed per fp32-equivalent product (hi.hi, hi.lo, lo.hi)
PEAK_HBM_GBS = 8000.0
PEAK_VECTOR_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector (= fp32 matrix) peak
PEAK_F64_MFMA_TFLOPS = 78.6      # v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD x 4 x 256 CUs x 2.4 GHz (= the fp64 vector rate)
# HBM traffic per launch is not measurable from inside the process: it comes from the rocprofv3 PMC passes of this same
# command committed under profiles/ (tools/profile_round.sh; 2 x FETCH_SIZE - gfx950 under-reports wide streaming reads
# by 2x, MI355X_MICROARCH.md - + WRITE_SIZE).  profiles/pmc_traffic.json: {config: {kernel class: [bytes, source file]}}
PMC_TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


# Algorithmic work of one launch of each kernel class: MACs x 2, fp32-equivalent.
# Every product runs as three f16 MFMAs (split operands), so the matrix cores execute 3x these FLOPs.
def class_work(B, n, L, S):
    R = B * 2 * n
    att = B * 2 * 4 * (2 * 2.0 * n * n * 32)
    return {
        'encoder': {'flops': 2.0 * R * (32 * 64 + 64 * 128 + 64 * 128 + 256 * 128 + 4 * 32 + 33 * 64)},
        'layer_first': {'flops': 2.0 * R * (128 * 384)},                                # q|k|v projection of layer 0 only
        'layer': {'flops': 2.0 * R * (256 * 256 + 256 * 128 + 128 * 384)},              # mlp + residual + next q|k|v
        'layer_last': {'flops': 2.0 * R * (256 * 256 + 256 * 128 + 128 * 128)},         # mlp + residual + final_proj
        'attention_full': {'flops': att},
        'attention_topk': {'flops': att},
        'scores': {'flops': B * 2.0 * n * n * 128},
        # Sinkhorn (scaling form, the coupling block register-resident for all S iterations): HBM traffic = the scores read once
        # + what the extraction needs = the on-chip-resident form of SURVEY 8d, 2 (n+1)^2 4 B per pair; the arithmetic is
        # 2 x S visits of the (n+1)^2 block, one FMA (2 FLOP) each, on the VECTOR pipe (no matrix-core work by nature)
        'sinkhorn': {'bytes': B * 4.0 * 2.0 * (n + 1) * (n + 1), 'vector_flops': B * 2.0 * (2.0 * S * (n + 1) * (n + 1)),
                     'streamed_bytes': B * 4.0 * (2.0 * S * (n + 1) * (n + 1))},
        'extract': {'bytes': B * 4.0 * (n + 1) * (n + 1)},
    }


def kernel_breakdown(net, dev, inputs, B, n, L, S, steps=5):
    """Average launch duration of every kernel class, measured live with HIP events on the launch stream
    inside the library (mdgat_profile), over `steps` extra forwards after the timed region."""
    net.profile(dev, True)
    with torch.no_grad():
        for _ in range(steps):
            net._run(*inputs)
    prof = net.profile(dev, False)
    # a large batch runs as balanced slices inside the library (api.hip: forward_sliced; one Sinkhorn launch per slice):
    # the work of ONE launch is that of a slice
    slices = max(1, prof['sinkhorn'][1] // steps) if 'sinkhorn' in prof else 1
    work = class_work(B / slices, n, L, S)
    rows = []
    for name, (ms, launches) in prof.items():
        if launches == 0:
            continue
        rows.append({'kernel': name, 'launches_per_step': launches // steps, 'ms': ms / launches,
                     'step_ms': ms / steps, **work[name]})
    return rows


def qk_roofline(dev, B, n, reps=20):
    """The Q K^T phase of the streamed full-attention kernel in isolation, HIP events on the launch stream."""
    g = torch.Generator(dev).manual_seed(0)
    qkv = torch.randn(B, 2 * n, 3, 4, 32, device=dev, generator=g) * 1.3
    probe = ops.QkProbe(qkv, n, n)
    for _ in range(3):
        probe.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        probe.run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    flops = B * 2 * 4 * (2.0 * n * n * 32)            # Q K^T only: half of an attention launch
    useful = flops / (ms * 1e-3) / 1e12
    sus = ops.mfma_sustained(dev)
    # the same phase as a standalone kernel with 32 (as shipped) and 64 queries per wave (every K fragment
    # feeding two sets of products); the 64-query form exists only in isolation - the full kernel cannot hold its registers
    standalone = {}
    for sets in (1, 2):
        for _ in range(3):
            probe.run(nq_sets=sets)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            probe.run(nq_sets=sets)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps
        standalone[f'standalone_{32 * sets}q_per_wave'] = {
            'avg_launch_ms': t, 'frac_executed': SPLIT_FACTOR * flops / (t * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
            'frac_executed_of_sustained': SPLIT_FACTOR * flops / (t * 1e-3) / 1e12 / sus['tflops']}
    return {'kernel': 'attention_stream_kernel, Q K^T phase in isolation (softmax and P.V knocked out)', 'bound': 'mfma',
            'avg_launch_ms': ms, 'algorithmic_flops_per_launch': flops, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'achieved': useful, 'frac': useful / PEAK_F16_MFMA_TFLOPS,
            'achieved_executed': SPLIT_FACTOR * useful, 'frac_executed': SPLIT_FACTOR * useful / PEAK_F16_MFMA_TFLOPS,
            'sustained_peak': sus['tflops'], 'sustained_clock_ghz': sus['clock_ghz'],
            'frac_executed_of_sustained': SPLIT_FACTOR * useful / sus['tflops'],
            'variants': standalone,
            'note': 'achieved / frac = algorithmic (fp32-equivalent) Q K^T FLOP/s; *_executed = f16 MFMA FLOP/s executed (3 MFMAs '
                    'per fp32-class product: no term can be dropped at the 1e-4 bar, profiles/precision_ablation_r2.txt)'}


def exact_mode_block(dev, B, n, L, S, steps=6, windows=3):
    """The reference-exact mode (MDGAT(arithmetic='fp64'): fp64 inputs, weights and matrix-core arithmetic through the last
    dynamic layer, csrc/f64.hip) on a bounded batch of the same workload: pairs/s, ms/pair, its kernel classes, and the fp64 GEMM
    class against the v_mfma_f64 roofline.  Never `value`."""
    import gc
    gc.collect()            # (modules of earlier legs release their handles now, not inside a timed loop)
    cfg = synth.default_config(L=L, sinkhorn_iterations=S, arithmetic='fp64')
    net = MDGAT(cfg).double()
    net.load_state_dict(synth.make_state_dict(L=L, seed=0))
    net = net.eval().to(dev)
    data = synth.make_batch(B, n, n, first_pair=0, dtype=torch.float64, device=dev)
    inputs = (data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'], data['descriptors1'])
    sched = net._topk_schedule()
    # the tail (every layer, final_proj, scores, Sinkhorn, the extraction's arg-maxes) is fp64 as well (config['sinkhorn_arithmetic'] =
    # 'auto'; frames beyond 575 keypoints: the streaming form of the fp64 Sinkhorn, beyond 2175 the fp32-class tail)
    tail64 = n <= 2175 and getattr(net, 'sinkhorn_arithmetic', 'fp32') != 'fp32'
    n64 = 2 * L if tail64 else max([i + 1 for i, k in enumerate(sched) if k > 0], default=0)
    one = tuple(t[:1].contiguous() for t in inputs)

    def timed(m):
        for _ in range(3):
            m._run(*inputs)
        torch.cuda.synchronize()
        ws = []
        for _ in range(windows):
            t0 = time.perf_counter()
            for _ in range(steps):
                m._run(*inputs)
            torch.cuda.synchronize()
            ws.append((time.perf_counter() - t0) / steps)
        # one pair per call, as test.py:132 runs the matcher (batch_size = 1)
        for _ in range(3):
            m._run(*one)
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):     # (median of five loops: a collection of an earlier module - its handle's hipFree - may land in one)
            t0 = time.perf_counter()
            for _ in range(20):
                m._run(*one)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 20 * 1e3)
        return sorted(ws)[len(ws) // 2], sorted(reps)[len(reps) // 2]

    with torch.no_grad():
        dt, one_ms = timed(net)
        fp32_tail = None
        if tail64:       # the same with the fp32-class tail behind the last dynamic layer (rounds 5 / 6; sinkhorn_arithmetic = 'fp32')
            net32 = MDGAT({**cfg, 'sinkhorn_arithmetic': 'fp32'}).double()
            net32.load_state_dict(synth.make_state_dict(L=L, seed=0))
            net32 = net32.eval().to(dev)
            dt32, one32 = timed(net32)
            net32.check(dev)
            net32._invalidate()
            fp32_tail = {'pairs_per_s': B / dt32, 'ms_per_step': 1e3 * dt32, 'one_pair_per_call_ms': one32,
                         'note': "config['sinkhorn_arithmetic'] = 'fp32': fp64 through the last dynamic layer only, the fp32-class kernels behind it - Z "
                                 'good to 7e-6 (inside the bar), an arg-max whose two candidates lie closer than that may fall the other way'}
        # the kernel classes with the launches NOT overlapping (one lane), as `roofline` / `kernels` of the headline: under two lanes
        # an interval between events also holds the other lane's launches
        net.set_lanes(1)
        net.profile(dev, True)
        for _ in range(3):
            net._run(*inputs)
        prof = net.profile(dev, False)
        net.set_lanes(2)
    net.check(dev)
    R = B * 2 * n
    att = B * 2 * 4 * (2 * 2.0 * n * n * 32)
    n_topk = sum(1 for k in sched[:n64] if k > 0)
    # algorithmic FLOPs of the fp64 classes per step
    flops = {'f64_gemm': 2.0 * R * (4 * 32 + 32 * 64 + 64 * 128 + 33 * 64 + 64 * 128 + 256 * 128) + n64 * 2.0 * R * (128 * 384 + 256 * 256 + 256 * 128) +
                         (2.0 * R * 128 * 128 + B * 2.0 * n * n * 128 - 2.0 * R * 128 * 384 if tail64 else 0.0),     # final_proj + scores instead of a next q|k|v
             'f64_attention_full': (n64 - n_topk) * att, 'f64_attention_topk': n_topk * att}
    ms, fl, _ = ops.mfma_f64_probe(dev, 3000)
    sustained = fl / ms / 1e9
    kernels = []
    for name, (tms, launches) in prof.items():
        if launches == 0:
            continue
        row = {'kernel': name, 'launches_per_step': launches // 3, 'step_ms': round(tms / 3, 3)}
        if name in flops and tms > 0:
            row['algorithmic_tflops'] = round(flops[name] / (tms / 3 * 1e-3) / 1e12, 1)
        if name == 'sinkhorn' and tail64 and n > 575 and tms > 0:
            # the streaming fp64 Sinkhorn reads its couplings once per iteration: 8 N (M + 1) bytes per pair; the class's
            # time also holds its init / last launch, the per-iteration b launches, the arg-max merge and the extraction
            gbs = B * 8.0 * n * (n + 1) * S / (tms / 3 * 1e-3) / 1e9
            row['hbm'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 3),
                          'note': 'algorithmic bytes 8 N (M + 1) S per pair over the whole Sinkhorn class of the step'}
        kernels.append(row)
    f64_ms = sum(prof[k][0] for k in flops) / 3
    dom = max(flops, key=lambda k: prof[k][0])
    # HBM bytes per launch of that class from the PMC passes of `MDGAT_FORWARD_LANES=1 tools/profile_f64.sh <tag> 64` (configs[1] only)
    traffic, traffic_source = pmc_traffic('1_f64' if (B, n, L, S) == (64, 512, 9, 100) else -1, dom)
    ach = flops[dom] / (prof[dom][0] / 3 * 1e-3) / 1e12
    total_f64 = sum(flops.values())
    return {'arithmetic': ("fp64 (v_mfma_f64_16x16x4_f64) for the encoders, all " + str(2 * L) + ' layers, final_proj and the score matrix; fp64 Sinkhorn, the '
                           "extraction's arg-maxes decided on the fp64 Z (csrc/sinkhorn_f64.hip" + ('' if n <= 575 else ': the streaming form, K in memory, one launch per iteration') + ')') if tail64 else
                          ("fp64 (v_mfma_f64_16x16x4_f64) for the encoders and layers 0.." + str(n64 - 1) + ' of ' + str(2 * L) +
                           ' (through the last dynamic layer); split-f16 kernels behind it'),
            'pairs_per_s': B / dt, 'ms_per_pair': 1e3 * dt / B, 'ms_per_step': 1e3 * dt, 'batch': B, 'steps': steps, 'windows': windows,
            'one_pair_per_call_ms': one_ms, 'fp32_tail': fp32_tail,
            'roofline': {'kernel': dom, 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / PEAK_F64_MFMA_TFLOPS, 'sustained_peak': sustained, 'frac_of_sustained': ach / sustained, 'traffic': traffic, 'traffic_source': traffic_source,
                         'all_f64_classes': {'achieved': total_f64 / (f64_ms * 1e-3) / 1e12, 'frac': total_f64 / (f64_ms * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS,
                                             'algorithmic_gflop_per_pair': total_f64 / B / 1e9},
                         'lanes': 1,
                         'note': 'algorithmic fp64 FLOPs of the class per step over its time per step (HIP events on the launch stream, '
                                 'mdgat_profile; launches not overlapping: mdgat_set_lanes(1) - pairs_per_s above is the two-lane '
                                 'configuration) against the fp64 matrix peak; sustained_peak = an MFMA-only loop on this device in this run '
                                 '(mdgat_mfma_f64_probe); the dynamic attention executes 1.5x its algorithmic FLOPs beyond 512 keys (Q K^T '
                                 'twice), 1x up to 512 (logits kept in registers)'},
            'kernels': kernels}


PARITY_FIXTURES = {(256, 4, 20): 'cfg_n256_L4_S20', (512, 9, 100): 'cfg_n512_L9_S100_b40', (2048, 9, 200): 'cfg_n2048_L9_S200_b'}


def parity_block(dev, n, L, S, stub=False):
    """Where both arithmetic modes stand against the north star's bar ON THIS BOX, IN THIS RUN (not a quoted string): the pairs the
    repository holds the REFERENCE's own outputs for at this workload's shape (tests/golden/cfg_*.npz, written by
    tools/make_goldens.py from the imported /root/reference; inputs and weights are regenerated from the seeds in the fixture)
    are matched in the fp32-class throughput mode and in the reference-exact mode; per mode: pairs whose Z lies within the
    literal 1e-4 of the reference on every held entry (every 8th row / column, the whole dustbin row and column), the largest
    |dZ|, whether matches0 / matches1 are bit-identical, the largest matching-score difference.  A few ms of GPU time, outside
    the timed windows."""
    import numpy as np
    name = PARITY_FIXTURES.get((n, L, S))
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}.npz') if name else None
    if not path or not os.path.exists(path):
        return {'error': f'no reference-held fixture for N={n} L={L} S={S}'}
    g = np.load(path)
    B, gn, gm, gL, gS, seed, first_pair = [int(x) for x in g['meta']]
    k = [None if x < 0 else int(x) for x in g['k']]
    sub = int(g['sub']) if 'sub' in g else 8
    ref_Z = np.concatenate([g['Z_sub'].reshape(B, -1), g['Z_lastrow'], g['Z_lastcol']], axis=1)
    sd = synth.make_state_dict(L=gL, seed=seed, bin_score=float(g['bin_score']) if 'bin_score' in g else 1.0)
    data = synth.make_batch(B, gn, gm, first_pair=first_pair, device=dev)
    modes = {}
    for mode in ('fp32', 'fp64'):
        net = MDGAT(synth.default_config(L=gL, k=k, sinkhorn_iterations=gS, arithmetic=mode)).double()
        net.load_state_dict(sd)
        net = net.eval().to(dev)
        with torch.no_grad():
            m0, m1, s0, s1, Z = net._run(data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'],
                                         data['descriptors1'], want_Z=True)
        if not stub:
            torch.cuda.synchronize()
            net.check(dev)
        Zc = Z.cpu().double().numpy()
        mine = np.concatenate([Zc[:, ::sub, ::sub].reshape(B, -1), Zc[:, -1, :], Zc[:, :, -1]], axis=1)
        err = np.abs(mine - ref_Z).max(1)
        es = max(np.abs(s0.cpu().double().numpy() - g['default_mscores0']).max(), np.abs(s1.cpu().double().numpy() - g['default_mscores1']).max())
        mm = int((m0.cpu().numpy() != g['default_matches0']).sum() + (m1.cpu().numpy() != g['default_matches1']).sum())
        modes[mode] = {'pairs': B, 'pairs_within_1e-4': int((err < 1e-4).sum()), 'max_abs_dZ': float(err.max()),
                       'matches_identical': mm == 0, 'matches_differing': mm, 'arg_maxes': int(m0.numel() + m1.numel()),
                       'max_abs_d_mscores': float(es)}
        net._invalidate()
    return {'fixture': f'tests/golden/{name}.npz: {B} pairs, N={gn} M={gm} L={gL} S={gS}, weights seed {seed} - outputs of the imported '
                       'reference (tools/make_goldens.py)',
            'bar': 'Z within 1e-4 of the reference on every held entry of a pair; matches bit-identical',
            'modes': modes,
            'note': 'fp32 = the throughput path (value / roofline above): a flipped top-k near-tie moves Z by a few 1e-4 around one '
                    'keypoint; fp64 = the reference-exact mode (exact_mode), what a float64 module - net.double(), test.py:193 - runs'}


def cpu_baseline(n, L, S, budget_s=15.0, max_pairs=64):
    """The CPU oracle (oracle/: fp64 PyTorch restatement pinned to the reference) on this box's cores."""
    from oracle import mdgat_oracle as O
    cores = synth.effective_cpu_count()
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(L=L, seed=0)
    cfg = synth.default_config(L=L, sinkhorn_iterations=S)
    with torch.no_grad():
        O.mdgat_forward(sd, cfg, synth.make_batch(1, n, n, first_pair=0))       # warm-up
        done, t0 = 0, time.perf_counter()
        while done < max_pairs and time.perf_counter() - t0 < budget_s:
            O.mdgat_forward(sd, cfg, synth.make_batch(1, n, n, first_pair=done))
            done += 1
        dt = time.perf_counter() - t0
    return {'value': done / dt, 'unit': 'keypoint-pairs/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{done} pairs, one at a time, N=M={n} L={L} S={S}, fp64 PyTorch CPU oracle, '
                      f'{torch.get_num_threads()} threads, {dt:.1f} s'}


def pmc_traffic(config, kernel):
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            entry = json.load(f).get(str(config), {}).get(kernel)
        return (float(entry[0]), entry[1]) if entry else (None, None)
    except (OSError, ValueError):
        return None, None


def self_launch(args):
    """`python bench.py --gpus N` must produce N ranks or fail (the reference's multi-GPU is one line, test.py:158).  Started
    without a launcher and asked for more than one GPU, this process becomes the launcher: it re-executes the same command
    line under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1 at a free port) and exits with its status.
    Started BY a launcher, the world it was given must be the one asked for."""
    launched = 'RANK' in os.environ or 'WORLD_SIZE' in os.environ
    if launched:
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world != args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to print a '
                             f'{world}-rank number under an n_gpus={args.gpus} request')
        return
    if args.gpus <= 1:
        return
    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and not os.environ.get('MDGAT_SHARE_DEVICE'):
        raise SystemExit(f'bench.py: --gpus {args.gpus} but this box has {torch.cuda.device_count()} GPU(s)')
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(shard._free_port()), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    print(f'[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '1'))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS),
                    help='BASELINE.json configs index (default 1 = the headline workload)')
    ap.add_argument('--batch', type=int, default=None, help='pairs per GPU per step (default: the configuration\'s)')
    ap.add_argument('--attention-dtype', default=None, choices=['fp32', 'f16'],
                    help="'f16': single-f16 attention products (BASELINE configs[2]; outside the parity bar, not the headline)")
    ap.add_argument('--windows', type=int, default=5,
                    help='the timed window of --steps steps is repeated this many times; value = the median window')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-dict-api', action='store_true', help='skip the forward(dict) throughput leg')
    ap.add_argument('--no-latency', action='store_true', help='skip the one-pair-per-call latency block')
    ap.add_argument('--arithmetic', default='fp32', choices=['fp32', 'fp64'],
                    help="'fp64': time the reference-exact mode (MDGAT(arithmetic='fp64')) as the step; never the headline")
    ap.add_argument('--no-exact-mode', action='store_true', help='skip the exact_mode block (reference-exact fp64 mode on a bounded batch)')
    ap.add_argument('--no-parity', action='store_true', help='skip the parity block (profiling runs: its 8-pair forwards would enter the per-kernel averages)')
    args = ap.parse_args()
    self_launch(args)

    c = CONFIGS[args.config]
    B = args.batch if args.batch is not None else c['B']
    n, L, S = c['n'], c['L'], c['S']
    att = args.attention_dtype or c['att']

    rank, world, local = shard.init_distributed(args.gpus)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the process group has {world} rank(s)')
    # tests/bench_stub_runner.py (CPU, no GPU in the build container) replaces MDGAT._run by a recorder to drive THIS file's
    # multi-rank control flow under gloo; it says so in the environment and the line it prints carries "stub": true.
    stub = os.environ.get('MDGAT_BENCH_STUB') == '1' and not torch.cuda.is_available()
    if not torch.cuda.is_available() and not stub:
        raise SystemExit('bench.py measures the HIP path on an MI355X: no GPU is visible (there is no CPU fallback)')
    dev = torch.device('cpu') if stub else torch.device('cuda', local)
    if not stub:
        torch.cuda.set_device(dev)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    cfg = synth.default_config(L=L, sinkhorn_iterations=S)
    cfg['attention_dtype'] = att
    cfg['arithmetic'] = args.arithmetic
    f64 = args.arithmetic == 'fp64'
    net = MDGAT(cfg).double().eval() if f64 else MDGAT(cfg).eval()
    if rank == 0:
        net.load_state_dict(synth.make_state_dict(L=L, seed=0, dtype=torch.float64 if f64 else torch.float32))
    shard.broadcast_weights(net, dev, rank, world)      # RCCL broadcast of the packed blob (no-op at world 1)

    first, count = shard.partition(B * world, rank, world)
    data = synth.make_batch(count, n, n, first_pair=first, dtype=torch.float64 if f64 else torch.float32, device=dev)
    inputs = (data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'], data['descriptors1'])

    def step():
        return net._run(*inputs)

    with torch.no_grad():
        # initialisation, not warm-up: the first forwards of a process load the code objects, size the workspace and
        # meet a one-off runtime stall (one early step of 25-80 ms, usually the 3rd or 4th: tools/step_times.py)
        for _ in range(8):
            step()
        sync()
        for _ in range(args.warmup):
            step()
        # EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks: one window.  A window at
        # B = 64 is 20 x 3.3 ms = 67 ms - a single sample on a box whose clocks move - so the window is repeated and the
        # MEDIAN window is reported (all windows are listed in the line).
        windows, own_windows = [], []
        for _ in range(max(1, args.windows)):
            shard.barrier(world)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            own = time.perf_counter() - t0          # this rank's own time for the window (before it waits for the others)
            shard.barrier(world)
            windows.append(shard.max_over_ranks(time.perf_counter() - t0, dev, world))
            own_windows.append(own)
    dt = sorted(windows)[len(windows) // 2]
    # every rank's pair count and own time per step (median window), gathered on all ranks: a straggler shows in the line,
    # and the partition must add up to the job
    mine = torch.tensor([float(count), 1e3 * sorted(own_windows)[len(own_windows) // 2] / args.steps], dtype=torch.float64)
    per_rank = shard.gather_matches(mine[None].to(dev if not stub else 'cpu'), world).cpu()
    assert int(per_rank[:, 0].sum()) == B * world, (per_rank[:, 0].tolist(), B, world)
    # asynchronous status of the forwards timed above: an f16 range violation raises here (the outputs would be invalid)
    range_violation = False
    try:
        status = {'sinkhorn_fallback': False} if stub else net.check(dev)
    except RuntimeError as e:          # (the f16 range guard: the outputs of the timed forwards are invalid - say so in the line)
        print(f'[bench] rank {rank}: {e}', file=sys.stderr, flush=True)
        status, range_violation = {'sinkhorn_fallback': False}, True
    range_violation = shard.max_over_ranks(float(range_violation), dev if not stub else 'cpu', world) > 0

    if rank == 0:
        pairs = B * world * args.steps
        parity = att == 'fp32'
        out = {
            'metric': 'keypoint-pairs/sec',
            'value': pairs / dt,
            'unit': 'keypoint-pairs/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'ms_per_pair': 1e3 * dt / args.steps / B,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64 (v_mfma_f64) through the last dynamic layer, f32 (3 split-f16 MFMAs) behind it: the reference-exact mode' if f64 else
                     'f32 (products as 3 split-f16 MFMAs, fp32 accumulate)' if parity else
                     'f16 attention products (single f16 operands, fp32 accumulate), f32 elsewhere: NOT the parity path',
            'data': 'synthetic',
            'config': {'workload': f'batch={B} synthetic pairs per GPU, N=M={n} keypoints, 33-D FPFH, L={L}, '
                                   f'{S} Sinkhorn iterations, ' + ('fp32' if parity else 'f16 attention / fp32 Sinkhorn') +
                                   f' (BASELINE.json {c["name"]}' + ('' if B == c['B'] else f' at batch {B}') + ')',
                       'baseline_config': args.config, 'pairs_per_gpu': B, 'keypoints': n, 'L': L, 'sinkhorn_iterations': S,
                       'parallelism': f'pairs sharded {world}-way, no data-path collective',
                       'collectives': torch.distributed.get_backend() if torch.distributed.is_initialized() else 'none',
                       # what the communicator saw (RCCL when collectives == 'nccl'): must equal n_gpus
                       'rccl_world_size': torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       'pairs_per_rank': [int(v) for v in per_rank[:, 0].tolist()]},
            'timing': {'windows': len(windows), 'value_is': 'median window', 'window_ms': [round(1e3 * w, 3) for w in windows],
                       'best_pairs_per_s': pairs / min(windows), 'worst_pairs_per_s': pairs / max(windows),
                       'per_rank_ms_per_step': [round(v, 4) for v in per_rank[:, 1].tolist()]},
            # mdgat_async_status after the timed windows (the timed step is the asynchronous MDGAT._run)
            'status': {'sinkhorn_fallback': bool(status['sinkhorn_fallback']), 'range_violation': bool(range_violation)},
        }
        # where both modes stand against the literal bar, MEASURED here on the reference-held pairs of this shape (parity_block)
        try:
            out['parity'] = {'skipped': '--no-parity'} if args.no_parity else parity_block(dev, n, L, S, stub=stub)
        except RuntimeError as e:
            print(f'[bench] parity: {e}', file=sys.stderr, flush=True)
            out['parity'] = {'error': str(e)}
        if 'modes' in out['parity']:
            pm = out['parity']['modes']
            out['status']['literal_1e-4_pairs'] = {m: f"{pm[m]['pairs_within_1e-4']}/{pm[m]['pairs']}" for m in pm}
            out['status']['matches_identical'] = {m: pm[m]['matches_identical'] for m in pm}
            out['status']['parity_source'] = 'measured in this run: `parity`'
        if f64:
            args.no_breakdown = True        # (the fp64 classes are broken down in exact_mode; `roofline` is then that block's)
        if stub:
            out['stub'] = True
            args.no_breakdown = args.no_cpu_baseline = True
        elif not args.no_dict_api:
            # the reference's dict API on the same batch: forward(dict) -> dict, which synchronises (the host-side test of
            # mdgat.py:465 and the status check) - what a caller of test.py:201 gets per call
            ddata = {k: v for k, v in data.items()}
            with torch.no_grad():
                for _ in range(3):
                    net(ddata)
                sync()
                k_steps = max(5, min(args.steps, 30))
                t0 = time.perf_counter()
                for _ in range(k_steps):
                    net(ddata)
                sync()
                out['dict_api'] = {'pairs_per_s': B * k_steps / (time.perf_counter() - t0), 'steps': k_steps, 'n_gpus': 1,
                                   'note': 'rank 0 only: net(dict) as test.py:201 calls it - one host synchronisation per call'}
        if not stub and not args.no_latency and B > 1:
            # ms per pair when the matcher is called the way test.py:132 calls it - ONE pair per call (batch_size = 1) - on the
            # same network: the asynchronous _run back to back, and the dict API with its host synchronisation.  The other half
            # of BASELINE.json's metric ("keypoint-pairs/sec + ms/pair"); never `value`.
            one = synth.make_batch(1, n, n, first_pair=first, dtype=torch.float32, device=dev)
            one_in = (one['keypoints0'], one['scores0'], one['descriptors0'], one['keypoints1'], one['scores1'], one['descriptors1'])
            with torch.no_grad():
                for _ in range(10):
                    net._run(*one_in)
                sync()
                reps = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(50):
                        net._run(*one_in)
                    sync()
                    reps.append((time.perf_counter() - t0) / 50)
                for _ in range(3):
                    net(one)
                t0 = time.perf_counter()
                for _ in range(30):
                    net(one)
                sync()
                t_dict = (time.perf_counter() - t0) / 30
            out['latency'] = {'pairs_per_call': 1, 'ms_per_pair': 1e3 * sorted(reps)[len(reps) // 2], 'ms_per_pair_dict_api': 1e3 * t_dict,
                              'note': f'one pair per call (test.py:132 runs batch_size = 1), N=M={n}, L={L}, {S} Sinkhorn iterations: '
                                      'median of 5 loops of 50 back-to-back asynchronous forwards; dict API = net(dict) with its host '
                                      'synchronisation per call'}
        if not args.no_breakdown:
            rows = kernel_breakdown(net, dev, inputs, B, n, L, S)
            nsl = next((r['launches_per_step'] for r in rows if r['kernel'] == 'sinkhorn'), 1)
            two_lanes = nsl > 1 and net.lanes != 1 and os.environ.get('MDGAT_FORWARD_LANES') != '1'
            if nsl > 1:
                out['config']['batch_slices'] = (f'{nsl} slices of {-(-B // nsl)} pairs per step inside the library' +
                                                 (', alternating between two streams (two slices in flight)' if two_lanes else ' (cache blocking)') +
                                                 '; per-launch work below is a slice\'s')
            dom = max(rows, key=lambda r: r['step_ms'])

            def roofline_block(dom, pmc_key):
                traffic, source = pmc_traffic(pmc_key if B == c['B'] and att == c['att'] else -1, dom['kernel'])
                if 'flops' in dom:
                    alg = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
                    ach = (SPLIT_FACTOR if parity or not dom['kernel'].startswith('attention') else 1.0) * alg
                    roof = {'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': alg, 'peak': PEAK_F16_MFMA_TFLOPS,
                            'unit': 'TFLOP/s', 'frac': alg / PEAK_F16_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': source,
                            'avg_launch_ms': dom['ms'], 'algorithmic_flops_per_launch': dom['flops'],
                            'achieved_executed': ach, 'frac_executed': ach / PEAK_F16_MFMA_TFLOPS,
                            'note': 'achieved / frac = ALGORITHMIC (fp32-equivalent) FLOPs of one launch (SURVEY 8d, DESIGN.md '
                                    'section 5) over its average duration, against the dense f16 MFMA peak; *_executed = the f16 '
                                    'MFMA FLOPs the matrix cores run for it = 3 x algorithmic (every product is hi.hi + hi.lo + '
                                    'lo.hi); traffic = HBM bytes per launch from the PMC passes of this command named in '
                                    'traffic_source (not measured in this run)'}
                else:
                    ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
                    roof = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                            'frac': ach / PEAK_HBM_GBS, 'traffic': traffic, 'traffic_source': source, 'avg_launch_ms': dom['ms'],
                            'bytes_per_launch': dom['bytes'],
                            'note': 'bytes = the on-chip-resident algorithmic traffic 2 (n+1)^2 4 B per pair (SURVEY 8d): the kernel keeps '
                                    'the coupling block in registers for all iterations, so it is bound by neither HBM nor the matrix '
                                    'cores but by vector issue and the partner hand-off latency - see `vector`'}
                    if 'vector_flops' in dom:
                        v = dom['vector_flops'] / (dom['ms'] * 1e-3) / 1e12
                        roof['vector'] = {'achieved': v, 'peak': PEAK_VECTOR_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': v / PEAK_VECTOR_F32_TFLOPS,
                                          'flops_per_launch': dom['vector_flops'],
                                          'note': '2 S (n+1)^2 FMAs per pair on the vector pipe against the fp32 vector peak'}
                if roof['bound'] == 'mfma':
                    # the ceiling that exists on this box: the matrix cores under nothing but MFMAs on random operands (the chip
                    # clocks to its power budget: ~1.6 instead of 2.4 GHz on this pool), measured live (mdgat_mfma_probe)
                    sus = ops.mfma_sustained(dev)
                    roof.update({'sustained_peak': sus['tflops'], 'sustained_clock_ghz': sus['clock_ghz'],
                                 'frac_executed_of_sustained': roof['achieved_executed'] / sus['tflops'],
                                 'sustained_note': 'sustained_peak = f16 MFMA TFLOP/s of an MFMA-only loop on random operands on '
                                                   'this device, measured in this run (mdgat_mfma_probe); frac stays against the '
                                                   'dense peak at 2.4 GHz; frac_executed_of_sustained = executed rate over it'})
                return roof

            roof = roofline_block(dom, str(args.config))
            if two_lanes:
                # The timed configuration keeps two half-batches in flight on two streams (api.hip: forward_batched): every launch
                # shares the device with a launch of the other lane.  What a kernel is worth is measured with the launches NOT
                # overlapping: the same kernel class launched alone over the whole batch (mdgat_set_lanes(1), the configuration
                # of rounds 1-3; a kernel trace of `MDGAT_FORWARD_LANES=1 python bench.py` shows the same durations) - that is
                # `roofline` / `kernels`.  `roofline_two_lanes` / `kernels_two_lanes` are the launches of the timed configuration:
                # half the work per launch, and the interval between consecutive events on a lane's stream, which includes the
                # time a launch waits for the other lane's workgroups to leave the CUs (a kernel trace, which clocks a kernel from
                # its first wave, shows shorter durations: profiles/).
                shared = roof
                shared['lanes'] = 2
                shared['note'] += ('; TWO lanes in flight: a launch covers half of the batch and shares the device with a launch of '
                                   'the other lane; avg_launch_ms is the interval between consecutive events on the lane\'s stream '
                                   '(includes waiting for CUs)')
                out['roofline_two_lanes'] = shared
                out['kernels_two_lanes'] = [{'kernel': r['kernel'], 'launches_per_step': r['launches_per_step'],
                                             'avg_ms': round(r['ms'], 4), 'step_ms': round(r['step_ms'], 3)} for r in rows]
                net.set_lanes(1)
                rows = kernel_breakdown(net, dev, inputs, B, n, L, S)
                net.set_lanes(2)
                roof = roofline_block(next(r for r in rows if r['kernel'] == dom['kernel']), f'{args.config}:single_lane')
                roof['lanes'] = 1
                roof['note'] += ('; measured with the launches of the batch NOT overlapping (mdgat_set_lanes(1): one launch per class '
                                 'and layer over the whole batch) - the timed windows above run two half-batch lanes concurrently, '
                                 'see roofline_two_lanes')
            out['roofline'] = roof
            sk = next((r for r in rows if r['kernel'] == 'sinkhorn'), None)
            if sk is not None:
                v = sk['vector_flops'] / (sk['ms'] * 1e-3) / 1e12
                out['roofline_sinkhorn'] = {
                    'kernel': 'sinkhorn', 'bound': 'vector', 'achieved': v, 'peak': PEAK_VECTOR_F32_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': v / PEAK_VECTOR_F32_TFLOPS, 'avg_launch_ms': sk['ms'], 'flops_per_launch': sk['vector_flops'],
                    'fma_visits_per_s': sk['vector_flops'] / 2.0 / (sk['ms'] * 1e-3), 'step_ms': sk['step_ms'],
                    'hbm': {'achieved': sk['bytes'] / (sk['ms'] * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                            'frac': sk['bytes'] / (sk['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 'bytes_per_launch': sk['bytes']},
                    'note': '2 S (n+1)^2 FMA visits per pair (one FMA = 2 FLOP) on the vector pipe, coupling block register-resident '
                            'for all S iterations, against the fp32 vector peak; hbm = the on-chip-resident algorithmic traffic '
                            '2 (n+1)^2 4 B per pair (SURVEY 8d).  Bound by neither: 100 dependent iterations of reduction -> LDS -> '
                            'barrier -> partner hand-off through L2 (profiles/NOTES_r5.md section 6)'}
            if n % 64 == 0 and att == 'fp32':
                out['roofline_qk'] = qk_roofline(dev, B, n)
            out['kernels'] = [{'kernel': r['kernel'], 'launches_per_step': r['launches_per_step'], 'avg_ms': round(r['ms'], 4),
                               'step_ms': round(r['step_ms'], 3),
                               'algorithmic_tflops': round(r['flops'] / (r['ms'] * 1e-3) / 1e12, 1) if 'flops' in r else None,
                               **({'vector_tflops': round(r['vector_flops'] / (r['ms'] * 1e-3) / 1e12, 2)} if 'vector_flops' in r else {})}
                              for r in rows]
        if not stub and not args.no_exact_mode and att == 'fp32':
            try:
                out['exact_mode'] = exact_mode_block(dev, min(B, 64 if n <= 512 else 4), n, L, S)
            except RuntimeError as e:      # (the headline above is measured and stands; the failure is reported, not hidden)
                print(f'[bench] exact_mode: {e}', file=sys.stderr, flush=True)
                out['exact_mode'] = {'error': str(e)}
            if f64 and 'roofline' in out['exact_mode']:
                out['roofline'] = out['exact_mode']['roofline']
            if 'modes' in out.get('parity', {}) and 'error' not in out['exact_mode']:
                out['exact_mode']['parity'] = {**out['parity']['modes']['fp64'], 'fixture': out['parity']['fixture'],
                                               'source': 'measured in this run (`parity`)'}
        if not stub:
            # the legs behind the timed windows (dict API, latency, breakdown) ran more forwards: their status as well
            try:
                after = net.check(dev)
                out['status']['sinkhorn_fallback'] = out['status']['sinkhorn_fallback'] or bool(after['sinkhorn_fallback'])
            except RuntimeError as e:
                print(f'[bench] {e}', file=sys.stderr, flush=True)
                out['status']['range_violation'] = True
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, L, S, max_pairs=128 if n <= 512 else 4)
        print(json.dumps(out), flush=True)
    shard.finalize(world)


if __name__ == '__main__':
    main()
