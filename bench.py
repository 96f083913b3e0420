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

Rank 0 prints ONE JSON line: metric keypoint-pairs/sec (whole job), plus
  roofline     - the dominant kernel of the step against the fp32-MFMA roofline, its average launch
                 duration measured live with HIP events on the launch stream;
  cpu_baseline - the CPU oracle (fp64 PyTorch restatement of the reference, "port") timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mdgat_matcher_amd import MDGAT, ops, shard, synth  # noqa: E402

N_KPTS = 512
L_LAYERS = 9
S_ITERS = 100
BATCH = 64
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def event_time_ms(fn, reps, warmup=2):
    """Average duration of fn() in ms, HIP events on torch's current stream (= the launch stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def kernel_breakdown(dev, B, n, L, S, topk_sched, reps=5):
    """Per-kernel average launch duration (HIP events) on scratch tensors of the bench shapes, with the
    algorithmic FLOPs / bytes of one launch."""
    P = 2 * n
    R = B * P
    g = torch.Generator(dev).manual_seed(0)
    x = torch.randn(R, 128, device=dev, generator=g)
    msgx = torch.randn(R, 256, device=dev, generator=g)
    hid = torch.randn(R, 256, device=dev, generator=g)
    wqkv = torch.randn(384, 128, device=dev, generator=g) * 0.09
    w1 = torch.randn(256, 256, device=dev, generator=g) * 0.06
    w2 = torch.randn(128, 256, device=dev, generator=g) * 0.06
    b384 = torch.randn(384, device=dev, generator=g)
    b256 = torch.randn(256, device=dev, generator=g)
    b128 = torch.randn(128, device=dev, generator=g)
    qkv = torch.randn(B, P, 3, 4, 32, device=dev, generator=g)
    scores = torch.randn(B, n, n, device=dev, generator=g) * 3
    n_full = sum(1 for k in topk_sched if k == 0)
    dyn_ks = sorted({k for k in topk_sched if k > 0})
    rows = []

    def add(name, launches, fn, flops=0.0, bytes_=0.0, bound='mfma'):
        ms = event_time_ms(fn, reps)
        rows.append({'kernel': name, 'launches_per_step': launches, 'ms': ms, 'flops': flops, 'bytes': bytes_, 'bound': bound})

    add('gemm_qkv_128x384', 2 * L, lambda: ops.pointwise(x, wqkv, b384), flops=2.0 * R * 128 * 384)
    add('attention_full', n_full, lambda: ops.attention(qkv, n, n, False, 0), flops=B * 1024.0 * n * n)
    for k in dyn_ks:
        cnt = sum(1 for kk in topk_sched if kk == k)
        add(f'attention_top{k}', cnt, lambda k=k: ops.attention(qkv, n, n, False, k), flops=B * 1024.0 * n * n)
    add('gemm_mlp1_256x256', 2 * L, lambda: ops.pointwise(msgx, w1, b256, relu=True), flops=2.0 * R * 256 * 256)
    add('gemm_mlp2_256x128', 2 * L, lambda: ops.pointwise(hid, w2, b128, residual=x), flops=2.0 * R * 256 * 128)
    add('sinkhorn', 1, lambda: ops.sinkhorn(scores, 1.0, S), bytes_=B * 4.0 * (S * n * n + (n + 1) * (n + 1)), bound='hbm')
    return rows


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH, help='pairs per GPU per step (BASELINE config: 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    args = ap.parse_args()

    rank, world, local = shard.init_distributed(args.gpus)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    cfg = synth.default_config(L=L_LAYERS, sinkhorn_iterations=S_ITERS)
    net = MDGAT(cfg).eval()
    if rank == 0:
        net.load_state_dict(synth.make_state_dict(L=L_LAYERS, seed=0, dtype=torch.float32))
    shard.broadcast_weights(net, dev, rank, world)      # RCCL broadcast of the packed blob (no-op at world 1)

    B = args.batch
    first, count = shard.partition(B * world, rank, world)
    data = synth.make_batch(count, N_KPTS, N_KPTS, first_pair=first, dtype=torch.float32, device=dev)
    inputs = (data['keypoints0'], data['scores0'], data['descriptors0'], data['keypoints1'], data['scores1'], data['descriptors1'])

    def step():
        return net._run(*inputs)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        shard.barrier(world)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        shard.barrier(world)
        dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dev, world)

    if rank == 0:
        pairs = B * world * args.steps
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
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'batch={B} synthetic pairs per GPU, N=M={N_KPTS} keypoints, 33-D FPFH, L={L_LAYERS}, '
                                   f'{S_ITERS} Sinkhorn iterations, fp32 (BASELINE.json configs[1])',
                       'pairs_per_gpu': B, 'keypoints': N_KPTS, 'L': L_LAYERS, 'sinkhorn_iterations': S_ITERS,
                       'parallelism': f'pairs sharded {world}-way, no data-path collective'},
        }
        if not args.no_breakdown:
            sched = net._topk_schedule()
            rows = kernel_breakdown(dev, B, N_KPTS, L_LAYERS, S_ITERS, sched)
            for r in rows:
                r['step_ms'] = r['ms'] * r['launches_per_step']
            dom = max(rows, key=lambda r: r['step_ms'])
            if dom['bound'] == 'mfma':
                ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
                roof = {'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
                        'avg_launch_ms': dom['ms'], 'flops_per_launch': dom['flops']}
            else:
                ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
                roof = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': ach / PEAK_HBM_GBS, 'traffic': None, 'avg_launch_ms': dom['ms'],
                        'bytes_per_launch': dom['bytes']}
            out['roofline'] = roof
            out['kernels'] = [{'kernel': r['kernel'], 'launches_per_step': r['launches_per_step'], 'avg_ms': round(r['ms'], 4),
                               'step_ms': round(r['step_ms'], 3),
                               'tflops': round(r['flops'] / (r['ms'] * 1e-3) / 1e12, 2) if r['flops'] else None}
                              for r in rows]
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N_KPTS, L_LAYERS, S_ITERS)
        print(json.dumps(out), flush=True)
    shard.finalize(world)


if __name__ == '__main__':
    main()
