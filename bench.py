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
  roofline     - the dominant kernel class of the step against the f16 MFMA roofline, its average launch
                 duration measured live with HIP events on the launch stream (mdgat_profile);
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

from mdgat_matcher_amd import MDGAT, shard, synth  # noqa: E402

N_KPTS = 512
L_LAYERS = 9
S_ITERS = 100
BATCH = 64
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
SPLIT_FACTOR = 3.0               # f16 MFMAs executed per fp32-equivalent product (hi.hi, hi.lo, lo.hi)
PEAK_HBM_GBS = 8000.0
# HBM traffic per launch from rocprofv3 PMC passes of this same command (profiles/r1j_pmc_fetch_write_kb.txt):
# 2 x FETCH_SIZE (gfx950 under-reports wide streaming reads by 2x, MI355X_MICROARCH.md) + WRITE_SIZE, bytes.
# Algorithmic bytes of a layer launch at B=64: read x + msg 67.1 MB, write x + q/k/v 134.2 MB.
PMC_TRAFFIC_BYTES = {'layer': (2 * 37847.0 + 131072.0) * 1024, 'attention_full': (2 * 49212.6 + 32768.0) * 1024,
                     'attention_topk': (2 * 49326.5 + 32768.0) * 1024, 'sinkhorn': (2 * 74001.8 + 108986.3) * 1024}


# Algorithmic work of one launch of each kernel class: MACs x 2, fp32-equivalent.
# Every product runs as three f16 MFMAs (split operands), so the matrix cores execute 3x these FLOPs.
def class_work(B, n, L, S, sched):
    R = B * 2 * n
    per = {
        'encoder': {'flops': 2.0 * R * (32 * 64 + 64 * 128 + 64 * 128 + 256 * 128 + 4 * 32 + 33 * 64)},
        'layer': {'flops': 2.0 * R * (256 * 256 + 256 * 128 + 128 * 384)},
        'attention_full': {'flops': B * 2 * 4 * (2 * 2.0 * n * n * 32)},
        'attention_topk': {'flops': B * 2 * 4 * (2 * 2.0 * n * n * 32)},
        'scores': {'flops': B * 2.0 * n * n * 128},
        # log-domain Sinkhorn: 2 x S element visits of the (n+1)^2 matrix, 4 bytes each if it were streamed
        'sinkhorn': {'bytes': B * 4.0 * (2.0 * S * (n + 1) * (n + 1))},
        'extract': {'bytes': B * 4.0 * (n + 1) * (n + 1)},
    }
    return per


def kernel_breakdown(net, dev, inputs, B, n, L, S, sched, steps=5):
    """Average launch duration of every kernel class, measured live with HIP events on the launch stream
    inside the library (mdgat_profile), over `steps` extra forwards after the timed region."""
    net.profile(dev, True)
    with torch.no_grad():
        for _ in range(steps):
            net._run(*inputs)
    prof = net.profile(dev, False)
    work = class_work(B, n, L, S, sched)
    rows = []
    for name, (ms, launches) in prof.items():
        if launches == 0:
            continue
        rows.append({'kernel': name, 'launches_per_step': launches // steps, 'ms': ms / launches,
                     'step_ms': ms / steps, **work[name]})
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
    ap.add_argument('--attention-dtype', default='fp32', choices=['fp32', 'f16'],
                    help="'f16': single-f16 attention products (BASELINE configs[2]; outside the parity bar, not the headline)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    args = ap.parse_args()

    rank, world, local = shard.init_distributed(args.gpus)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    cfg = synth.default_config(L=L_LAYERS, sinkhorn_iterations=S_ITERS)
    cfg['attention_dtype'] = args.attention_dtype
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
        # initialisation, not warm-up: the first forwards of a process load the code objects, size the workspace and
        # meet a one-off runtime stall (one early step of 25-80 ms, usually the 3rd or 4th: tools/step_times.py)
        for _ in range(8):
            step()
        torch.cuda.synchronize()
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
            'dtype': 'f32 (products as 3 split-f16 MFMAs, fp32 accumulate)' if args.attention_dtype == 'fp32' else
                     'f16 attention products (single f16 operands, fp32 accumulate), f32 elsewhere: NOT the parity path',
            'data': 'synthetic',
            'config': {'workload': f'batch={B} synthetic pairs per GPU, N=M={N_KPTS} keypoints, 33-D FPFH, L={L_LAYERS}, '
                                   f'{S_ITERS} Sinkhorn iterations, ' + ('fp32 (BASELINE.json configs[1])' if args.attention_dtype == 'fp32' else 'f16 attention / fp32 Sinkhorn (BASELINE.json configs[2] at this batch)'),
                       'pairs_per_gpu': B, 'keypoints': N_KPTS, 'L': L_LAYERS, 'sinkhorn_iterations': S_ITERS,
                       'parallelism': f'pairs sharded {world}-way, no data-path collective'},
        }
        if not args.no_breakdown:
            sched = net._topk_schedule()
            rows = kernel_breakdown(net, dev, inputs, B, N_KPTS, L_LAYERS, S_ITERS, sched)
            dom = max(rows, key=lambda r: r['step_ms'])
            if 'flops' in dom:
                alg = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
                ach = SPLIT_FACTOR * alg
                roof = {'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F16_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': ach / PEAK_F16_MFMA_TFLOPS,
                        'traffic': PMC_TRAFFIC_BYTES.get(dom['kernel']) if B == BATCH else None,
                        'avg_launch_ms': dom['ms'], 'algorithmic_flops_per_launch': dom['flops'],
                        'algorithmic_tflops': alg,
                        'note': 'achieved = f16 MFMA FLOP/s executed = 3 x the fp32-equivalent algorithmic rate '
                                '(every product is hi.hi + hi.lo + lo.hi on the f16 matrix cores)'}
            else:
                ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
                roof = {'kernel': dom['kernel'], 'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': ach / PEAK_HBM_GBS, 'traffic': None, 'avg_launch_ms': dom['ms'],
                        'bytes_per_launch': dom['bytes']}
            out['roofline'] = roof
            out['kernels'] = [{'kernel': r['kernel'], 'launches_per_step': r['launches_per_step'], 'avg_ms': round(r['ms'], 4),
                               'step_ms': round(r['step_ms'], 3),
                               'algorithmic_tflops': round(r['flops'] / (r['ms'] * 1e-3) / 1e12, 1) if 'flops' in r else None}
                              for r in rows]
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(N_KPTS, L_LAYERS, S_ITERS)
        print(json.dumps(out), flush=True)
    shard.finalize(world)


if __name__ == '__main__':
    main()
