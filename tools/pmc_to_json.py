#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE summaries of tools/profile_round.sh (KB per launch, per kernel) into
profiles/pmc_traffic.json, which bench.py reads for `roofline.traffic`:
    python tools/pmc_to_json.py <config index> <fetch summary> <write summary> <name of the committed profile file>
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 under-reports wide streaming reads by 2x: MI355X_MICROARCH.md)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = (('layer_split_kernelILi0ELi1E', 'layer_first'), ('layer_split_kernelILi1ELi1E', 'layer'), ('layer_split_kernelILi1ELi2E', 'layer_last'),
           ('layer_kernelILi0ELi1E', 'layer_first'), ('layer_kernelILi1ELi1E', 'layer'), ('layer_kernelILi1ELi2E', 'layer_last'),
           ('attention_stream_kernel', 'attention_full'), ('attention_topk', 'attention_topk'), ('attention_kernelILb1E', 'attention_topk'),
           ('attention_kernelILb0E', 'attention_full'), ('sinkhorn_scaling_kernel', 'sinkhorn'), ('scores_kernel', 'scores'),
           ('encoder_kernel', 'encoder'), ('extract_kernel', 'extract'),
           ('layer_tail_f64_kernel', 'f64_gemm'), ('encoder_f64_kernel', 'f64_gemm'), ('gemm_f64_kernel', 'f64_gemm'), ('attention_f64_kernelILb0E', 'f64_attention_full'), ('attention_f64_kernelILb1E', 'f64_attention_topk'))


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r'(\S+)\s+' + counter + r'\s+n=\s*(\d+)\s+avg=\s*([\d.]+)', line)
        if not m:
            continue
        for pat, cls in CLASSES:
            if pat in m.group(1):
                n, avg = int(m.group(2)), float(m.group(3))
                tot = out.setdefault(cls, [0, 0.0])
                tot[0] += n
                tot[1] += n * avg
                break
    return {k: v[1] / v[0] for k, v in out.items()}


def main():
    config, fetch, write, source = sys.argv[1:5]
    f, w = read(fetch, 'FETCH_SIZE'), read(write, 'WRITE_SIZE')
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[str(config)] = {k: [round((2 * f[k] + w.get(k, 0.0)) * 1024), source] for k in sorted(f)}
    json.dump(data, open(path, 'w'), indent=1, sort_keys=True)
    print(json.dumps(data[str(config)]))


if __name__ == '__main__':
    main()
