"""Full attention alone at the sizes in ATTN_SIZES ("B:N,B:N,..."), for rocprofv3 (tools/kprof.sh with KPROF_SCRIPT)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdgat_matcher_amd import ops  # noqa: E402

for item in os.environ.get('ATTN_SIZES', '64:512').split(','):
    B, n = (int(v) for v in item.split(':'))
    qkv = torch.randn(B, 2 * n, 3, 4, 32, device='cuda:0') * 1.3
    for _ in range(12):
        ops.attention(qkv, n, n, False, 0)
    torch.cuda.synchronize()
