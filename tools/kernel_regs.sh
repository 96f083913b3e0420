#!/bin/bash
# usage: tools/kernel_regs.sh attention [extra hipcc flags] -> per-kernel VGPR / AGPR / spill / LDS / scratch of csrc/<name>.hip (device-only compile)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
FL=""
[ "$N" = attention ] && FL="-fno-slp-vectorize"
[ "$N" = f64 ] && FL="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $FL "$@" --cuda-device-only -c $R/mdgat_matcher_amd/csrc/$N.hip -o /tmp/kr_$N.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr_$N.co --output=/tmp/kr_$N.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kr_$N.elf | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    agpr=blk.split()[0]
    name=g('name')
    import subprocess
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    print(f\"vgpr {g('vgpr_count'):>4} agpr {agpr:>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>4} scratch {g('private_segment_fixed_size'):>6} lds {g('group_segment_fixed_size'):>6}  {dn[:150]}\")
"
