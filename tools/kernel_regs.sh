#!/bin/bash
# usage: tools/kernel_regs.sh [lib.so] [kernel-name regex]  -> VGPRs / SGPRs / scratch / LDS of the kernels of a library (from the code object's metadata)
lib=${1:-pathtracer_amd/libmipt.so}; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fatbin "$lib"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/code.o || { echo "unbundle failed"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/code.o | PAT="$pat" python3 -c "
import sys, re, os, subprocess
t = sys.stdin.read()
pat = os.environ['PAT']
for b in t.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', b) or [None, '?'])[1]
    name = g('name')
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if re.search(pat, dem):
        print('%-44s vgpr %3s sgpr %3s scratch %5s lds %6s vgpr_spills %s' % (dem[:44], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count')))
"
rm -rf $tmp
