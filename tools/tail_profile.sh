# usage: tools/tail_profile.sh   (on the GPU box)  — how a launch of the closest-hit kernel ramps up and drains.
# Builds libmipt_tail.so from a PATCHED COPY of csrc/ (the sources in the tree are not touched: their hash names the profiled library): every wave of
# k_wf_traverse<0> at depth `b` records when it started and when it ended (s_memrealtime, 100 MHz); tools/tail_profile.py renders and prints the shape.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
T=/tmp/tailsrc; rm -rf $T; mkdir -p $T/pathtracer_amd; cp -r $R/pathtracer_amd/csrc $T/pathtracer_amd/csrc; cp -r $R/include $T/include
python3 - $T <<'PY'
import sys
T=sys.argv[1]
p=T+'/pathtracer_amd/csrc/mipt_persistent.h'
s=open(p).read()
a='template <int MODE>\n__global__ void __launch_bounds__(MIPT_TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? MIPT_EXTEND_WAVES : MIPT_TRAVERSE_WAVES))) k_wf_traverse('
assert s.count(a)==1
s=s.replace(a,'__device__ unsigned long long g_tail_prof[2 * 16384];\n__device__ int g_tail_b = 1;\n'+a)
a='	MIPT_DECLARE_LDS_STACK(stk, wf.spill, MIPT_TRAV_BLOCK);\n	unsigned char* leafmap = lds_leafmap_ + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256;\n	auto extend_q'
assert s.count(a)==1
s=s.replace(a,'	const unsigned long long t_begin_ = __builtin_amdgcn_s_memrealtime();\n'+a)
a='	if (MODE == 2) traverse_queue<false>(sc, nodes, tris, wf, extend_q(b + 1), refill_threshold, inner_min_flags, stk, leafmap);\n}'
assert s.count(a)==1
s=s.replace(a,a[:-1]+'	if (MODE == 0 && b == g_tail_b && (threadIdx.x & 63) == 0) { const unsigned w = blockIdx.x * (MIPT_TRAV_BLOCK / 64) + (threadIdx.x >> 6); if (w < 16384) { g_tail_prof[2 * w] = t_begin_; g_tail_prof[2 * w + 1] = __builtin_amdgcn_s_memrealtime(); } }\n}')
open(p,'w').write(s)
p=T+'/pathtracer_amd/csrc/mipt.hip'
s=open(p).read()
s+='''
extern "C" int mipt_debug_tail_profile(unsigned long long* out, int nwaves, int b) {
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_prof), (size_t)nwaves * 16) != hipSuccess) return MIPT_ERR_HIP;
	std::vector<unsigned long long> z((size_t)2 * 16384, 0ull);
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_prof), z.data(), z.size() * 8) != hipSuccess) return MIPT_ERR_HIP;
	if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_b), &b, 4) != hipSuccess) return MIPT_ERR_HIP;
	return MIPT_OK;
}
'''
open(p,'w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -pthread -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value -o $R/pathtracer_amd/libmipt_tail.so $T/pathtracer_amd/csrc/mipt.hip
echo built
