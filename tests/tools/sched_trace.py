"""TEST INFRASTRUCTURE (uses the oracle): writes the traversal event trace that tests/tools/sched_sim.c replays.

The oracle restates the reference's ordered traversal (TriangleMesh.cpp:1133-1319); with a trace buffer installed it logs, per
ray and mesh, the nodes it pops in order (inner node: stack entries left behind it; leaf: triangle count).  The persistent
traversal kernels of the library visit exactly these nodes in exactly this order per ray (that is what the parity suite
checks), so WHICH lane steps WHEN under a given scheduling policy can be replayed on the CPU from the trace alone.

usage: python tests/tools/sched_trace.py <c1|c2|c3> <out.bin> [block_stride=3] [spp=2]
Pixels: every block_stride-th 8x8 block of the 1080p frame in both directions (the device numbers its path slots 8x8 block
by block, sample index outermost: a wave's first 64 rays are one block at one sample index)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import Oracle          # noqa: E402
from pathtracer_amd import scenes           # noqa: E402


def main():
    wl, out = sys.argv[1], sys.argv[2]
    stride = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    mesh, cfg, mat, text = scenes.workload(wl, 1920, 1080, spp, None)
    O = Oracle()
    O.apply_config(cfg)
    scenes.install(O, mesh, mat)
    O.prepare()
    pix = []
    for bi in range(0, (cfg.H + 7) // 8, stride):
        for bj in range(0, (cfg.W + 7) // 8, stride):
            for a in range(64):
                i, j = bi * 8 + (a >> 3), bj * 8 + (a & 7)
                if i < cfg.H and j < cfg.W:
                    pix.append((i, j))
    ij = np.ascontiguousarray(np.array(pix, np.int32))
    cap = 1 << 30
    buf = np.zeros(cap, np.uint8)
    f = O.cdll.o_trace_samples
    f.restype = C.c_size_t
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    n = f(O.ctx, ij.shape[0], ij.ctypes.data, 0, spp, buf.ctypes.data, cap)
    assert n <= cap, "trace buffer too small"
    buf[:n].tofile(out)
    print(text, ": %d pixels x %d spp, %d trace bytes -> %s" % (ij.shape[0], spp, n, out))


if __name__ == "__main__":
    main()
