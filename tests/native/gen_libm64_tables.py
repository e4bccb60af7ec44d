"""Writes pathtracer_amd/csrc/mipt_libm64_tables.h: the lookup tables of the host libm's double-precision exp / pow / sin / cos / acos / atan2
(glibc 2.35, Ubuntu 2.35-0ubuntu3.x, x86-64), read from the .rodata of the installed libm.so.6 — the library the
reference links against.  The addresses below are those of that build; each table is checked against values that are
known independently (first entries, constants next to the table) before anything is written.

    python tests/native/gen_libm64_tables.py [/lib/x86_64-linux-gnu/libm.so.6]
"""
import math, os, struct, sys

LICENCE = """//
// PROVENANCE AND LICENCE.  These arrays are numeric constant tables of the GNU C Library (glibc 2.35, Ubuntu 22.04's
// libm.so.6), read out of the shipped binary's .rodata — the same values as sysdeps/ieee754/dbl-64/{e_exp_data.c,
// e_pow_log_data.c, sincostab.c, asincos.tbl, uatan.tbl, root.tbl} of the glibc source tree.  glibc is
//   Copyright (C) Free Software Foundation, Inc.,
// distributed under the GNU Lesser General Public License, version 2.1 or (at your option) any later version
// (https://www.gnu.org/licenses/old-licenses/lgpl-2.1.html); the exp / pow tables originate from ARM's optimized-routines
// (Copyright (C) Arm Limited, MIT licence) and the sin / cos / asin / atan tables from the IBM Accurate Mathematical Library
// (Copyright (C) IBM / Free Software Foundation, LGPL-2.1-or-later).  This file and csrc/mipt_libm64.h (the routines
// restated from the same library so that the device computes what the reference's libm calls return) are therefore
// LGPL-2.1-or-later material embedded in this library: redistribute the library under terms compatible with it (the
// relinking requirement of LGPL section 6 is met by shipping this source; the generator, tests/native/gen_libm64_tables.py,
// regenerates the file from any installed glibc 2.35 libm.so.6).  Nothing here comes from the reference repository.
"""


LIBM = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
EXP_DATA = 0xaf960       # struct exp_data: invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], tab[2*128]
POW_LOG_DATA = 0xb1b20   # struct pow_log_data: ln2hi, ln2lo, poly[7], tab[128] of {invc, pad, logc, logctail}
SINCOSTAB = 0xaeb80      # __sincostab: 440 doubles, {sin, sin tail, cos, cos tail} of k/128, k = 0..109
ASNCS = 0xb90a0          # asncs (asincos.tbl): 2568 doubles, per interval {x0, Taylor coefficients of asin at x0, asin(x0) in two parts}
POWTWO = 0xb8bc0         # powtwo[28] = 2^k, followed by inroot[128] (root.tbl: 1/sqrt of the interval midpoints), used by acos near 1
INROOT = 0xb8ca0
CIJ = 0xbe0e0            # cij (uatan.tbl): 241 rows {x_i, atan(x_i), 5 Taylor coefficients}, x_i ~ (i + 16.25) / 256

b = open(LIBM, "rb").read()
# .rodata of this build is mapped at its file offset (readelf -S: address == offset)
d = lambda a, n=1: struct.unpack_from("<%dd" % n, b, a)
q = lambda a, n=1: struct.unpack_from("<%dQ" % n, b, a)

assert d(EXP_DATA)[0] == 128 / math.log(2) or abs(d(EXP_DATA)[0] - 128 / math.log(2)) < 1e-13, "exp_data not where expected"
assert d(EXP_DATA + 8)[0] == 1.5 * 2.0 ** 52
exp_tab = q(EXP_DATA + 0x70, 256)
assert exp_tab[0] == 0 and exp_tab[1] == 0x3ff0000000000000, "exp table does not start with 2^0"
for k in (1, 37, 127):       # tab[2k+1] + (k << 45) are the bits of 2^(k/128) rounded to double
    assert abs(struct.unpack("<d", struct.pack("<Q", exp_tab[2 * k + 1] + (k << 45)))[0] - 2.0 ** (k / 128.0)) < 3e-16

assert abs(d(POW_LOG_DATA)[0] - math.log(2)) < 1e-10 and d(POW_LOG_DATA + 16)[0] == -0.5
pow_tab = d(POW_LOG_DATA + 72, 512)
for i in (0, 50, 127):       # logc + logctail = -log(invc)
    invc, pad, logc, tail = pow_tab[4 * i:4 * i + 4]
    assert pad == 0.0 and abs(logc + tail + math.log(invc)) < 1e-15, i

sc = d(SINCOSTAB, 440)
for k in (0, 1, 64, 109):
    assert abs(sc[4 * k] + sc[4 * k + 1] - math.sin(k / 128.0)) < 1e-16 and abs(sc[4 * k + 2] + sc[4 * k + 3] - math.cos(k / 128.0)) < 1e-16, k

asn = d(ASNCS, 2568)
assert asn[0] == 0.126953125 and abs(asn[8] - math.asin(asn[0])) < 1e-15 and abs(asn[1] - 1 / math.sqrt(1 - asn[0] ** 2)) < 1e-15, "asncs not where expected"
for n, stride in ((352, 11), (1056, 12), (992 + 13 * 64, 13), (884 + 14 * 0x6c, 14), (768 + 15 * 0x74, 15)):   # first interval of each range: x0, asin'(x0), asin(x0)
    assert abs(asn[n + 1] - 1 / math.sqrt(1 - asn[n] ** 2)) < 1e-13 and abs(asn[n + stride - 3] - math.asin(asn[n])) < 1e-7, n
powtwo = d(POWTWO, 28)
assert all(powtwo[k] == 2.0 ** k for k in range(28)) and POWTWO + 28 * 8 == INROOT
inroot = d(INROOT, 128)
assert abs(inroot[0] - 1 / math.sqrt(0.50390625)) < 1e-3 and all(inroot[k] > inroot[k + 1] for k in range(127)) and abs(inroot[127] - 1 / math.sqrt(2.0)) < 6e-3
cij = d(CIJ, 241 * 7)
for i in (0, 100, 240):
    assert abs(cij[7 * i + 1] - math.atan(cij[7 * i])) < 1e-16 and abs(cij[7 * i + 2] - 1 / (1 + cij[7 * i] ** 2)) < 1e-15 and abs(cij[7 * i] - (i + 16.25) / 256) < 2e-3, i

out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "pathtracer_amd", "csrc", "mipt_libm64_tables.h")
with open(out, "w") as f:
    f.write("// mipt_libm64_tables.h — GENERATED by tests/native/gen_libm64_tables.py from the installed libm.so.6 (glibc 2.35): the tables of\n"
            "// the double-precision exp / pow (exp_data.tab, pow_log_data.tab), sin / cos (__sincostab), acos (asncs, inroot) and atan2 (cij).  Bit patterns, not decimal.\n" + LICENCE + "#pragma once\n#include <stdint.h>\n")
    def arr(name, vals, per):
        f.write("MIPT_L64_TABLE uint64_t %s[%d] = {\n" % (name, len(vals)))
        for i in range(0, len(vals), per):
            f.write("\t" + ", ".join("0x%016xull" % v for v in vals[i:i + per]) + ",\n")
        f.write("};\n")
    arr("mipt_l64_exp_tab", list(exp_tab), 4)
    arr("mipt_l64_pow_log_tab", [struct.unpack("<Q", struct.pack("<d", v))[0] for v in pow_tab], 4)
    arr("mipt_l64_sincos_tab", [struct.unpack("<Q", struct.pack("<d", v))[0] for v in sc], 4)
    arr("mipt_l64_asncs_tab", [struct.unpack("<Q", struct.pack("<d", v))[0] for v in asn], 4)
    arr("mipt_l64_inroot_tab", [struct.unpack("<Q", struct.pack("<d", v))[0] for v in inroot], 4)
    arr("mipt_l64_cij_tab", [struct.unpack("<Q", struct.pack("<d", v))[0] for v in cij], 7)
print("wrote", out)
