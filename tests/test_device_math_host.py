"""CPU test of the PRODUCT's sinf/cosf: the header the HIP kernels compile (csrc/mipt_sincos.h) is
built with g++ and compared with the host libm — the libm the reference's direction sampling calls —
on every float in [0, 6.5] (arguments on the path are float(2*pi)*u with u in [0,1])."""
import os
import subprocess

from helpers import ROOT


def test_sincos_bit_exact_with_libm(tmp_path):
    exe = str(tmp_path / "sincos_check")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "sincos_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "6.5"], check=True, capture_output=True, text=True).stdout.split()
    n, bad_sin, bad_cos = int(out[0]), int(out[1]), int(out[2])
    assert n > 1_000_000_000
    assert bad_sin == 0 and bad_cos == 0, (bad_sin, bad_cos)


def test_powf_bit_exact_with_libm(tmp_path):
    """csrc/mipt_powf.h against the host libm's powf: every float x in (0, 2] for the constant exponents 5.f
    (Schlick) and 2.2f, plus 20 M random (x, y) pairs over a wide domain and over the Phong lobe's domain."""
    exe = str(tmp_path / "powf_check")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "powf_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "20000000", "1"], check=True, capture_output=True, text=True).stdout.split()
    n, bad, unhandled = int(out[0]), int(out[1]), int(out[2])
    assert n > 2_000_000_000
    assert bad == 0 and unhandled == 0, (bad, unhandled)


def test_acosf_atanf_atan2f_bit_exact_with_libm(tmp_path):
    """csrc/mipt_invtrig.h against the host libm: acosf on every float in [-1, 1], atanf over all magnitudes, atan2f on
    50 M random pairs (the env-map lookup's domain and a wide log-uniform one) and the special points."""
    exe = str(tmp_path / "invtrig_check")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "invtrig_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "50000000"], check=True, capture_output=True, text=True).stdout.split()
    n, bad = int(out[0]), [int(v) for v in out[1:4]]
    assert n > 4_000_000_000
    assert bad == [0, 0, 0], bad


def test_expf_logf_tanf_bit_exact_with_libm(tmp_path):
    """csrc/mipt_explog.h (fog branch: fogContribution / int_exponential) against the host libm: expf and logf on every
    second float of the whole 32-bit range, tanf on those with |x| < 120 (the full sweep, 10.8 G evaluations with 0
    mismatches, is `explog_check 1`)."""
    exe = str(tmp_path / "explog_check")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "explog_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "2"], check=True, capture_output=True, text=True).stdout.split()
    n, bad = int(out[0]), [int(v) for v in out[1:4]]
    assert n > 5_000_000_000
    assert bad == [0, 0, 0], bad
