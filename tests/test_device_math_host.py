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


def _cpu_has_fma_avx2():
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    return "fma" in flags and "avx2" in flags


def test_exp_pow_sincos_acos_atan2_double_bit_exact_with_libm(tmp_path):
    """csrc/mipt_libm64.h (subsurface weight, random_Phong's sampling frame, the MERL transform) against the host libm's
    double-precision exp / pow / sincos / acos / atan2: the whole finite range of exp incl. subnormal results, pow over the
    Phong lobe's domain and wide log-uniform ones, sincos over the path's domains (2 pi x float, float angles) and
    everything below 1.05e8, acos over [-1, 1] (uniform, towards 0, towards +-1, components of unit vectors) and outside,
    atan2 over all quadrants, ratios, magnitudes and every pair of special values — 580 M evaluations here (8.7 G with 0
    mismatches: `libm64_check 300000000`).  glibc runs its FMA builds of exp, pow, acos and atan2 on CPUs with FMA + AVX2,
    which is what the header restates: on other CPUs the comparison is void."""
    import pytest
    if not _cpu_has_fma_avx2():
        pytest.skip("host CPU without FMA + AVX2: glibc selects other variants of exp / pow than the ones restated")
    exe = str(tmp_path / "libm64_check")
    subprocess.run(["g++", "-O2", "-fopenmp", "-ffp-contract=off", "-mfma", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "libm64_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "20000000"], check=True, capture_output=True, text=True).stdout.split()
    n, bad_exp, bad_pow, bad_sincos, sincos_vs_sin_cos, bad_acos, bad_atan2 = (int(v) for v in out[:7])
    assert n > 500_000_000
    assert (bad_exp, bad_pow, bad_sincos, bad_acos, bad_atan2) == (0, 0, 0, 0, 0)
    assert sincos_vs_sin_cos > 0        # libm's sincos() is not its sin() next to its cos(): the reason sincos is what is restated


def test_libm64_tables_are_those_of_the_installed_libm(tmp_path):
    """csrc/mipt_libm64_tables.h is generated from the .rodata of libm.so.6 (tests/native/gen_libm64_tables.py): where the
    installed libm is the build the addresses were taken from, regenerating gives the committed file."""
    import pytest
    import shutil
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    gen = os.path.join(ROOT, "tests", "native", "gen_libm64_tables.py")
    committed = os.path.join(ROOT, "pathtracer_amd", "csrc", "mipt_libm64_tables.h")
    if not os.path.exists(libm) or os.path.getsize(libm) != 940560:
        pytest.skip("another build of libm.so.6 than Ubuntu GLIBC 2.35-0ubuntu3.x")
    work = tmp_path / "tree" / "tests" / "native"
    os.makedirs(work)
    os.makedirs(tmp_path / "tree" / "pathtracer_amd" / "csrc")
    shutil.copy(gen, work / "gen_libm64_tables.py")
    subprocess.run(["python3", str(work / "gen_libm64_tables.py"), libm], check=True, capture_output=True)
    assert open(tmp_path / "tree" / "pathtracer_amd" / "csrc" / "mipt_libm64_tables.h").read() == open(committed).read()
