"""regenie_amd/host/fmt_g6.h -- the `%g` (six significant digits) formatter of the .loco / .prs writers -- prints what the C library prints:
tests/harness/fmt_g6_check.cpp compares it with snprintf("%g") on the boundary cases (zeros, decade edges, the fixed / scientific switch at 1e-5
and 1e6, denormals, the largest double, NaN, infinities), on exact six-digit ties and their binary scalings (round-half-even), on 1.2 million
decimal strings that end in ...5 or ...49999999999 at the seventh digit (the nearest doubles to a tie), and on random normal, log-uniform,
raw-bit-pattern and short-decimal values.  The check of 200 million values quoted in the header was run once by hand (argv[1] = 40000000)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fmt_g6_prints_what_printf_g_prints(tmp_path):
    exe = str(tmp_path / "fmt_g6_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "harness", "fmt_g6_check.cpp"), "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe, "1000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout
