"""csrc/epipolar_geometry.h divides by the map size with a reciprocal and one FMA correction step (div_by_const); the claim that
this is the correctly rounded IEEE quotient -- i.e. bit-identical to the reference's float32 division (multiview.py:30-35) -- is
checked here against the C compiler's division: every divisor a map size can produce, a dense sweep of numerators (all
exponents the coordinates can take, random mantissas + the mantissa patterns next to rounding boundaries)."""
import os
import subprocess
import tempfile

SRC = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline float f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static uint32_t rng = 12345u;
static inline uint32_t next(void) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; }
int main(void)
{
    long long bad = 0, n = 0;
    for (int size = 2; size <= 1024; ++size) {
        const volatile float one = 1.0f;
        const float c = (float)size, r = one / c;
        for (int e = 127 - 30; e <= 127 + 30; ++e)
            for (int t = 0; t < 3000; ++t) {
                uint32_t m = next() & 0x7fffffu;
                if (t < 64) m = (uint32_t)t;                       /* mantissas next to a power of two ... */
                else if (t < 128) m = 0x7fffffu - (uint32_t)(t - 64);   /* ... and below the next one */
                const uint32_t u = ((uint32_t)(t & 1) << 31) | ((uint32_t)e << 23) | m;
                const float a = f(u), want = a / c;
                const float q0 = a * r, er = fmaf(-q0, c, a), q = fmaf(er, r, q0);
                if (memcmp(&q, &want, 4)) ++bad;
                ++n;
            }
    }
    printf("%lld %lld\n", bad, n);
    return 0;
}
"""


def test_division_by_map_size_is_correctly_rounded():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
        bad, n = map(int, subprocess.check_output([exe]).split())
    assert n > 100_000_000 and bad == 0, (bad, n)
