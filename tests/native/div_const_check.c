/* Exhaustive check of the constant-divisor division used by the single-reduction PCG kernels (blub_pcg1.hip.h, precond_exact):
 *     q0 = RN(y * c),  r = RN(fma(-m, q0, y)) (exact),  q = RN(fma(r, c, q0)),   c = RN(1/m), m in {3, 5, 7}  (3 and 5: the PCG preconditioner's diagonal; 7: the extrapolation's neighbour count)
 * equals the correctly rounded quotient RN(y / m) for EVERY f32 significand (one binade, both signs: scaling by powers of two
 * commutes with rounding away from the subnormal range), plus a sweep over all binades with a coarse significand stride.
 * Prints the number of mismatches; exit status 0 iff there are none.  Build: gcc -O2 -mfma -ffp-contract=off */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float div_const(float y, float m, float c) {
    const float q0 = y * c;
    const float r = fmaf(-m, q0, y);
    return fmaf(r, c, q0);
}

int main(void) {
    const float ms[3] = {3.0f, 5.0f, 7.0f};
    const float cs[3] = {0x1.555556p-2f, 0x1.99999ap-3f, 0x1.24924ap-3f};
    unsigned long long bad = 0, tested = 0;
    for (int k = 0; k < 3; ++k) {
        if (cs[k] != 1.0f / ms[k]) { printf("constant %d is not the rounded reciprocal\n", k); return 2; }
        for (uint32_t sig = 0; sig < (1u << 23); ++sig)             /* every significand of [1, 2) and [-2, -1) */
            for (uint32_t sign = 0; sign < 2; ++sign) {
                const float y = from_bits((sign << 31) | (127u << 23) | sig);
                bad += div_const(y, ms[k], cs[k]) != y / ms[k];
                ++tested;
            }
        for (uint32_t e = 4; e < 252; ++e)                          /* all binades whose quotient and remainder stay normal */
            for (uint32_t sig = 0; sig < (1u << 23); sig += 4099) {
                const float y = from_bits((e << 23) | sig);
                bad += div_const(y, ms[k], cs[k]) != y / ms[k];
                ++tested;
            }
    }
    printf("%llu tested, %llu mismatches\n", tested, bad);
    return bad != 0;
}
