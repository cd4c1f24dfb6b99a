/* oracle/check_sincosf.c -- TEST INFRASTRUCTURE.  xo_sincosf (xrit_oracle.c: glibc 2.35's __sincosf_fma restated) against the
 * C library's sincosf for EVERY float with |x| < 120:   make -C oracle check_sincosf && oracle/check_sincosf
 * (2 246 049 792 arguments, ~15 core-seconds).  Prints the number of arguments whose sine or cosine differs in any bit. */
#define _GNU_SOURCE
#include "xrit_oracle.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
    const float lim = 120.0f;
    uint32_t ul;
    memcpy(&ul, &lim, 4);
    uint64_t bad = 0, total = 0;
#pragma omp parallel for reduction(+ : bad, total) schedule(dynamic, 1 << 20)
    for (int64_t i = 0; i < (int64_t)ul; i++) {
        for (int sg = 0; sg < 2; sg++) {
            const uint32_t u = (uint32_t)i | ((uint32_t)sg << 31);
            float x, s0, c0, s1, c1;
            memcpy(&x, &u, 4);
            sincosf(x, &s0, &c0);
            xo_sincosf(x, &s1, &c1);
            total++;
            if (memcmp(&s0, &s1, 4) || memcmp(&c0, &c1, 4)) {
                bad++;
                if (bad < 5) printf("x=%a libm (%a, %a) restated (%a, %a)\n", x, s0, c0, s1, c1);
            }
        }
    }
    printf("arguments %llu, differing %llu\n", (unsigned long long)total, (unsigned long long)bad);
    return bad != 0;
}
