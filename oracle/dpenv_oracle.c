/*
 * dpenv_oracle.c - CPU oracle (plain C) for the ReVolt DP env.step path.
 * TEST INFRASTRUCTURE ONLY - see dpenv_oracle.h.  Build: `make -C oracle`.
 * Compiled with -ffp-contract=off so the float build is a plain IEEE evaluation.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include "dpenv_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads used by the batched step loop (cpu_baseline leg of bench.py); returns the count in force */
int dpo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11):
 * the build-owned counter-based generator behind the reset sampler (not in the reference,
 * which uses numpy's global Mersenne Twister - quirk Q8). */
void dpo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the thrust-loss preset's numbers (tests/calibration/fit_thrust_loss_preset.py; the library holds the same: dpenv_default_vessel_ex) */
#define THRUST_LOSS_KR_STERN 0.001149
#define THRUST_LOSS_KLF_STERN 0.08173
#define THRUST_LOSS_KLR_STERN 0.05039
#define THRUST_LOSS_KL_BOW 0.0

#define REAL double
#define SFX f64
#define M_FMOD fmod
#define M_COPYSIGN copysign
#define M_ATAN2 atan2
#define M_COS cos
#define M_SIN sin
#define M_SQRT sqrt
#define M_EXP exp
#define M_FABS fabs
#define M_FLOOR floor
#define M_LOG log
#define WRAP_LITERAL 1
#include "dpenv_oracle_impl.h"
#undef REAL
#undef SFX
#undef M_FMOD
#undef M_COPYSIGN
#undef M_ATAN2
#undef M_COS
#undef M_SIN
#undef M_SQRT
#undef M_EXP
#undef M_FABS
#undef M_FLOOR
#undef M_LOG
#undef WRAP_LITERAL

#define REAL float
#define SFX f32
#define M_FMOD fmodf
#define M_COPYSIGN copysignf
#define M_ATAN2 atan2f
#define M_COS cosf
#define M_SIN sinf
#define M_SQRT sqrtf
#define M_EXP expf
#define M_FABS fabsf
#define M_FLOOR floorf
#define M_LOG logf
#define WRAP_LITERAL 0
#include "dpenv_oracle_impl.h"
