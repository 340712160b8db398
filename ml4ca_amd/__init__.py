"""ml4ca_amd - MI355X-native batched ReVolt dynamic-positioning environment.

The env.step hot path of simensov/ml4ca as a hand-written HIP kernel for gfx950 behind a C ABI
(include/dpenv.h, ml4ca_amd/lib/libdpenv.so), with a Gym-style Python mirror of the reference's
environment interface.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import (AOS, BF16, DONE_FAULT, DONE_TERMINAL, DONE_TIMELIMIT, F32, FINAL, FULL, LIMITED, SIMPLE, SOA,  # noqa: F401
                   DpenvError, default_vessel)
from .env import (ENVIRONMENTS, BatchedRevoltEnv, Revolt, RevoltFinal, RevoltLimited, RevoltSimple,  # noqa: F401
                  thrust_map, variant_constants)

__all__ = ['BatchedRevoltEnv', 'Revolt', 'RevoltSimple', 'RevoltLimited', 'RevoltFinal', 'ENVIRONMENTS',
           'thrust_map', 'variant_constants', 'default_vessel', 'DpenvError']
