// dpenv_policy_ws.hip - the two-wave closed-loop rollout in the f16 network arithmetic (DPENV_POLICY_F16 + DPENV_LAUNCH_TWO_WAVE):
// policy_rollout_ws_kernel<.., PREC_F16, GROUPS> of dpenv_policy_ws.h.  Its own translation unit: the instantiations take minutes.
// Reference: rollout loop spinup/algos/tf1/ppo/ppo.py:289-322, networks core.py:29-33,80-107.
#include "dpenv_policy_ws.h"

#ifdef DPENV_WS_SELFCHECK
namespace dpenv {
#include "dpenv_diag.inc"      // diagnostic builds only: pk_probe_kernel (tools/ws_pk_probe.py)
}
using namespace dpenv;
#define DPENV_DIAG_LAUNCHERS
#include "dpenv_diag.inc"
#endif

extern "C" hipError_t dpenv_dev_launch_policy_rollout_ws(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext, hipStream_t s)
{
    return dpenv_ws_launch::launch<dpenv::PREC_F16>(*a, *pa, mode, ext, s);
}
