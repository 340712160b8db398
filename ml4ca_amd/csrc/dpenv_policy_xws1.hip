// dpenv_policy_xws1.hip - the two-wave closed-loop rollout, DPENV_POLICY_F32: actor and critic in the split-f16 arithmetic
// (policy_rollout_ws_kernel<.., PREC_F32, GROUPS> of dpenv_policy_ws.h; arithmetic: mlp_eval_x in dpenv_policy_dev.h).
// Reference: rollout loop spinup/algos/tf1/ppo/ppo.py:289-322, fp32 networks core.py:29-33,80-107.
#include "dpenv_policy_ws.h"

extern "C" hipError_t dpenv_dev_launch_policy_rollout_xws_f32(const dpenv::StepArgs* a, const dpenv::PolicyArgs* pa, int mode, int ext, hipStream_t s)
{
    return dpenv_ws_launch::launch<dpenv::PREC_F32>(*a, *pa, mode, ext, s);
}
