"""GPU tests of the runtime contract of the C ABI: stream ordering, HIP-graph capture, independent handles,
large batches, lifetime, argument checking."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests import tolerances as TOL

pytestmark = pytest.mark.gpu


def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_steps_are_stream_ordered_and_graph_capturable():
    """reset/step/rollout do no allocation and no host sync (include/dpenv.h): they run on a side stream and can be
    captured into a HIP graph whose replay reproduces eager execution bit for bit."""
    torch = torch_()
    n, K = 4096 + 9, 12
    rng = np.random.RandomState(2)
    e1, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=16, seed=3)
    e2, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=16, seed=3)
    acts = H.to_dev(rng.normal(0, 0.8, size=(K, n, 7)).astype(np.float32))
    st = H.to_dev(H.random_state(rng, n, spread=0.4))
    ctr = H.to_dev(np.zeros((2, n), np.int32))
    e1.set_state(st, ctr)
    e2.set_state(st, ctr)
    eager = []
    for k in range(K):
        o, r, d, _ = e1.step(acts[k])
        eager.append((o.clone(), r.clone(), d.clone()))
    s1, c1 = e1.get_state()
    # capture the same K steps on a side stream, writing into per-step output rows
    obs = torch.empty((K, n, 9), device=e2.device)
    rew = torch.empty((K, n), device=e2.device)
    done = torch.empty((K, n), dtype=torch.uint8, device=e2.device)
    side = torch.cuda.Stream(device=e2.device)
    side.wait_stream(torch.cuda.current_stream(e2.device))
    with torch.cuda.stream(side):
        e2.step(acts[0], out=(obs[0], rew[0], done[0]))        # warm up the launch path outside capture
    torch.cuda.current_stream(e2.device).wait_stream(side)
    torch.cuda.synchronize()
    e2.set_state(st, ctr)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(K):
            e2.step(acts[k], out=(obs[k], rew[k], done[k]))
    e2.set_state(st, ctr)                                      # capture executed nothing: state still the start state
    g.replay()
    torch.cuda.synchronize()
    for k in range(K):
        assert torch.equal(obs[k], eager[k][0]) and torch.equal(rew[k], eager[k][1]) and torch.equal(done[k], eager[k][2]), k
    s2, c2 = e2.get_state()
    assert torch.equal(s1, s2) and torch.equal(c1, c2)
    # replaying again continues from the new state (the graph holds pointers, not values)
    g.replay()
    torch.cuda.synchronize()
    s3, _ = e2.get_state()
    assert not torch.equal(s3, s2)


def test_handles_are_independent_and_survive_recreation():
    torch = torch_()
    n = 1000
    rng = np.random.RandomState(5)
    act = H.to_dev(H.random_actions(rng, n, 7))
    outs = []
    for rep in range(6):                                        # create / use / destroy repeatedly
        ea, _ = H.make_pair('final_cont', n, seed=1)
        eb, _ = H.make_pair('limited', n, seed=2)               # a second live handle of a different variant
        ea.reset()
        eb.reset()
        o, r, d, _ = ea.step(act)
        eb.step(H.to_dev(H.random_actions(rng, n, 5)))
        outs.append((o.clone(), r.clone()))
        ea.close()
        eb.close()
        ea.close()                                              # closing twice is harmless
    for o, r in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(r, outs[0][1])


def test_one_million_envs():
    """16x the benchmark batch in one launch: against the OpenMP oracle on every env."""
    torch = torch_()
    n = 1 << 20
    rng = np.random.RandomState(8)
    env, orc = H.make_pair('final_cont', n)
    O.set_threads(16)
    st = H.random_state(rng, n, spread=0.7)
    ctr = np.zeros((2, n), np.int32)
    act = H.random_actions(rng, n, 7)
    env.set_state(H.to_dev(st), H.to_dev(ctr))
    obs, rew, done, _ = env.step(H.to_dev(act))
    ost, octr = st.copy(), ctr.copy()
    oobs = np.zeros((n, 9), np.float32)
    orew = np.zeros(n, np.float32)
    odone = np.zeros(n, np.uint8)
    orc.step_into(ost, octr, act, oobs, orew, odone)
    TOL.assert_close(obs.cpu().numpy(), oobs, TOL.OBS_FLOOR, what='obs at 2^20 envs')
    TOL.assert_close(rew.cpu().numpy(), orew, TOL.REWARD_FLOOR, what='reward at 2^20 envs')
    assert (done.cpu().numpy() != odone).mean() < 1e-4
    st2, _ = env.get_state()
    TOL.assert_close(st2.cpu().numpy()[0:3].T, ost[0:3].T, TOL.ETA_FLOOR, what='eta at 2^20 envs')


def test_argument_checking_is_loud():
    import ml4ca_amd
    torch = torch_()
    env, _ = H.make_pair('final_cont', 64)
    good = torch.zeros((64, 7), device=env.device)
    with pytest.raises(ValueError):
        env.step(torch.zeros((64, 6), device=env.device))                       # wrong act_dim
    with pytest.raises(ValueError):
        env.step(good.cpu())                                                    # host tensor
    with pytest.raises(ValueError):
        env.step(good.double())                                                 # wrong dtype
    with pytest.raises(ValueError):
        env.step(torch.zeros((7, 64), device=env.device).t())                   # not contiguous
    with pytest.raises(ValueError):
        env.step(good, new_ref=torch.zeros((64, 3), device=env.device))         # new_ref is [3, n]
    with pytest.raises(ValueError):
        env.reset(init=torch.zeros((64, 6), device=env.device))
    with pytest.raises(ml4ca_amd.DpenvError):
        env.set_current(torch.zeros(64, device=env.device), torch.zeros(64, device=env.device))   # current not enabled
    with pytest.raises(ml4ca_amd.DpenvError):
        ml4ca_amd.BatchedRevoltEnv(64, variant='simple', extended_state=True)   # IndexError in the reference too
    with pytest.raises(ml4ca_amd.DpenvError):
        ml4ca_amd.BatchedRevoltEnv(64, vessel_params=np.zeros(32, np.float32))  # singular mass matrix
    env.step(good)                                                              # and the handle still works


def test_soak_262_million_env_steps():
    """65 536 envs x 4 000 steps of the fused rollout with auto-reset, drifting current and Gaussian actions: no
    fault bit, nothing non-finite, every observation of a running env inside the termination bounds, episode
    bookkeeping consistent with the time limit."""
    torch = torch_()
    n, chunk, reps = 65536, 100, 40
    env, _ = H.make_pair('final_cont', n, auto_reset=True, seed=17, current=True, current_drift=True)
    env.set_current(torch.full((n,), 0.2, device=env.device), torch.full((n,), float(np.deg2rad(135)), device=env.device))
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(4)
    acts = (torch.randn((chunk, n, 7), generator=g, device=env.device) * 0.6065).contiguous()
    b = torch.tensor(env.real_ss_bounds, device=env.device)
    n_end = torch.zeros((), dtype=torch.int64, device=env.device)
    for r in range(reps):
        obs, rew, done = env.rollout(acts)
        assert not bool((done & 4).any()), 'fault bit raised'
        assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
        n_end += (done != 0).sum()
        # rows returned after a finished step are reset observations: inside the sampled box; the others were not terminal
        assert bool((obs[..., :6].abs() <= b * 1.0001)[done == 0].all())
        assert float(rew.max()) <= 3.5 + 1e-5
    st, ctr = env.get_state()
    ctr = ctr.cpu().numpy()
    T = chunk * reps
    assert (ctr[0] < env.max_ep_len).all() and (ctr[0] >= 0).all()
    assert (ctr[1] >= 1 + T // env.max_ep_len).all()           # at least one reset per time limit
    assert int(n_end) == int(ctr[1].sum()) - n                  # every finished episode was re-sampled exactly once
    vc, beta = env.get_current()
    assert 0.1 < float(vc.mean()) < 0.3 and float(vc.std()) < 0.05


def test_reset_actions_option():
    """ENV:30,179-188 (--reset_acts): episodes start with previous thrust clip(N(0, 0.1) * 100); thrust only, angles at
    their defaults; the reset observation carries it; masked resets leave the other envs alone."""
    import ml4ca_amd
    from ml4ca_amd import _lib
    torch = __import__('torch')
    n = 4096
    env = ml4ca_amd.BatchedRevoltEnv(n, reset_acts=True, seed=5)
    obs = env.reset()
    st, _ = env.get_state()
    pt = st[_lib.S['PT_BOW']:_lib.S['PT_BOW'] + 3]
    assert abs(float(pt.mean())) < 1.0 and 9.0 < float(pt.std()) < 11.0 and float(pt.abs().max()) <= 100.0
    assert torch.allclose(obs[:, 6:9], (pt / 100.0).t())
    assert torch.equal(st[_lib.S['A_BOW']], torch.full((n,), float(np.pi / 2), device=env.device))
    mask = torch.zeros(n, dtype=torch.uint8, device=env.device)
    mask[: n // 2] = 1
    env.reset(mask=mask)
    st2, _ = env.get_state()
    pt2 = st2[_lib.S['PT_BOW']:_lib.S['PT_BOW'] + 3]
    assert torch.equal(pt2[:, n // 2:], pt[:, n // 2:]) and not torch.equal(pt2[:, : n // 2], pt[:, : n // 2])
    # the draw is made inside the kernels (Philox keyed by seed, global env id, episode): same seed -> same thrust, and the
    # in-kernel auto-reset applies it too
    env_b = ml4ca_amd.BatchedRevoltEnv(n, reset_acts=True, seed=5)
    env_b.reset()
    st_b, _ = env_b.get_state()
    assert torch.equal(st_b[_lib.S['PT_BOW']:_lib.S['PT_BOW'] + 3], pt)
    env_c = ml4ca_amd.BatchedRevoltEnv(n, reset_acts=True, auto_reset=True, max_ep_len=20, seed=5, terminate=False)
    env_c.reset()
    zero = torch.zeros((n, 7), device=env_c.device)
    for _ in range(10):                                       # max_ep_len 20 -> 10 agent steps per episode
        o, r, d, _ = env_c.step(zero)
    assert bool((d & 2).all())
    st_c, ctr_c = env_c.get_state()
    pt_c = st_c[_lib.S['PT_BOW']:_lib.S['PT_BOW'] + 3]
    assert int(ctr_c[1].min()) == 2 and 9.0 < float(pt_c.std()) < 11.0 and not torch.equal(pt_c, pt)
    assert torch.allclose(o[:, 6:9], (pt_c / 100.0).t())


def test_two_wave_policy_rollout_full_size_soak():
    """65 536 envs (an env wave and a network wave on EVERY SIMD of the chip), auto-reset, drifting current, bf16 rows:
    ten launches of 50 steps in the two-wave form against the one-wave form, continued from each other's final state -
    every block identical bit for bit, everything finite.  Odd launches draw the exploration noise in the kernel."""
    import ml4ca_amd
    from ml4ca_amd.policy import ActorCritic, policy_rollout
    torch = __import__('torch')
    n, T = 65536, 50
    envs = {}
    for form in ('one_wave', 'two_wave'):
        e = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, seed=3, current=True, current_drift=True, obs_dtype='bfloat16')
        ActorCritic(9, 7, (80, 80, 80), seed=4, device=e.device).upload(e, launch_form=form)
        e.set_current(torch.full((n,), 0.2, device=e.device), torch.full((n,), 2.356, device=e.device))
        e.reset()
        envs[form] = e
    g = torch.Generator(device=envs['one_wave'].device).manual_seed(9)
    for launch in range(10):
        noise = torch.randn((T, n, 7), generator=g, device=envs['one_wave'].device) if launch % 2 == 0 else None
        outs = {form: policy_rollout(envs[form], T, noise=noise, sample=True) for form in envs}
        for k in outs['one_wave']:
            a, b = outs['one_wave'][k], outs['two_wave'][k]
            assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b), (launch, k)
        assert bool(torch.isfinite(outs['two_wave']['rew']).all()) and bool(torch.isfinite(outs['two_wave']['val']).all())
        assert not bool((outs['two_wave']['done'] & 4).any())
    s0, c0 = envs['one_wave'].get_state()
    s1, c1 = envs['two_wave'].get_state()
    assert torch.equal(s0, s1) and torch.equal(c0, c1)
    assert int(c1[1].min()) >= 1                    # every env went through at least one in-kernel reset


def test_bench_line_with_every_record_at_reduced_size():
    """bench.py end to end on one GPU at reduced size: the ONE stdout line (contract keys, roofline, compact cpu_baseline; under 4 KB, and
    stdout + stderr together under the 8 KB tail the driver keeps) and the side records in bench_side.json: cpu_baseline with its legs and the
    per-quantity error ledger, the side legs, the config-4 record (every leg present, no error) and the vessel-class record.  The N > 1 runs of the driver
    execute exactly this code plus the collectives (rehearsed over gloo in tests/test_host_cpu.py and profiles/r03c_rehearsal_*)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    side_path = os.path.join(root, 'gpurun_out', 'bench_side_test.json') if os.path.isdir(os.path.join(root, 'gpurun_out')) else '/tmp/bench_side_test.json'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--envs', '8192', '--steps', '100', '--warmup', '50', '--cpu-seconds', '1.5',
                        '--config4', '1', '--config4-envs', '4096', '--classes', '3', '--side-json', side_path], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    # the driver keeps an 8 KB tail of stdout + stderr: the line is built for <= 4 KB (bench.py asserts < 8 KB itself), stderr stays short
    assert len(lines[0]) < 4096, len(lines[0])
    assert len(p.stdout) + len(p.stderr) < 8192, (len(p.stdout), len(p.stderr), p.stderr[-2000:])
    r = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in r, k
    assert set(r) <= {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'repeats', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                      'dtype', 'data', 'config', 'roofline', 'roofline_valu', 'cpu_baseline', 'side_records'}, set(r)
    assert r['n_gpus'] == 1 and r['steps'] == 100 and r['warmup'] == 50 and r['dtype'] == 'f32' and r['vs_baseline'] is None
    assert abs(r['value'] - 8192 / (r['ms_per_step'] * 1e-3)) < 1e-6 * r['value'] and 'workload' in r['config']
    rf = r['roofline']
    assert rf['bound'] == 'hbm' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and rf['achieved'] < rf['peak']
    cb = r['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] >= cb['value_1thread'] > 0 and 'sample' in cb
    led = cb['gpu_vs_cpu']
    assert led['done_mismatches_elsewhere'] == 0 and led['max_rel_err_obs'] < 1e-5 and led['max_rel_err_reward'] < 1e-5
    assert any(l.startswith('bench.py: headline: {') for l in p.stderr.splitlines())
    # everything else: the side records
    s = json.load(open(side_path))
    full = s['cpu_baseline_full']
    assert full['value'] == cb['value'] == max(l['env_steps_per_s'] for l in full['legs'])
    assert set(full['gpu_vs_cpu']['per_quantity']) >= {'obs.x~', 'obs.u', 'reward', 'reward.pos'}
    assert s['group']['world_size'] == 1 and len(s['per_rank']['wall_s']) == 1 and 'math' in s['headline_notes']
    assert 'side_legs_error' not in s, s.get('side_legs_error')
    for leg in ('eager_loop', 'fused_rollout', 'policy_rollout', 'config5_ppo_rollout', 'multi_handle'):
        assert leg in s, leg
    assert 'two_wave' in s['policy_rollout']['policy_dtype_f32']['launch_form']
    assert 'error' not in s['multi_handle'] and s['multi_handle']['at_metric_size']['envs_total'] == 8192
    c4 = s['config4']
    assert 'error' not in c4, c4
    for leg in ('step_only', 'fused_rollout', 'closed_loop', 'exchange_76B', 'episode_plus_sync_exchange_76B',
                'episode_with_previous_exchange_in_flight_76B', 'exchange_compact', 'summary'):
        assert leg in c4, leg
    assert c4['envs_per_rank'] == 4096 and c4['exchange_compact']['alone']['bytes_per_env_step'] == 58
    assert set(c4['closed_loop']) >= {'policy_dtype_f16', 'policy_dtype_f32_actor', 'policy_dtype_f32'}
    vc = s['vessel_classes']
    assert {'classes_1', 'classes_3', 'classes_16'} <= set(vc) and vc['classes_3']['step_us'] > 0


def test_bench_two_ranks_started_by_bench_itself_on_one_gpu():
    """The N > 1 bench line end to end, as far as one GPU allows: `python bench.py --gpus 2` with no launcher around it starts its two
    ranks as a child process, both use this GPU (--same-device) and talk over gloo; the line must carry the headline (2 x envs), the
    process-group record (2 ranks, their pids) and every leg of the config-4 record, collectives included.  What the driver's N > 1 runs
    add is RCCL instead of gloo and a GPU per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    side_path = os.path.join(root, 'gpurun_out', 'bench_side_test2.json') if os.path.isdir(os.path.join(root, 'gpurun_out')) else '/tmp/bench_side_test2.json'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--same-device', '--envs', '4096',
                        '--steps', '20', '--warmup', '5', '--config4-envs', '1024', '--side-json', side_path], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['total_envs'] == 8192 and r['scaling'] == 'weak' and r['cpu_baseline'] is None
    assert r['config']['backend'] == 'gloo' and 'roofline' in r
    assert abs(r['value'] - 8192 / (r['ms_per_step'] * 1e-3)) < 1e-6 * r['value']
    s = json.load(open(side_path))
    assert s['group']['world_size'] == 2 and s['group']['backend'] == 'gloo' and len({x['pid'] for x in s['group']['ranks']}) == 2
    assert len(s['per_rank']['wall_s']) == 2 and s['per_rank']['ms_per_step_max'] >= s['per_rank']['ms_per_step_min'] > 0
    assert 'fused_rollout' not in s                               # the single-GPU side legs stay home when there is more than one rank
    c4 = s['config4']
    assert 'error' not in c4, c4
    assert c4['ranks'] == 2 and c4['total_envs'] == 2048
    for leg in ('step_only', 'fused_rollout', 'closed_loop', 'exchange_76B', 'episode_plus_sync_exchange_76B',
                'episode_with_previous_exchange_in_flight_76B', 'exchange_compact'):
        assert leg in c4, leg
    assert c4['exchange_76B']['recv_GBps_per_rank'] > 0 and c4['exchange_compact']['alone']['recv_GBps_per_rank'] > 0
    assert {'chunks_1', 'chunks_4', 'chunks_8'} <= set(c4['exchange_compact'])


def test_bench_side_records_cannot_cost_the_headline_line():
    """bench.py --side-timeout: a per-rank watchdog ends a run whose side records do not finish (a hung collective of the config-4 record on a node
    it was never run on) with the exit code of a successful run - the ONE stdout line is already out, stderr says what happened."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--envs', '8192', '--steps', '20', '--warmup', '5', '--cpu-seconds', '0.6',
                        '--side-timeout', '0.2', '--side-json', '/tmp/bench_side_watchdog.json'], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and 'roofline' in json.loads(lines[0]) and 'cpu_baseline' in json.loads(lines[0])
    assert 'side records still running after' in p.stderr and 'exit code 0' in p.stderr
