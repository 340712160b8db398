"""Soft pin of the BUILD-OWNED plant's low-speed hull forces: the reference's 32 recorded Cybersea station-keeping runs
(results/all_plots/dyn_pos/, a 0.2 m/s current from 16 directions, two allocators; tests/golden/cybersea_dynpos.npz written by
tests/golden/gen_cybersea_dynpos.py) through the oracle's plant (tests/calibration/dynpos_pin.py).  Not parity - Cybersea is closed, the plant
stage stays "parity unpinned" - but a bound on how far the plant's hull is from the one the reference trained on, per flow angle, and the
numbers behind DESIGN.md section 3's row."""
import os

import numpy as np

from tests.calibration import dynpos_pin as DP

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_the_records_are_station_keeping_runs():
    d = np.load(os.path.join(G, 'cybersea_dynpos.npz'))
    assert d['n'].shape == (32, 224, 3) and d['alpha'].shape == (32, 224, 3) and d['pose'].shape == (32, 224, 3)
    assert sorted(set(d['current_dir_deg'].tolist())) == sorted([-158, -135, -113, -90, -68, -45, -23, 0, 23, 45, 68, 90, 113, 135, 158, 180])
    assert (np.bincount(d['allocator']) == 16).all() and float(d['current_speed']) == np.float32(0.2)
    pose = d['pose'].astype(np.float64)
    assert np.abs(pose[:, :, :2]).max() < 1.5 and np.degrees(np.abs(pose[:, :, 2])).max() < 20.0 and np.degrees(np.abs(pose[:, :, 2].mean(1))).max() < 3.0     # on station, heading 0 on average: the CURRENT turns
    assert np.allclose(d['alpha'][:, :, 0], np.pi / 2, atol=1e-6)                                      # the bow thruster's azimuth never moves
    assert np.abs(d['n']).max() <= 100.0
    # no mean acceleration: the pose trend over the 45 s window is a few cm/s at most (0.04 m/s x 264 kg / 45 s = 0.2 N)
    t = d['t']
    v = np.array([[np.polyfit(t, pose[k, :, j], 1)[0] for j in range(2)] for k in range(32)])
    assert np.abs(v).max() < 0.04


def test_net_wrench_of_the_recorded_commands_in_the_oracle_plant_is_bounded_per_flow_angle():
    for preset, (bx, by, bn) in (('no_loss', (4.0, 5.5, 5.6)), ('thrust_loss', (2.6, 5.8, 5.1)), ('dynpos_fit', (4.0, 4.6, 4.3))):
        w = DP.wrenches(preset)
        net = np.abs(w['net'])
        assert net[:, 0].max() < bx and net[:, 1].max() < by and net[:, 2].max() < bn, (preset, net.max(0))
        rms = np.sqrt((w['net'] ** 2).mean(0))
        assert rms[0] < 1.3 and rms[1] < 3.0 and rms[2] < 2.7, (preset, rms)
        # the plant's own hull: port / starboard mirror images (flow from +b and from -b), zero sway force and moment in head and following flow
        ang, hull = w['angle'], w['hull']
        for b in (23, 45, 68, 90, 113, 135, 158):
            p, m = hull[(ang == b) & (w['allocator'] == 0)][0], hull[(ang == -b) & (w['allocator'] == 0)][0]
            assert abs(p[0] - m[0]) < 0.08 and abs(p[1] + m[1]) < 0.25 and abs(p[2] + m[2]) < 0.25, (b, p, m)    # (recorded headings differ by a degree or two)


def test_which_way_the_plant_departs_from_cybersea_at_low_speed():
    """over 16 flow angles the recorded thrust is LESS than what the plant's hull needs to hold station - in sway 0.5 (pseudo-
    inverse) to 0.9 (RL) of it, in yaw 0.1 to 0.4 -, and the ratio depends on the allocator: the two use the thrusters in different regimes, so a hull
    refit alone cannot close it (a hull property would give both allocators the same ratio)."""
    w = DP.wrenches('no_loss')
    s_all, e_all = DP.ratios(w)
    s_ps, _ = DP.ratios(w, w['allocator'] == 0)
    s_rl, _ = DP.ratios(w, w['allocator'] == 1)
    assert 0.55 < s_all[1] < 0.85 and e_all[1] < 0.08
    assert 0.40 < s_ps[1] < 0.65 and 0.70 < s_rl[1] < 1.0
    assert 0.0 < s_ps[2] < 0.3 and 0.25 < s_rl[2] < 0.55
    # the sway force the water puts on the plant's hull in a beam current of 0.2 m/s: Yv v + Yvv v|v| = 19.8 x 0.2 + 80.3 x 0.04 = 7.17 N
    beam = w['hull'][(w['angle'] == 90) & (w['allocator'] == 1)][0]
    assert abs(beam[1] - 7.17) < 0.05
