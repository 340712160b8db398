#!/usr/bin/env python3
"""Generate tests/golden/iae.npz by importing the reference's own metric code.

Runs ONLY in the build container (needs /root/reference); start as ``python3 -B tools/gen_golden_iae.py``.

Imported from the reference and executed as-is (behind a stub for matplotlib, which is absent here and which the two functions
never touch):  results/all_plots/common.py  ->  IAE (:60-74), absolute_error (:57-58), get_secondly_averages (:37-55).
Input data: the recorded Cybersea box test of the RL allocator, results/all_plots/box_test/bagfile__RL_{observer_eta_ned,
reference_filter_state_desired}.csv, prepared the way box_test/plot_pos.py does it (offset by the first reference sample :33-46,
per-second averages :120-135, normalisation [5, 5, 25] :174, times shifted by -1 s :172).
Only data is written: the averaged series (inputs) and IAE's two return values (expected outputs)."""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF = '/root/reference/results/all_plots'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'iae.npz')

for name in ('matplotlib', 'matplotlib.pyplot', 'matplotlib.gridspec'):
    m = types.ModuleType(name)
    m.rcParams = {}
    sys.modules[name] = m
sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
sys.modules['matplotlib'].gridspec = sys.modules['matplotlib.gridspec']
sys.path.insert(0, REF)
import common   # noqa: E402  (the reference's module)

box = os.path.join(REF, 'box_test')
ref_data = np.genfromtxt(os.path.join(box, 'bagfile__RL_reference_filter_state_desired.csv'), delimiter=',')
pos_data = np.genfromtxt(os.path.join(box, 'bagfile__RL_observer_eta_ned.csv'), delimiter=',')
n0, e0 = ref_data[1:, 1:2][0, 0], ref_data[1:, 2:3][0, 0]
ref_series = [ref_data[1:, 1:2] - n0, ref_data[1:, 2:3] - e0, ref_data[1:, 3:4]]
ref_time = ref_data[1:, -1:]
pos_series = [pos_data[1:, 1:2] - n0, pos_data[1:, 2:3] - e0, pos_data[1:, 6:7]]
pos_time = pos_data[1:, 7:]

ref_avg = [common.get_secondly_averages(ref_time, d) for d in ref_series]
pos_avg = [common.get_secondly_averages(pos_time, d.reshape(d.shape[0],).tolist()) for d in pos_series]
refs = np.array([d for (_, d) in ref_avg]).T            # [seconds, 3]: N, E [m], yaw [deg]
etas = np.array([d for (_, d) in pos_avg]).T
m = min(len(refs), len(etas))
refs, etas = refs[:m], etas[:m]
times = (np.array(ref_avg[0][0]) - 1.0)[:m]
norm = np.array([5.0, 5.0, 25.0])
integrals, cumsum = common.IAE(etas / norm, refs / norm, times.tolist())
# a second case with an irregular time base and synthetic series (exercises the trapezoid with varying dt)
rng = np.random.RandomState(0)
t2 = np.cumsum(rng.uniform(0.05, 0.4, size=200))
a2, b2 = rng.normal(size=(200, 3)), rng.normal(size=(200, 3))
i2, c2 = common.IAE(a2, b2, t2.tolist())
np.savez(OUT, eta=etas, ref=refs, time=times, norm=norm, integrals=np.array(integrals), cumsum=np.array(cumsum),
         t2=t2, a2=a2, b2=b2, integrals2=np.array(i2), cumsum2=np.array(c2),
         abs_err_probe=np.array([common.absolute_error(np.array([1.0, 2.0, 3.0]), np.array([0.5, -1.0, 3.5]))]))
print('IAE of the recorded RL box test (reference code): %.4f over %d s; wrote %s' % (cumsum[-1], m, OUT))
