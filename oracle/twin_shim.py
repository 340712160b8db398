"""TEST INFRASTRUCTURE (oracle/): a single-env plant object with the reference's plant plug-in seam, backed by the
oracle's float64 3-DOF plant.

The reference env never talks to the simulator directly: it is handed a duck-typed object with
    val(module, feature, val=None, report=False)   read (val is None) or write one simulator feature
    step(steps)                                    advance the 100 Hz simulation `steps` sub-steps
(specific/digitwin.py:50-114, 213-219; injected at specific/customEnv.py:22-36).  TwinShim offers exactly that
seam over dpo_plant, so the reference's own Revolt* classes can be run end to end (imported behind gym/keras stubs by
tests/golden/gen_closedloop.py) around the SAME plant the oracle and the HIP kernel integrate.  Those closed-loop
trajectories are the fixtures tests/golden/closedloop_*.npz: they pin the composition decode -> command write ->
plant -> three plant reads -> observation / reward / termination -> late new_ref against the reference's Python,
with a plant that actually moves (the scripted-plant fixtures of tools/gen_golden.py hold it still).

Features understood (the set the env touches, customEnv.py:47-61,117-122,159-175, misc/simtools.py:52-59):
  reads   Hull.Eta (6), Hull.Nu (6), Hull.Yaw
  writes  THR{1,2,3}.ThrustOrTorqueCmdMtc (% thrust: bow, port, starboard), THR{1,2,3}.AzmCmdMtc (rad),
          Hull.PosNED, Hull.PosAttitude, Hull.VelocityNu (latched, applied while Hull.StateResetOn = 1),
          Hull.StateResetOn, THR1.LinActuator, THRi.MtcOn (accepted, no dynamics behind them)
Unknown features print and return None, as the reference's DigiTwin does (digitwin.py:71-76).
"""
import copy

import numpy as np

from oracle import oracle as O


class TwinShim(object):
    def __init__(self, vessel=None, substep_dt=0.01, current=None):
        self._cfg = O.make_config(substep_dt=substep_dt)
        self._orc = O.Oracle(self._cfg, np.float64, vessel=vessel)
        self.vessel = self._orc.vessel.copy()
        self.eta = np.zeros(3)           # N, E, psi
        self.nu = np.zeros(3)            # u, v, r
        self.thrust = np.zeros(3)        # % : THR1 bow, THR2 port, THR3 starboard (customEnv.py:48-50)
        self.azimuth = np.array([np.pi / 2, 0.0, 0.0])
        self.current = None if current is None else np.asarray(current, np.float64)   # (vcN, vcE) in NED
        self._latched = {}
        self._reset_on = 0
        self.n_substeps_run = 0
        self.log = []

    # ---- the seam ----------------------------------------------------------------------------------------
    def val(self, module, feat, val=None, report=False):
        if val is None:
            if module == 'Hull' and feat == 'Eta':
                return [float(self.eta[0]), float(self.eta[1]), 0.0, 0.0, 0.0, float(self.eta[2])]
            if module == 'Hull' and feat == 'Nu':
                return [float(self.nu[0]), float(self.nu[1]), 0.0, 0.0, 0.0, float(self.nu[2])]
            if module == 'Hull' and feat == 'Yaw':
                return float(self.eta[2])
            print('TwinShim: unknown feature %s.%s' % (module, feat))
            return None
        self.log.append((module, feat, copy.copy(val)))
        if module in ('THR1', 'THR2', 'THR3'):
            k = int(module[3]) - 1
            if feat == 'ThrustOrTorqueCmdMtc':
                self.thrust[k] = float(val)
            elif feat == 'AzmCmdMtc':
                self.azimuth[k] = float(val)
            elif feat not in ('MtcOn', 'LinActuator'):
                print('TwinShim: unknown feature %s.%s' % (module, feat))
        elif module == 'Hull':
            if feat in ('PosNED', 'PosAttitude', 'VelocityNu'):
                self._latched[feat] = list(val)
            elif feat == 'StateResetOn':
                self._reset_on = int(val)
            else:
                print('TwinShim: unknown feature %s.%s' % (module, feat))
        else:
            print('TwinShim: unknown feature %s.%s' % (module, feat))
        return None

    def step(self, steps=1):
        steps = int(steps)
        if steps < 1:
            print('TwinShim: bad step count')       # digitwin.py:219
            return
        self.log.append(('step', '', steps))
        if self._reset_on:
            # models are held at the latched initial values while the reset flag is up (customEnv.py:164-167)
            if 'PosNED' in self._latched:
                self.eta[0], self.eta[1] = self._latched['PosNED'][0], self._latched['PosNED'][1]
            if 'PosAttitude' in self._latched:
                self.eta[2] = self._latched['PosAttitude'][2]
            if 'VelocityNu' in self._latched:
                v6 = self._latched['VelocityNu']
                self.nu[:] = [v6[0], v6[1], v6[5]]
            self._latched = {}
            return
        self._cfg.n_substeps = steps
        self.eta, self.nu = self._orc.plant(self.eta, self.nu, self.thrust, self.azimuth, current=self.current)
        self.n_substeps_run += steps
