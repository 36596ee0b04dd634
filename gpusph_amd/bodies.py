"""Host-side kinematics of moving bodies with prescribed motion: the part of ProblemCore::bodies_timestep
(src/ProblemCore.cc:484-610) that does not need the Chrono rigid-body solver (MB_MOVING and MB_FORCES_MOVING bodies).

Per integrator step the problem's `moving_bodies_callback(index, t0, t1, initial_kdata, kdata)` advances the body's
kinematic data from t0 to t1 and returns the translation `dx` (3) and the rotation matrix `dr` (3x3) of that interval;
the engines then receive, per body: translation, step rotation, linear and angular velocity, and the centre of rotation as
grid cell + cell-local position (calc_grid_and_local_pos).  On the predictor the interval is [t, t + dt/2] and the
kinematic data of time t are saved; on the corrector they are restored and the interval is [t, t + dt]."""
import copy
from dataclasses import dataclass, field

import numpy as np


@dataclass
class KinematicData:
    """src/Object.h KinematicData subset: centre of rotation (global, double), linear and angular velocity"""
    crot: np.ndarray = field(default_factory=lambda: np.zeros(3))
    lvel: np.ndarray = field(default_factory=lambda: np.zeros(3))
    avel: np.ndarray = field(default_factory=lambda: np.zeros(3))


class MovingBodies:
    def __init__(self, problem, centres):
        self.problem = problem
        self.initial = [KinematicData(crot=np.array(c, dtype=np.float64)) for c in centres]
        self.kdata = copy.deepcopy(self.initial)
        self.storage = copy.deepcopy(self.initial)

    def __len__(self):
        return len(self.kdata)

    def grid_and_local(self, x):
        p = self.problem
        g = np.clip(np.floor((x - p.m_origin) / p.m_cellsize).astype(np.int64), 0, p.m_gridsize - 1)
        return g.astype(np.int32), (x - p.m_origin - (g + 0.5) * p.m_cellsize).astype(np.float32)

    def timestep(self, step, dt, t):
        """returns dict(trans (n,3) f32, rot (n,9) f32, lvel, avel, cg_grid (n,3) i32, cg_pos (n,3) f32)"""
        dt1 = dt / 2.0 if step == 1 else dt
        n = len(self)
        out = dict(trans=np.zeros((n, 3), np.float32), rot=np.zeros((n, 9), np.float32), lvel=np.zeros((n, 3), np.float32),
                   avel=np.zeros((n, 3), np.float32), cg_grid=np.zeros((n, 3), np.int32), cg_pos=np.zeros((n, 3), np.float32))
        for i in range(n):
            if step == 1:
                self.storage[i] = copy.deepcopy(self.kdata[i])
            else:
                self.kdata[i] = copy.deepcopy(self.storage[i])
            kd = self.kdata[i]
            dx, dr = self.problem.moving_bodies_callback(i, t, t + dt1, self.initial[i], kd)
            out["trans"][i] = np.asarray(dx, dtype=np.float64)
            out["rot"][i] = np.asarray(dr, dtype=np.float64).reshape(9)
            out["lvel"][i] = kd.lvel; out["avel"][i] = kd.avel
            out["cg_grid"][i], out["cg_pos"][i] = self.grid_and_local(kd.crot)
        return out
