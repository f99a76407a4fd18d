"""Robot constants that every caller of the landing solver passes through ``p``.

The reference evaluates a composite-rigid-body pass of its 18-body Mini-Cheetah spatial_v2 model
once, at the home pose, and feeds ``mass``, ``Ib = diag(Ic(1:3,1:3))`` and
``Ib_inv = diag(inv(Ic(1:3,1:3)))`` to the solver function
(generate_landingCtrller_IPOPT.m:100-104,222-224).  This module restates that offline pass:

* link inertias / locations: utilities_general/dynamics-utilities/get_robot_params.m:50-115 ('mc3D')
* tree: get_robot_model.m:134-234 ('quad3D'), mirrored links via flipAlongAxis (:852-889)
* composite inertia: get_mass_matrix.m:19-54
* Pluecker transforms: spatial_v2/spatial/{plux.m,rotx.m,roty.m}, spatial_v2/3D/{skew.m,rz.m}
"""
import functools

import numpy as np

HIP_SRBM = np.array([[0.19, -0.1, 0.0], [0.19, 0.1, 0.0], [-0.19, -0.1, 0.0], [-0.19, 0.1, 0.0]])  # get_robot_params.m:90-91
GRAVITY = np.array([0.0, 0.0, -9.81])  # get_robot_model.m:140
Q_LEG_HOME = np.array([0.0, -1.45, 2.65])  # generate_landingCtrller_IPOPT.m:100


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _spatial_inertia(m, com, rot):  # spatialInertia.m (3-argument form)
    c = _skew(np.asarray(com, float))
    top = np.hstack([rot + m * (c @ c.T), m * c])
    bot = np.hstack([m * c.T, m * np.eye(3)])
    return np.vstack([top, bot])


def _flip_y(I):  # get_robot_model.m:852-889 flipAlongAxis(I,'Y')
    hm = I[0:3, 3:6]
    h = 0.5 * np.array([hm[2, 1] - hm[1, 2], hm[0, 2] - hm[2, 0], hm[1, 0] - hm[0, 1]])
    Ibar = I[0:3, 0:3]
    m = I[5, 5]
    P = np.zeros((4, 4))
    P[0:3, 0:3] = 0.5 * np.trace(Ibar) * np.eye(3) - Ibar
    P[0:3, 3] = h
    P[3, 0:3] = h
    P[3, 3] = m
    X = np.diag([1.0, -1.0, 1.0, 1.0])
    P = X @ P @ X
    m, h, E = P[3, 3], P[0:3, 3], P[0:3, 0:3]
    out = np.eye(6)
    out[0:3, 0:3] = np.trace(E) * np.eye(3) - E
    out[0:3, 3:6] = _skew(h)
    out[3:6, 0:3] = _skew(h).T
    out[3:6, 3:6] = m * np.eye(3)
    return out


def _plux(E, r):  # plux.m (E,r -> X)
    return np.block([[E, np.zeros((3, 3))], [-E @ _skew(r), E]])


def _rot6(axis, q):  # rotx.m / roty.m
    c, s = np.cos(q), np.sin(q)
    if axis == "x":
        E = np.array([[1, 0, 0], [0, c, s], [0, -s, c]])
    else:
        E = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    return np.block([[E, np.zeros((3, 3))], [np.zeros((3, 3)), E]])


@functools.lru_cache(maxsize=None)
def composite_body_inertia():
    """6x6 composite rigid-body inertia Ic of the whole robot at q_home, body frame."""
    abad = _spatial_inertia(0.54, [0, 0.036, 0], 1e-6 * np.array([[381, 58, 0.45], [58, 560, 0.95], [0.45, 0.95, 444]]))
    hip = _spatial_inertia(0.634, [0, 0.016, -0.02], 1e-6 * np.array([[1983, 245, 13], [245, 2103, 1.5], [13, 1.5, 408]]))
    knee = _spatial_inertia(0.064, [0, 0, -0.061], 1e-6 * np.array([[6, 0, 0], [0, 248, 0], [0, 0, 245.0]]))
    body = _spatial_inertia(3.3, [0, 0, 0], 1e-6 * np.diag([11253.0, 36203.0, 42673.0]))
    abad_loc = np.array([0.19 * 2, 0.049 * 2, 0.0]) * 0.5
    hip_loc = np.array([0.0, 0.062, 0.0])
    knee_loc = np.array([0.0, 0.0, -0.209])
    side = np.array([[1, 1, -1, -1], [-1, 1, -1, 1], [1, 1, 1, 1]], float)
    rz_pi = np.array([[np.cos(np.pi), np.sin(np.pi), 0], [-np.sin(np.pi), np.cos(np.pi), 0], [0, 0, 1.0]])
    Ic = body.copy()
    leg_side = -1
    for leg in range(4):
        s = side[:, leg]
        links = [abad, hip, knee] if leg_side > 0 else [_flip_y(abad), _flip_y(hip), _flip_y(knee)]
        Xtree = [_plux(np.eye(3), s * abad_loc),
                 _plux(rz_pi, np.zeros(3)) @ _plux(np.eye(3), s * hip_loc),
                 _plux(np.eye(3), s * knee_loc)]
        Xup = [_rot6("x", Q_LEG_HOME[0]) @ Xtree[0], _rot6("y", Q_LEG_HOME[1]) @ Xtree[1], _rot6("y", Q_LEG_HOME[2]) @ Xtree[2]]
        I2 = links[2]
        I1 = links[1] + Xup[2].T @ I2 @ Xup[2]
        I0 = links[0] + Xup[1].T @ I1 @ Xup[1]
        Ic = Ic + Xup[0].T @ I0 @ Xup[0]
        leg_side = -leg_side
    return Ic


def robot_constants():
    """(mass, Ib[3], Ib_inv[3]) exactly as generate_landingCtrller_IPOPT.m:102-104,222-224 forms them."""
    Ic = composite_body_inertia()
    mass = float(Ic[5, 5])
    I3 = Ic[0:3, 0:3]
    return mass, np.diag(I3).copy(), np.diag(np.linalg.inv(I3)).copy()
