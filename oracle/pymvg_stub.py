"""Stand-in for the `pymvg` package -- TEST INFRASTRUCTURE ONLY (used by tests/golden/make_triangulation_golden.py).

The reference's test-time lifting (`vision/triangulation.py:400-441` -> `vision/multi_camera_system.py:199-225`) goes
through its vendored copies of pymvg's `CameraModel` / `MultiCameraSystem`, which import helper functions from the
`pymvg` package itself (requirements.txt:11, version not pinned; pymvg 2.x at the time of the repository) -- absent
from this image and not installable offline.  Only three of those helpers are executed on the path
`build_multi_camera_system` -> `CameraModel._from_parts` -> `find3d`:

  * `pymvg.quaternions.quaternion_from_matrix` / `quaternion_matrix` -- the published algorithm of C. Gohlke's
    transformations.py in pymvg's (x, y, z, w) component order, restated below;
  * `pymvg.util._undistort` -- OpenCV's five fixed-point iterations of the Brown model (the reference always passes
    zero distortion coefficients, triangulation.py:363-367: the iteration is then the identity);
  * `pymvg.ros_compat.sensor_msgs.msg.CameraInfo` -- a plain attribute bag.

Every other imported name resolves to a placeholder that raises when called.  The vendored reference modules also
spell numpy < 1.20 / < 2.0 names (`np.float`, `np.alltrue`, `np.array(x, copy=False)` meaning "copy only if needed");
`numpy_legacy_proxy()` supplies those to the reference modules only -- numpy itself is not patched.
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np

_EPS = np.finfo(float).eps * 4.0


def quaternion_matrix(quaternion):
    """Homogeneous rotation matrix of a quaternion (x, y, z, w)."""
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < _EPS:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array((
        (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
        (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
        (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
        (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)


def quaternion_from_matrix(matrix):
    """Quaternion (x, y, z, w) of a homogeneous rotation matrix."""
    q = np.empty((4,), dtype=np.float64)
    M = np.asarray(matrix, dtype=np.float64)[:4, :4]
    t = np.trace(M)
    if t > M[3, 3]:
        q[3] = t
        q[2] = M[1, 0] - M[0, 1]
        q[1] = M[0, 2] - M[2, 0]
        q[0] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    return q


def _undistort(xd, yd, D):
    """OpenCV undistortPoints: five fixed-point iterations of the Brown model (k1, k2, p1, p2, k3)."""
    xd = np.asarray(xd, dtype=np.float64)
    yd = np.asarray(yd, dtype=np.float64)
    x, y = xd.copy(), yd.copy()
    k1, k2, t1, t2, k3 = [float(v) for v in D[:5]]
    for _ in range(5):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2)
        delta_x = 2.0 * t1 * x * y + t2 * (r2 + 2.0 * x * x)
        delta_y = t1 * (r2 + 2.0 * y * y) + 2.0 * t2 * x * y
        x = (xd - delta_x) * icdist
        y = (yd - delta_y) * icdist
    return x, y


class _Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _placeholder_module(name, **real):
    mod = types.ModuleType(name)
    mod.__dict__.update(real)

    def __getattr__(attr):                       # PEP 562: `from pymvg.util import anything` succeeds
        if attr.startswith("__"):
            raise AttributeError(attr)

        def unavailable(*a, **k):
            raise NotImplementedError("%s.%s is not part of the pymvg stand-in (oracle/pymvg_stub.py)" % (name, attr))

        return unavailable

    mod.__getattr__ = __getattr__
    return mod


def install():
    """Register the stand-in under the name `pymvg` (no-op if a real pymvg is importable)."""
    try:
        import pymvg  # noqa: F401

        return False
    except ImportError:
        pass
    pkg = _placeholder_module("pymvg")
    pkg.__path__ = []
    util = _placeholder_module("pymvg.util", _undistort=_undistort, Bunch=_Bag,
                               is_string=lambda v: isinstance(v, str))
    quat = _placeholder_module("pymvg.quaternions", quaternion_matrix=quaternion_matrix,
                               quaternion_from_matrix=quaternion_from_matrix)
    align = _placeholder_module("pymvg.align")
    msgs = types.ModuleType("pymvg.ros_compat.sensor_msgs")
    msgs.msg = _Bag(CameraInfo=_Bag)
    ros = _placeholder_module("pymvg.ros_compat", sensor_msgs=msgs)
    for m in (pkg, util, quat, align, ros, msgs):
        sys.modules[m.__name__] = m
    pkg.util, pkg.quaternions, pkg.align, pkg.ros_compat = util, quat, align, ros
    return True


def numpy_legacy_proxy():
    """A module object that behaves like numpy with the pre-2.0 names the vendored pymvg files use."""
    proxy = types.ModuleType("numpy_legacy_proxy")

    def array(obj, *a, **k):
        if k.get("copy") is False:               # numpy < 2: "copy only if needed"
            k.pop("copy")
            return np.asarray(obj, *a, **k)
        return np.array(obj, *a, **k)

    legacy = dict(array=array, float=float, int=int, bool=bool, alltrue=np.all)

    def __getattr__(attr):
        if attr in legacy:
            return legacy[attr]
        return getattr(np, attr)

    proxy.__getattr__ = __getattr__
    return proxy
