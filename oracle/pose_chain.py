"""ctypes face of oracle/pose_chain.cpp: the hand pose chain of the fitting loops (fitting_single.py:206-226) on the CPU
in double precision, with its exact Jacobian.  TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline leg)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libpose_chain_oracle.so')
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError('%s is missing: run `make -C oracle` (or __graft_entry__.build())' % _LIB)
        _lib = ctypes.CDLL(_LIB)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.oracle_pose_chain.restype = ctypes.c_int
        _lib.oracle_pose_chain.argtypes = [dp, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, dp]
    return _lib


def pose_chain(ori_pose, bone_len, params, is_right=True, want_jac=True):
    """ori_pose [F,21,3] (MANO order), bone_len [F,20], params [F,36] = [joint_refine_angle 20 | palm_refine_angle 7 |
    palm_rot_refine 6 | palm_trans_refine 3] -> bone_transformation_inv [F,21,4,4], joint_3d [F,21,3], jac [F,399,36]."""
    lib = _load()
    a = np.ascontiguousarray(ori_pose, dtype=np.float64).reshape(-1, 21, 3)
    b = np.ascontiguousarray(bone_len, dtype=np.float64).reshape(-1, 20)
    p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 36)
    F = a.shape[0]
    assert b.shape[0] == F and p.shape[0] == F
    bt, j3 = np.empty((F, 21, 4, 4)), np.empty((F, 21, 3))
    jac = np.empty((F, 399, 36)) if want_jac else None
    dp = ctypes.POINTER(ctypes.c_double)
    q = lambda x: x.ctypes.data_as(dp) if x is not None else None
    rc = lib.oracle_pose_chain(q(a), q(b), 1 if is_right else 0, q(p), F, q(bt), q(j3), q(jac))
    assert rc == 0
    return bt, j3, jac
