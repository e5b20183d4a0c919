"""TomoDJCabinetTask / TomoSPCCabinetTask -- host-side mirrors of /root/reference/omniisaacgymenvs/tasks/tomodj_cabinet.py and
tasks/tomospc_cabinet.py: the FrankaCabinet skeleton (same kernels, csrc/franka_cabinet.hip) with 9 / 8 DOFs, their own
default joint positions, both finger views read correctly, and the Tomo reward variant (``dist_reward = -d``, ``rot_reward``
zeroed below 0.6 and doubled from 0.8, finger-height tests ``<=`` / ``>=``, bonus thresholds 0.1 / 0.19, done at 0.19:
tomodj_cabinet.py:564-684).  The reference's attribute names (``tomodj_dof_targets``, ``_tomodjs`` ...) alias the shared
``franka_*`` storage.  tomodj_cabinet.py also moves two debug cubes per env through USD inside get_observations (:397-422);
there is no stage here and nothing on the hot path reads them.

Not mirrored: the three ``*newhand*`` clones (five-finger keypoint observation and reward, 22-26 DOFs).
"""
from __future__ import annotations

from .. import _lib
from .franka_cabinet import FrankaCabinetTask


class _TomoCabinetTask(FrankaCabinetTask):
    _PREFIX = "tomo"
    REWARD_VARIANT = _lib.CABINET_REWARD_TOMO
    OPEN_BONUS_THRESHOLD = 0.1    # tomodj_cabinet.py:675
    OPEN_DONE_THRESHOLD = 0.19    # tomodj_cabinet.py:676, :569
    FINGER_VIEW_NAMES = ("left_fingers_view", "right_fingers_view")
    RIGHT_FINGER_FROM_RIGHT_VIEW = True  # tomodj_cabinet.py:427-429, tomospc_cabinet.py (_rfingers)

    def _alias(self, name: str):
        p = type(self)._PREFIX
        if name.startswith(p + "_"):
            return "franka_" + name[len(p) + 1:]
        if name == f"_{p}s":
            return "_frankas"
        if name == f"num_{p}_dofs":
            return "num_franka_dofs"
        return None

    def __getattr__(self, name):  # only reached when normal lookup fails
        alias = self._alias(name)
        if alias is None:
            raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")
        return getattr(self, alias)

    def __setattr__(self, name, value):
        object.__setattr__(self, self._alias(name) or name, value)


class TomoDJCabinetTask(_TomoCabinetTask):
    _PREFIX = "tomodj"
    NUM_DOFS = 9
    DEFAULT_DOF_POS = [0.52, 0.52, 0.26, -0.26, -0.25, 0.17, -0.52, 0.035, 0.035]  # tomodj_cabinet.py:112-114
    ROBOT_VIEW_NAME = "tomodj_view"


class TomoSPCCabinetTask(_TomoCabinetTask):
    _PREFIX = "tomospc"
    NUM_DOFS = 8
    DEFAULT_DOF_POS = [-0.52, 0.26, 0.79, 0.35, 1.05, 0, 0.035, 0.035]  # tomospc_cabinet.py:292-294
    ROBOT_VIEW_NAME = "tomospc_view"
