"""
global_racetrajectory_optimization_amd -- MI355X-native minimum-curvature raceline QP engine.

Scope: the hot path behind trajectory_planning_helpers.opt_min_curv / iqp_handler as called
from the reference's main_globaltraj.py [REF main_globaltraj.py:264-284, 344-350], and nothing else.

  csrc/                          hand-written HIP (gfx950) kernels + the C ABI declared in include/mcq.h -> libmcq.so
  engine.py                      ctypes host binding of the C ABI (batch API, multi-GPU sharding helper)
  trajectory_planning_helpers/   the drop-in package the reference imports as `tph`
  harness.py                     runs the reference's main_globaltraj.py untouched on top of the drop-in
"""
__all__ = ["engine"]
