"""Host-side shim of tph.import_veh_dyn_info -- boundary [REF main_globaltraj.py:211-213]."""
import numpy as np


def import_veh_dyn_info(ggv_import_path: str = None, ax_max_machines_import_path: str = None) -> tuple:
    ggv = None
    if ggv_import_path is not None:
        with open(ggv_import_path, "rb") as fh:
            ggv = np.loadtxt(fh, comments="#", delimiter=",")
        if ggv.ndim == 1:
            ggv = np.expand_dims(ggv, 0)
        if ggv.shape[1] != 3:
            raise RuntimeError("ggv diagram must consist of the three columns [vx, ax_max, ay_max]!")
        if np.any(ggv[:, 0] > 200.0) or np.any(ggv[:, 1:] > 50.0) or np.any(ggv < 0.0):
            raise RuntimeError("ggv seems unreasonable!")
    ax_max_machines = None
    if ax_max_machines_import_path is not None:
        with open(ax_max_machines_import_path, "rb") as fh:
            ax_max_machines = np.loadtxt(fh, comments="#", delimiter=",")
        if ax_max_machines.ndim == 1:
            ax_max_machines = np.expand_dims(ax_max_machines, 0)
        if ax_max_machines.shape[1] != 2:
            raise RuntimeError("ax_max_machines must consist of the two columns [vx, ax_max_machines]!")
        if np.any(ax_max_machines[:, 0] > 200.0) or np.any(ax_max_machines[:, 1] > 20.0) or np.any(ax_max_machines < 0.0):
            raise RuntimeError("ax_max_machines seems unreasonable!")
    return ggv, ax_max_machines
