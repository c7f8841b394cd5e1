"""Host-side shim of tph.side_of_line: +1 left of a->b, -1 right, 0 on the line."""
import numpy as np


def side_of_line(a, b, z) -> float:
    return float(np.sign((b[0] - a[0]) * (z[1] - a[1]) - (b[1] - a[1]) * (z[0] - a[0])))
