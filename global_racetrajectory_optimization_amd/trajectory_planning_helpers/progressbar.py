"""Host-side shim of tph.progressbar ([REF main_globaltraj.py:462-464])."""
import sys


def progressbar(i: int, i_total: int, prefix: str = "", suffix: str = "", decimals: int = 1, length: int = 50) -> None:
    frac = i / float(i_total) if i_total else 1.0
    filled = int(length * frac)
    sys.stdout.write("\r%s |%s| %s%% %s" % (prefix, "#" * filled + "-" * (length - filled),
                                            ("{0:." + str(decimals) + "f}").format(100.0 * frac), suffix))
    if i >= i_total:
        sys.stdout.write("\n")
    sys.stdout.flush()
