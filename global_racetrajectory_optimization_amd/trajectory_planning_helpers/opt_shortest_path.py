"""tph.opt_shortest_path [REF main_globaltraj.py:287-290] -- a different objective on the same constraints; out of scope of
the hot path this round."""


def opt_shortest_path(reftrack, normvectors, w_veh, print_debug=False):
    raise NotImplementedError("opt_shortest_path is outside the minimum-curvature hot path (SURVEY.md section 8f-4)")
