"""Path dump format: PathGeometric::printAsMatrix as written by
ConstrainedPlanningCommon.cpp:219-222 and parsed by the reference's scripts — the recorded outputs
must round-trip byte for byte."""
import os

from conftest import GOLDEN


def test_recorded_paths_round_trip():
    from closed_chain_motion_planner_amd import format_path_matrix, parse_path_matrix

    for obj in ("Wine_Bottle", "dumbbell"):
        text = open(os.path.join(GOLDEN, "paths", obj + "_path.txt")).read()
        states = parse_path_matrix(text)
        assert states.shape[1] == 14
        assert format_path_matrix(states) == text
