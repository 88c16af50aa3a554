"""Path and planner-graph dump formats: PathGeometric::printAsMatrix as written by
ConstrainedPlanningCommon.cpp:219-222 and parsed by the reference's scripts, and PlannerData::printGraphML /
printGraphviz as written by ConstrainedProblem::dumpGraph (ConstrainedPlanningCommon.h:73-87).  The reference's
recorded outputs (debug/*, copied byte for byte by tests/golden/make_fixtures.py) are the known answers: parsing
them and writing them again must give the same bytes — from the Python mirror and from the C++ adapter."""
import os
import subprocess

import pytest

from conftest import GOLDEN, HERE, ROOT

DUMPS = os.path.join(GOLDEN, "dumps")


def test_recorded_paths_round_trip():
    from closed_chain_motion_planner_amd import format_path_matrix, parse_path_matrix

    for obj in ("Wine_Bottle", "dumbbell"):
        text = open(os.path.join(GOLDEN, "paths", obj + "_path.txt")).read()
        states = parse_path_matrix(text)
        assert states.shape[1] == 14
        assert text.endswith(" \n\n")  # trailing space per value, one empty line after the last state
        assert format_path_matrix(states) == text


@pytest.mark.parametrize("obj", ["Wine_Bottle", "dumbbell", "stefan"])
def test_recorded_graph_dumps_round_trip(obj):
    from closed_chain_motion_planner_amd import format_graphml, format_graphviz, parse_graphml

    gml = open(os.path.join(DUMPS, obj + "_node_info.graphml")).read()
    dot = open(os.path.join(DUMPS, obj + "_graph_info.dot")).read()
    nodes, edges, weights = parse_graphml(gml)
    assert nodes.shape == ({"Wine_Bottle": 10, "dumbbell": 4, "stefan": 0}[obj], 14)
    assert format_graphml(nodes, edges, weights) == gml
    assert format_graphviz(len(nodes), edges) == dot
    if obj != "stefan":  # the JSON roadmap fixture the geodesic tests use is the same graph
        from conftest import load_roadmap

        n2, e2 = load_roadmap(obj)
        assert (n2 == nodes).all() and e2 == edges


@pytest.fixture(scope="module")
def format_tool(tmp_path_factory):
    """tests/cpp/format_check.cpp: the writers of include/ccmp_ompl_adapter.hpp (part 1: no OMPL, no GPU needed)"""
    exe = str(tmp_path_factory.mktemp("fmt") / "format_check")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(HERE, "cpp", "format_check.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("obj", ["Wine_Bottle", "dumbbell", "stefan"])
def test_cpp_writers_reproduce_the_recorded_dumps(format_tool, obj, tmp_path):
    """ccmp::printAsMatrix / printGraphML / printGraphviz (C++ adapter) re-emit the reference's files byte for byte"""
    from closed_chain_motion_planner_amd import parse_graphml

    gml_path = os.path.join(DUMPS, obj + "_node_info.graphml")
    nodes, edges, weights = parse_graphml(open(gml_path).read())
    graph_in = tmp_path / "graph.txt"
    with open(graph_in, "w") as f:
        f.write("%d %d\n" % (len(nodes), len(edges)))
        for q in nodes:
            f.write(" ".join(repr(float(v)) for v in q) + "\n")
        for (a, b), w in zip(edges, weights):
            f.write("%d %d %r\n" % (a, b, w))
    out = subprocess.run([format_tool, "graphml", str(graph_in)], check=True, capture_output=True).stdout
    assert out == open(gml_path, "rb").read()
    out = subprocess.run([format_tool, "graphviz", str(graph_in)], check=True, capture_output=True).stdout
    assert out == open(os.path.join(DUMPS, obj + "_graph_info.dot"), "rb").read()
    if obj != "stefan":
        path = os.path.join(GOLDEN, "paths", obj + "_path.txt")
        out = subprocess.run([format_tool, "matrix", path], check=True, capture_output=True).stdout
        assert out == open(path, "rb").read()
