// Exercises the dump-format writers of include/ccmp_ompl_adapter.hpp part 1 (no OMPL, no GPU, nothing to link):
//   format_check matrix   <path.txt>    parse a printAsMatrix dump, write it again
//   format_check graphml  <graph.txt>   graph.txt: "N E", N lines of 14 reals, E lines "a b weight"
//   format_check graphviz <graph.txt>
// tests/test_path_format.py compares the output with the reference's recorded files byte for byte.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "ccmp_ompl_adapter.hpp"

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  std::ifstream in(argv[2]);
  if (!in) return 3;
  if (!std::strcmp(argv[1], "matrix")) {
    std::vector<double> q;
    double v;
    while (in >> v) q.push_back(v);
    if (q.size() % 14) return 4;
    ccmp::printAsMatrix(std::cout, q.data(), q.size() / 14);
    return 0;
  }
  size_t n = 0, e = 0;
  in >> n >> e;
  std::vector<double> nodes(n * 14), w(e);
  std::vector<std::pair<unsigned, unsigned>> edges(e);
  for (size_t i = 0; i < n * 14; ++i) in >> nodes[i];
  for (size_t k = 0; k < e; ++k) in >> edges[k].first >> edges[k].second >> w[k];
  if (!in) return 5;
  if (!std::strcmp(argv[1], "graphml")) ccmp::printGraphML(std::cout, nodes.data(), n, edges.data(), e, w.data());
  else if (!std::strcmp(argv[1], "graphviz")) ccmp::printGraphviz(std::cout, n, edges.data(), e);
  else return 2;
  return 0;
}
