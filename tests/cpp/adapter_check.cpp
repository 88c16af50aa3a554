// Exercises include/ccmp_ompl_adapter.hpp part 1 (ccmp::Projector) from plain C++ — what the
// reference-side adapter classes call.  Prints results as hex doubles so the Python test can compare
// them bit for bit with the oracle.  usage: adapter_check <config.yaml> <states.txt>
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "ccmp_ompl_adapter.hpp"

static void print_hex(const double *v, int n)
{
  for (int i = 0; i < n; i++) {
    uint64_t u;
    std::memcpy(&u, &v[i], 8);
    std::printf("%016" PRIx64 "%c", u, i + 1 == n ? '\n' : ' ');
  }
}

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  try {
    ccmp::Projector P(argv[1], 0);
    bool threw = false;
    try { P.setTolerance(0.0, 1.0); } catch (const ccmp::Error &e) { threw = e.code == CCMP_EINVAL; }
    std::printf("setTolerance_throws %d\n", threw ? 1 : 0);
    std::FILE *fp = std::fopen(argv[2], "r");
    if (!fp) return 3;
    std::vector<double> q;
    double v;
    while (std::fscanf(fp, "%lf", &v) == 1) q.push_back(v);
    std::fclose(fp);
    const size_t B = q.size() / 14;
    for (size_t i = 0; i < B; i++) {
      double x[14], f[2];
      std::memcpy(x, &q[14 * i], sizeof x);
      const bool sat0 = P.isSatisfied(x);
      const bool ok = P.project(x);  // in place
      P.function(x, f);
      std::printf("state %zu ok %d sat_before %d sat_after %d jv %d\n", i, ok ? 1 : 0, sat0 ? 1 : 0, P.isSatisfied(x) ? 1 : 0,
                  P.jointValid(x) ? 1 : 0);
      print_hex(x, 14);
      print_hex(f, 2);
    }
    std::vector<double> out(q.size());
    std::vector<uint8_t> ok(B);
    std::vector<uint16_t> it(B);
    P.projectBatch(q.data(), out.data(), ok.data(), it.data(), B);
    for (size_t i = 0; i < B; i++) std::printf("batch %zu ok %d iters %u\n", i, ok[i], it[i]);
    // sampler buffer: the stream must not depend on the refill size
    ccmp::SampleBuffer s1(P, 42, 4), s2(P, 42, 3);
    for (int i = 0; i < 7; i++) {
      double a[14], b[14];
      s1.next(a);
      s2.next(b);
      std::printf("sample %d same %d\n", i, std::memcmp(a, b, sizeof a) == 0 ? 1 : 0);
      print_hex(a, 14);
    }
    // extend step from the first state towards the second, host validity rejecting nothing / everything
    if (B >= 2) {
      std::vector<std::vector<double>> geo;
      bool g1 = ccmp::discreteGeodesic(P, &out[0], &out[14], true, [](const double *) { return true; }, &geo, 64);
      std::printf("geodesic ok %d n %zu\n", g1 ? 1 : 0, geo.size());
      for (auto &st : geo) print_hex(st.data(), 14);
      bool g2 = ccmp::discreteGeodesic(P, &out[0], &out[14], false, [](const double *) { return false; }, &geo, 64);
      std::printf("geodesic_rejected ok %d n %zu\n", g2 ? 1 : 0, geo.size());
      // a buffer that is too small must not cut the list: the adapter re-runs the edge with room for all of it
      bool g3 = ccmp::discreteGeodesic(P, &out[0], &out[14], true, [](const double *) { return true; }, &geo, 1);
      std::printf("geodesic_small_buffer ok %d n %zu\n", g3 ? 1 : 0, geo.size());
      // growTree's neighbour loop in one launch: edges 0->1, 1->0 and 0->0 (closer than delta: only `from`), nothing rejected
      {
        std::vector<double> fr, tt;
        const int pairs[3][2] = {{0, 1}, {1, 0}, {0, 0}};
        for (auto &pr : pairs) {
          fr.insert(fr.end(), &out[14 * pr[0]], &out[14 * pr[0]] + 14);
          tt.insert(tt.end(), &out[14 * pr[1]], &out[14 * pr[1]] + 14);
        }
        std::vector<std::vector<std::vector<double>>> lists;
        std::vector<char> reached;
        int calls = 0;
        ccmp::discreteGeodesicBatch(P, fr.data(), tt.data(), 3, false, [&](const double *) { calls++; return true; }, &lists, &reached, 64);
        for (int e = 0; e < 3; e++) {
          std::printf("gbatch %d ok %d n %zu\n", e, (int)reached[e], lists[e].size());
          print_hex(lists[e].back().data(), 14);
        }
        std::printf("gbatch checker_calls %d\n", calls);
      }
      // a LARGE batch takes another shape inside the adapter (lists of 16 + a budget of Newton rounds, then continuation of
      // the edges that stop short): every edge must come out as the single-edge call gives it
      {
        const size_t E = 1200;
        std::vector<double> fr, tt;
        for (size_t e = 0; e < E; e++) {
          const size_t a = e % B, b = (e + 1 + e / B) % B;
          fr.insert(fr.end(), &out[14 * a], &out[14 * a] + 14);
          tt.insert(tt.end(), &out[14 * b], &out[14 * b] + 14);
        }
        std::vector<std::vector<std::vector<double>>> lists;
        std::vector<char> reached;
        ccmp::discreteGeodesicBatch(P, fr.data(), tt.data(), E, true, [](const double *) { return true; }, &lists, &reached, 64);
        int bad = 0;
        for (size_t e = 0; e < E; e += 7) { // a seventh of them, one by one
          std::vector<std::vector<double>> one;
          const bool g = ccmp::discreteGeodesic(P, &fr[14 * e], &tt[14 * e], true, [](const double *) { return true; }, &one, 64);
          bool same = (g ? 1 : 0) == (int)reached[e] && one.size() == lists[e].size();
          for (size_t k = 0; same && k < one.size(); k++) same = std::memcmp(one[k].data(), lists[e][k].data(), 14 * sizeof(double)) == 0;
          if (!same) bad++;
        }
        std::printf("gbatch_big edges %zu mismatches %d\n", E, bad);
      }
      // sampleUniformNear / sampleGaussian around the first projected state through the look-ahead buffers
      ccmp::RefSampleBuffer nb(P, 42, ccmp::RefSampleBuffer::Near, 4), gb(P, 42, ccmp::RefSampleBuffer::Gaussian, 4);
      for (int i = 0; i < 6; i++) {  // 6 > look-ahead: the second refill continues the stream
        double a[14];
        nb.next(a, &out[0], 0.2);
        std::printf("near %d\n", i);
        print_hex(a, 14);
      }
      double g[14];
      gb.next(g, &out[0], 0.05);
      std::printf("gauss 0\n");
      print_hex(g, 14);
    }
    // several threads on ONE Projector (the reference touches its constraint from the goal-sampling thread and from
    // checkForSolution beside constructRoadmap): the context's pinned I/O block is shared, the Projector's mutex
    // serialises the calls — every thread must get the serial results
    {
      std::vector<double> serial(q.size());
      std::vector<int> serial_ok(B);
      for (size_t i = 0; i < B; i++) {
        std::memcpy(&serial[14 * i], &q[14 * i], 14 * sizeof(double));
        serial_ok[i] = P.project(&serial[14 * i]) ? 1 : 0;
      }
      int mismatches = 0;
      std::vector<std::thread> th;
      std::vector<int> bad(4, 0);
      for (int t = 0; t < 4; t++)
        th.emplace_back([&, t] {
          for (int rep = 0; rep < 8; rep++)
            for (size_t i = 0; i < B; i++) {
              double x[14];
              std::memcpy(x, &q[14 * i], sizeof x);
              const bool ok = P.project(x);
              if ((ok ? 1 : 0) != serial_ok[i] || std::memcmp(x, &serial[14 * i], sizeof x) != 0) bad[t]++;
              double f[2];
              P.function(x, f);
              if (P.isSatisfied(x) != (f[0] <= 1e-3 && f[1] <= 5e-3)) bad[t]++;
            }
        });
      for (auto &h : th) h.join();
      for (int t = 0; t < 4; t++) mismatches += bad[t];
      std::printf("threads mismatches %d\n", mismatches);
    }
    // proxy scene in front of the host's validity test: one sphere on either hand, one on arm 1's forearm, the
    // reference's sub_table; hands allowed against each other
    {
      std::vector<ccmp_sphere> sph(3);
      std::memset(sph.data(), 0, sph.size() * sizeof(ccmp_sphere));
      sph[0].frame = CCMP_FRAME(0, 7); sph[0].group = 0; sph[0].r = 0.04;
      sph[1].frame = CCMP_FRAME(1, 7); sph[1].group = 1; sph[1].r = 0.04;
      sph[2].frame = CCMP_FRAME(1, 3); sph[2].group = 2; sph[2].c[2] = 0.1; sph[2].r = 0.06;
      std::vector<ccmp_box> boxes{ccmp::ProxyScene::subTable(3)};
      uint32_t allowed[32] = {0};
      ccmp::ProxyScene::allow(allowed, 0, 1);
      ccmp::ProxyScene scene(P, sph, boxes, allowed);
      std::printf("scene pairs %d\n", scene.numPairs());
      for (size_t i = 0; i < B && i < 4; i++) {
        int32_t pair = 0;
        const double clr = scene.clearance(&q[14 * i], &pair);
        std::printf("clearance %zu pair %d\n", i, (int)pair);
        print_hex(&clr, 1);
      }
    }
    // one process, "several" GPUs with the collective: this box has one, so the communicator has one rank
    {
      ccmp::ShardedProjector S(argv[1], std::vector<int>{0});
      std::vector<uint64_t> counts;
      std::vector<double> valid = S.sampleProjectSharded(42, 0, 500, &counts);
      std::printf("sharded n_valid %zu counts %zu first %llu\n", valid.size() / 14, counts.size(), (unsigned long long)counts[0]);
      for (size_t i = 0; i < valid.size() / 14 && i < 3; i++) print_hex(&valid[14 * i], 14);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
