// CPU unit test of the product's HOST-side geometry (line_host.h, inst_host.h): plain g++, no GPU, no HIP runtime.  tests/test_host_logic.py feeds the same
// inputs to the oracle's restatements and compares the printed numbers.
//   host_logic_test line < in.txt   per row: plk(6) obs(4)        -> orth(4) plk_back(6) valid e1(3) e2(3)
//   host_logic_test tri  < in.txt   nobs start Rs(99) Ps(33) ric(9) tic(3) obs(4 nobs) -> tri plk(6) ptw1(3) ptw2(3)
//   host_logic_test box  < in.txt   n dims(3) seed pts(3n)        -> centre(3) of fit_box_ransac, ok + centre(3) of fit_box_camera
#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>
#include "line_host.h"
#include "inst_host.h"

using namespace be;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    if (mode == "line") {
        double v[10];
        while (std::cin >> v[0]) {
            for (int k = 1; k < 10; ++k) std::cin >> v[k];
            dvl::Plk l{ mk3(v[0], v[1], v[2]), mk3(v[3], v[4], v[5]) };
            double o[4]; dvl::plk_to_orth(l, o);
            const dvl::Plk b = dvl::orth_to_plk(o);
            d3 e1 = mk3(0, 0, 0), e2 = mk3(0, 0, 0);
            const bool ok = dvl::line_trimming(l, v + 6, e1, e2);
            std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g %.17g %.17g %.17g %.17g %.17g\n", o[0], o[1], o[2], o[3], b.n.x, b.n.y, b.n.z, b.v.x, b.v.y, b.v.z,
                        (int)ok, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z);
        }
        return 0;
    }
    if (mode == "tri") {
        int nobs, start;
        while (std::cin >> nobs >> start) {
            m33 Rs[11]; d3 Ps[11]; m33 ric; d3 tic;
            for (int f = 0; f < 11; ++f) for (int k = 0; k < 9; ++k) std::cin >> Rs[f].m[k];
            for (int f = 0; f < 11; ++f) std::cin >> Ps[f].x >> Ps[f].y >> Ps[f].z;
            for (int k = 0; k < 9; ++k) std::cin >> ric.m[k];
            std::cin >> tic.x >> tic.y >> tic.z;
            std::vector<dv_line_row> rows(1);
            dvl::LineMgr mgr; mgr.min_obs = 2;
            for (int k = 0; k < nobs; ++k) {
                dv_line_row r{}; r.id = 7; for (int q = 0; q < 4; ++q) std::cin >> r.left[q];
                mgr.add(start + k, &r, 1);
            }
            mgr.lms[0].start = start;
            mgr.triangulate(Rs, Ps, ric, tic);
            const dvl::LLm& L = mgr.lms[0];
            std::printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (int)L.tri, L.plk.n.x, L.plk.n.y, L.plk.n.z, L.plk.v.x, L.plk.v.y, L.plk.v.z,
                        L.ptw1.x, L.ptw1.y, L.ptw1.z, L.ptw2.x, L.ptw2.y, L.ptw2.z);
        }
        return 0;
    }
    if (mode == "box") {
        int n; double dims[3]; unsigned long long seed;
        while (std::cin >> n >> dims[0] >> dims[1] >> dims[2] >> seed) {
            std::vector<d3> pts(n);
            for (auto& p : pts) std::cin >> p.x >> p.y >> p.z;
            const d3 c = dvi::fit_box_ransac(pts, dims, seed);
            d3 cc = mk3(0, 0, 0);
            const bool ok = dvi::fit_box_camera(pts, dims, cc);
            std::printf("%.17g %.17g %.17g %d %.17g %.17g %.17g\n", c.x, c.y, c.z, (int)ok, cc.x, cc.y, cc.z);
        }
        return 0;
    }
    return 2;
}
