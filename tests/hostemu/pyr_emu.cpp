// Host walk of pyramid_kernel's phase code (r-vio_amd/csrc/pyr_sep.h): blocks x phases x threads in plain loops, a barrier = the end of a
// thread loop.  TEST INFRASTRUCTURE (tests/test_pyramid_phases.py); the product never builds or loads this file.
//   g++ -O1 -shared -fPIC tests/hostemu/pyr_emu.cpp -o tests/hostemu/libpyr_emu.so
#include <cstring>
#include <vector>
#include "../../r-vio_amd/csrc/pyr_sep.h"

extern "C" int pyr_emulate(const uint8_t* src, int w, int h, int stride, int levels, int copy0, uint8_t* o0, uint8_t* o1, uint8_t* o2, uint8_t* o3,
                           int poison, int reverse) {
    // reverse: walk the threads of every phase from 255 down — a phase that read what another thread of the SAME phase wrote would now differ
    PyrOut p;
    uint8_t* outs[4] = {o0, o1, o2, o3};
    int lw = w, lh = h;
    for (int l = 0; l < 4; ++l) { p.img[l] = outs[l]; p.w[l] = lw; p.h[l] = lh; lw = (lw + 1) / 2; lh = (lh + 1) / 2; }
    const int w3 = p.w[3], h3 = p.h[3];
    // the launch of rvio_hip.hip: one workgroup per 8x8 tile of level 3, whatever `levels` is
    for (int by = 0; by < (h3 + 7) / 8; ++by)
        for (int bx = 0; bx < (w3 + 7) / 8; ++bx) {
            PyrLds s;
            std::memset(&s, poison, sizeof s);      // LDS is never zero on entry: every byte read must have been written by this workgroup
            const PyrGeom g = pyr_geom(p, bx, by, levels, copy0);
            if (g.nl < 1) continue;
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase0(g, t, s, src, stride);
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase1(g, t, s, p);
            if (g.nl < 2) continue;
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase2(g, t, s, p);
            if (g.nl < 3) continue;
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase3(g, t, s);
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase4(g, t, s, p);
            if (g.nl < 4) continue;
            for (int i = 0, t; t = reverse ? PYR_T - 1 - i : i, i < PYR_T; ++i) pyr_phase5(g, t, s, p);
        }
    return (int)sizeof(PyrLds);
}
