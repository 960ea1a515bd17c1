// examples/render_ppm.cpp -- a minimal C++ host over the PbrtEngine shim: the light quad of the reference's default
// scene (include/RT/MeshFactory.h:423-429) above one ground triangle pair, rendered to ./output.ppm the way the
// reference's `render` console command does (main.cpp:212-219).
//   hipcc -std=c++17 -ffp-contract=off examples/render_ppm.cpp -Lrt_amd -lrt_amd -Wl,-rpath,'$ORIGIN/../rt_amd' -o examples/render_ppm
#include <cmath>
#include <cstring>

#include "../rt_amd/csrc/rt_engine.hpp"

static rt_triangle148 make_tri(const float a[3], const float b[3], const float c[3], int mat) {
    // Triangle(v0,v1,v2,mat,texIdx=-1): setTriangle, include/RT/triangle.h:102-111
    rt_triangle148 t;
    std::memset(&t, 0, sizeof(t));
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) {
        t.v0[k] = a[k];
        t.v1[k] = b[k];
        t.v2[k] = c[k];
        e1[k] = b[k] - a[k];
        e2[k] = c[k] - a[k];
        t.edge1[k] = e1[k];
        t.edge2[k] = e2[k];
    }
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], -(e1[0] * e2[2] - e1[2] * e2[0]), e1[0] * e2[1] - e1[1] * e2[0]};
    float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int k = 0; k < 3; k++) t.normal[k] = n[k] / len;
    t.calNormal = -1;
    t.matType = mat;
    t.shapeType = 0;
    t.texIdx = -1;
    return t;
}

int main(int argc, char **argv) {
    const float v[4][3] = {{4, 10, -4}, {-4, 10, -4}, {-4, 10, 4}, {4, 10, 4}};
    const float g[4][3] = {{-20, 0, -20}, {-20, 0, 20}, {20, 0, 20}, {20, 0, -20}};
    std::vector<rt_triangle148> tris;
    tris.push_back(make_tri(v[3], v[2], v[1], 3)); // flipped winding {3,2,1, 3,1,0}, material 3 = light
    tris.push_back(make_tri(v[3], v[1], v[0], 3));
    tris.push_back(make_tri(g[0], g[1], g[2], 8));
    tris.push_back(make_tri(g[0], g[2], g[3], 9));
    PbrtEngine &engine = PbrtEngine::getInstance();
    if (argc > 3) {
        engine.setNx(std::atoi(argv[1]));
        engine.setNy(std::atoi(argv[2]));
        engine.setNs(std::atoi(argv[3]));
        engine.uponPixelNumChange(); // "render param nx ny ns", main.cpp:195-201
    }
    engine.setTriangles(tris);
    const float from[3] = {5.5f, 2.5f, 8.0f}, at[3] = {0.0f, 1.3f, 0.0f};
    engine.setCamera(from, at, 40.0f, 10.0f, 0.0f);
    engine.renderScene(true);
    engine.join();
    return 0;
}
