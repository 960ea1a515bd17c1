// examples/render_ppm.cpp -- a minimal C++ host over the PbrtEngine shim, written the way the reference's main() drives
// its engine (main.cpp:711-713 addMesh; :195-201 render param; :212-219 render): the light quad of the reference's
// default scene (include/RT/MeshFactory.h:423-429) above two ground triangles, registered as meshes -- the ground before
// the light, to show the gather putting "light" first -- and rendered to ./output.ppm.
//   hipcc -std=c++17 -ffp-contract=off examples/render_ppm.cpp -Lrt_amd -lrt_amd -Wl,-rpath,'$ORIGIN/../rt_amd' -o examples/render_ppm
#include "../rt_amd/csrc/rt_engine.hpp"
#include "../rt_amd/csrc/rt_scene.hpp"

int main(int argc, char **argv) {
    using namespace rtscene;
    const std::vector<Vec3> g = {{-20, 0, -20}, {-20, 0, 20}, {20, 0, 20}, {20, 0, -20}};
    Mesh ground_a("ground_a", g, {0, 1, 2}, 8), ground_b("ground_b", g, {0, 2, 3}, 9);
    MESH_PTR light = light_mesh(); // corners (+-4, 10, +-4), winding {3,2,1, 3,1,0}, material 3
    PbrtEngine &engine = PbrtEngine::getInstance();
    if (argc > 3) {
        engine.setNx(std::atoi(argv[1]));
        engine.setNy(std::atoi(argv[2]));
        engine.setNs(std::atoi(argv[3]));
        engine.uponPixelNumChange(); // "render param nx ny ns", main.cpp:195-201
    }
    engine.addMesh(&ground_a);
    engine.addMesh(&ground_b);
    engine.addMesh(light.get());
    engine.genTextures(); // nothing registered: no texture in this scene
    const float from[3] = {5.5f, 2.5f, 8.0f}, at[3] = {0.0f, 1.3f, 0.0f};
    engine.setCamera(from, at, 40.0f, 10.0f, 0.0f);
    engine.renderScene(true);
    engine.join();
    return 0;
}
