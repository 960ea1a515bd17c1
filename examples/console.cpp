// examples/console.cpp -- the reference's console loop (main.cpp:68-332) over the PbrtEngine shim and the C++ scene
// restatement, without the window: reads one command per line from stdin, prints what the reference prints.
//   console                      interactive / scripted session on the default scene (needs an MI355X)
//   console --dump-scene FILE    same loop, but `render` writes the flattened scene (148-byte records) to FILE, prints
//                                "<n> <rand calls>" and exits: no GPU needed (tests compare it with scene_graph.py)
//   --out FILE                   where `render` leaves its PPM (default ./output.ppm)
//   --scene default|cat          cat = the reference's second scene, main.cpp:596-662 (rt_scene.hpp cat_scene): textured
//                                light walls, a textured metal wall, the signature boxes; stand-in textures 0..3
//   --camera px,py,pz,fx,fy,fz   position and front of the GL camera at start (stands in for the binary camera snapshot
//                                the reference loads, Camera::loadSnapShot, main.cpp:663: absent)
//   --obj NAME=FILE[:MAT]        (any number) meshes through the reference's own OBJ reader
// Commands kept: q | m <mesh> | camera posx|negx|posy|negy|posz|negz|info|aperture x|dist_to_focus [x] |
//                render | render -cache | render status | render param nx ny ns | render hd |
//                mx|my|mz|rx|ry|rz|s <x> | reset | radius.    GL-only commands are accepted and ignored.
//   hipcc -std=c++17 -ffp-contract=off examples/console.cpp -Lrt_amd -lrt_amd -Wl,-rpath,'$ORIGIN/../rt_amd' -o examples/console
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../rt_amd/csrc/rt_engine.hpp"
#include "../rt_amd/csrc/rt_scene.hpp"

using namespace rtscene;

// make_bvh_nodes makes one rand() call per invocation (pbrt_engine.cu:335-368)
static long long build_rand_calls(long long cnt) {
    if (cnt <= 0) return 0;
    if (cnt <= 2) return 1;
    return 1 + build_rand_calls(cnt / 2) + build_rand_calls(cnt - cnt / 2);
}

struct GlCamera { // the fields __renderScene reads (include/RT/camera.h:32-47,110-159,245-272)
    Vec3 pos{0, 0, 3}, front{0, 0, -1};
    float zoom = 45.0f, dist_to_focus = 10.0f, aperture = 0.0f, speed = 2.5f;
};

int main(int argc, char **argv) {
    CountedRand rng;
    std::string scene_name = "default", dump_path, out_file = "./output.ppm";
    std::vector<std::string> objs;
    GlCamera cam;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (i + 1 >= argc) {
            std::cerr << "option " << a << " needs a value\n";
            return 2;
        }
        const std::string v = argv[++i];
        if (a == "--dump-scene") dump_path = v;
        else if (a == "--out") out_file = v;
        else if (a == "--scene") scene_name = v;
        else if (a == "--obj") objs.push_back(v);
        else if (a == "--camera") {
            float f[6];
            if (std::sscanf(v.c_str(), "%f,%f,%f,%f,%f,%f", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5]) != 6) {
                std::cerr << "--camera px,py,pz,fx,fy,fz\n";
                return 2;
            }
            cam.pos = Vec3{f[0], f[1], f[2]};
            cam.front = Vec3{f[3], f[4], f[5]};
        } else {
            std::cerr << "unknown option " << a << "\n";
            return 2;
        }
    }
    if (scene_name != "default" && scene_name != "cat") {
        std::cerr << "--scene default|cat\n";
        return 2;
    }
    const bool cat = scene_name == "cat";
    Collection meshes = cat ? cat_scene() : default_scene(rng);
    for (const std::string &spec : objs) { // NAME=FILE[:MAT]
        const size_t eq = spec.find('='), colon = spec.rfind(':');
        if (eq == std::string::npos) {
            std::cerr << "--obj NAME=FILE[:MAT]\n";
            return 2;
        }
        const bool has_mat = colon != std::string::npos && colon > eq;
        std::ifstream in(spec.substr(eq + 1, has_mat ? colon - eq - 1 : std::string::npos));
        std::vector<Vec3> v;
        std::vector<unsigned> idx;
        std::string err = "file not found or not good";
        if (!in.good() || !load_obj(in, v, idx, err)) {
            std::cerr << "ERROR: loading obj:(" << spec << ") " << err << "\n";
            return 2;
        }
        registerModel(meshes, std::make_shared<Mesh>(spec.substr(0, eq), v, idx, has_mat ? std::atoi(spec.c_str() + colon + 1) : 0));
    }
    const bool dump = !dump_path.empty();
    const char *out_path = out_file.c_str();
    long long rand_calls = rng.calls; // rand() position handed to the next build
    bool scene_textures_set = !cat; // the cat scene names textures 0..3: registered before its first render
    Mesh *curr = nullptr;
    bool loop = true, texture0_set = false;
    std::string line;
    while (loop) {
        std::cout << ">>>:";
        if (!std::getline(std::cin, line)) break;
        line = line.substr(0, 29); // cin.getline(input, 30), main.cpp:80
        std::istringstream iss(line);
        std::vector<std::string> tokens;
        for (std::string t; iss >> t;) tokens.push_back(t);
        if (tokens.empty()) continue;
        try {
            const std::string &first = tokens[0];
            if (first == "q") {
                std::cout << "bye, press any buttion to end.\n";
                loop = false;
            } else if (first == "m") {
                auto it = meshes.find(tokens.at(1));
                if (it == meshes.end()) {
                    std::cout << "ERROR: Model " << tokens[1] << " not found.\n";
                    continue;
                }
                curr = it->second.get();
                std::cout << "setting model finished.\n";
            } else if (first == "camera") {
                const std::string &second = tokens.at(1);
                auto look = [&](Vec3 p, Vec3 f) {
                    cam.pos = p;
                    cam.front = f;
                };
                if (second == "posx") look({3, 0, 0}, {-1, 0, 0});
                else if (second == "negx") look({-3, 0, 0}, {1, 0, 0});
                else if (second == "posy") look({0, 3, 0}, {0.1f, -1, 0.1f});
                else if (second == "negy") look({0, -3, 0}, {-0.1f, 1, -0.1f});
                else if (second == "posz") look({0, 0, 3}, {0, 0, -1});
                else if (second == "negz") look({0, 0, -3}, {0, 0, 1});
                else if (second == "info") {
                    std::cout << cam.pos.x << " " << cam.pos.y << " " << cam.pos.z << std::endl;
                    std::cout << cam.front.x << " " << cam.front.y << " " << cam.front.z << std::endl;
                } else if (second == "aperture") cam.aperture = std::stof(tokens.at(2));
                else if (second == "dist_to_focus") {
                    if (tokens.size() > 2) {
                        cam.dist_to_focus = std::stof(tokens[2]);
                    } else {
                        if (!curr) throw std::runtime_error("no model selected");
                        const Vec3 d = cam.pos - curr->getPosition(); // getDistToFocus(Mesh*), camera.h:261-265
                        cam.dist_to_focus = std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z);
                        std::cout << "INFO: set camera distance_to_focus to " << cam.dist_to_focus << std::endl;
                    }
                } else if (second == "speed") {
                    if (tokens.size() < 3) std::cout << "curr camera speed: " << cam.speed << std::endl;
                    else cam.speed = std::stof(tokens[2]);
                }
            } else if (first == "render" && dump) {
                std::vector<rt_triangle148> tris = gather(meshes);
                std::ofstream(dump_path, std::ios::binary).write((const char *)tris.data(), (std::streamsize)(tris.size() * sizeof(rt_triangle148)));
                std::cout << tris.size() << " " << rand_calls << "\n";
                return 0;
            } else if (first == "render") {
                PbrtEngine &engine = PbrtEngine::getInstance(); // created on first use: the only thing that needs a GPU
                if (tokens.size() >= 2 && tokens[1] == "status") {
                    engine.checkStatus();
                } else if (tokens.size() >= 2 && (tokens[1] == "param" || tokens[1] == "hd")) {
                    const bool hd = tokens[1] == "hd";
                    engine.setNx(hd ? 1280 : std::stoi(tokens.at(2)));
                    engine.setNy(hd ? 720 : std::stoi(tokens.at(3)));
                    engine.setNs(hd ? 100 : std::stoi(tokens.at(4)));
                    // the reference resizes only when nx or ny changed (main.cpp:199-201); here ns reaches the library
                    // through rt_resize as well, so every `render param` resizes
                    engine.uponPixelNumChange();
                } else {
                    const bool build = !(tokens.size() >= 2 && tokens[1] == "-cache");
                    if (!scene_textures_set) { // stand-ins for the images the scene names (genTextures, main.cpp:715)
                        for (int t = 0; t < 4; t++) {
                            const std::vector<unsigned char> px = stand_in_texture(t);
                            engine.registerTexture(t, px.data(), 64, 64);
                        }
                        engine.genTextures();
                        scene_textures_set = texture0_set = true;
                    }
                    if (build) { // main.cpp:711-713 registers every mesh of the name-sorted collection with the engine
                        std::vector<Mesh *> scene;
                        for (const auto &kv : meshes) scene.push_back(kv.second.get());
                        engine.addMesh(scene); // flattened by renderScene, "light" first (pbrt_engine.cu:167-178)
                    }
                    const float from[3] = {cam.pos.x, cam.pos.y, cam.pos.z};
                    const float at[3] = {cam.pos.x + cam.front.x, cam.pos.y + cam.front.y, cam.pos.z + cam.front.z};
                    engine.setCamera(from, at, cam.zoom, cam.dist_to_focus, cam.aperture);
                    engine.rand_calls_before = rand_calls;
                    engine.renderScene(build, out_path);
                    engine.join();
                    if (build) rand_calls += build_rand_calls((long long)engine.m_vecTriangles.size());
                }
            } else if (curr) {
                if (first == "mx") curr->translateXCommand(std::stof(tokens.at(1)));
                else if (first == "my") curr->translateYCommand(std::stof(tokens.at(1)));
                else if (first == "mz") curr->translateZCommand(std::stof(tokens.at(1)));
                else if (first == "rx") curr->rotationCommand(0, std::stof(tokens.at(1)));
                else if (first == "ry") curr->rotationCommand(1, std::stof(tokens.at(1)));
                else if (first == "rz") curr->rotationCommand(2, std::stof(tokens.at(1)));
                else if (first == "s") curr->scaleCommand(std::stof(tokens.at(1)));
                else if (first == "reset") curr->resetAll();
                else if (first == "radius") std::cout << "radius => " << curr->getAverageRadius() << std::endl;
                else if (first == "new") { // addXyPlane / addYzPlane / addZxPlane, main.cpp:309-319
                    const std::string &k = tokens.at(1);
                    const int kind = (k == "xy" || k == "yx") ? 0 : (k == "yz" || k == "zy") ? 1 : (k == "zx" || k == "xz") ? 2 : -1;
                    if (kind >= 0) {
                        registerModel(meshes, plane_mesh(kind));
                        if (!dump && !texture0_set) { // the planes name texture 0 ("test01.png" in the reference)
                            const std::vector<unsigned char> t = default_texture();
                            PbrtEngine::getInstance().registerTexture(0, t.data(), 64, 64);
                            PbrtEngine::getInstance().genTextures(); // pbrt_engine.h:133
                            texture0_set = true;
                        }
                    }
                }
            }
        } catch (std::exception &e) {
            std::cout << "Console Error: " << e.what() << std::endl;
        }
    }
    std::cout << "Quitting this thread.\n";
    return 0;
}
