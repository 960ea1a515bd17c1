// rt_engine.hpp -- header-only C++ host shim over the C ABI (include/rt_abi.h) that keeps the reference's
// PbrtEngine surface (reference include/RT/pbrt_engine.h:51-203, src/pbrt/pbrt_engine.cu:65-236), so that a host
// such as the reference's main.cpp can switch engines by changing one include.  What the reference pulls from
// other singletons at render time (the viewer camera's position/front/zoom/focus/aperture, pbrt_engine.cu:104-106)
// is passed in explicitly: the GL-side classes are out of scope.  Meshes are registered as in the reference
// (addMesh, main.cpp:711-713) and flattened at render time, the one named "light" first (pbrt_engine.cu:167-178);
// any class with getName() and genTriangles(std::vector<T>&), T a 148-byte Triangle record, will do -- the
// reference's own Mesh, or rtscene::Mesh of rt_scene.hpp.
//
// Errors: like the reference's checkCudaErrors (include/RT/common.h:26-36) a failed call prints and exit(99)s.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

class PbrtEngine {
  public:
    static PbrtEngine &getInstance() { // pbrt_engine.h:53-56
        static PbrtEngine *engine = new PbrtEngine;
        return *engine;
    }

    explicit PbrtEngine(int _nx = 1280 / 2, int _ny = 720 / 2, int _ns = 100 / 10, int device = 0)
        : nx(_nx), ny(_ny), ns(_ns) { // NX/NY/NS, pbrt_engine.h:24-26,58
        check(rt_create(&ctx, device), "rt_create");
        uponPixelNumChange();
    }
    ~PbrtEngine() {
        if (ctx) rt_destroy(ctx);
    }
    PbrtEngine(const PbrtEngine &) = delete;
    PbrtEngine &operator=(const PbrtEngine &) = delete;

    void setNx(int _nx) { // pbrt_engine.h:65-77
        if (_nx != nx) nx_ny_changed = true;
        nx = _nx;
    }
    void setNy(int _ny) {
        if (_ny != ny) nx_ny_changed = true;
        ny = _ny;
    }
    void setNs(int _ns) { ns = _ns; }
    void uponPixelNumChange() { // pbrt_engine.h:81-84
        num_pixels = nx * ny;
        check(rt_resize(ctx, nx, ny, ns), "rt_resize");
        h_fb.assign((size_t)num_pixels * 3, 0.0f);
    }
    bool nXnYChanged() {
        bool t = nx_ny_changed;
        nx_ny_changed = false;
        return t;
    }

    void set(int _nx, int _ny, int _ns) { // pbrt_engine.h:86-91
        nx = _nx;
        ny = _ny;
        ns = _ns;
        num_pixels = nx * ny;
    }

    // addMesh, pbrt_engine.h:99-100 / pbrt_engine.cu:159-165: the vector form replaces the scene, the pointer form appends
    template <class M> void addMesh(M *_m) {
        m_vecScene.push_back(SceneMesh{[_m]() { return std::string(_m->getName()); },
                                       [_m](std::vector<rt_triangle148> &out) { gen(_m, &M::genTriangles, out); }});
    }
    template <class M> void addMesh(const std::vector<M *> &_m) {
        m_vecScene.clear();
        for (M *p : _m) addMesh(p);
    }
    // genTextures, pbrt_engine.h:133-152: the reference builds its texture objects from image files here
    // (cuManyTextures, src/cxx/texture.cpp:147-160); the host registers decoded RGBA8 images instead and this hands
    // them to the library
    void registerTexture(int index, const unsigned char *rgba8, int w, int h) {
        m_vecTextures.push_back(HostTexture{index, w, h, std::vector<unsigned char>(rgba8, rgba8 + (size_t)4 * w * h)});
    }
    void genTextures() {
        for (const HostTexture &t : m_vecTextures) setTexture(t.index, t.rgba.data(), t.w, t.h);
    }

    // m_vecTriangles directly, for hosts that flatten their scene themselves (light first, pbrt_engine.cu:167-178);
    // used only while no mesh is registered
    void setTriangles(const std::vector<rt_triangle148> &tris) { m_vecTriangles = tris; }
    void addTriangles(const rt_triangle148 *t, size_t n) { m_vecTriangles.insert(m_vecTriangles.end(), t, t + n); }

    // the five values camera_update reads from the viewer camera (pbrt_engine.cu:104-106)
    void setCamera(const float lookfrom[3], const float lookat[3], float zoom, float dist_to_focus, float aperture) {
        check(rt_set_camera(ctx, lookfrom, lookat, zoom, dist_to_focus, aperture), "rt_set_camera");
    }
    // one texture object (Texture::getTexture, src/cxx/texture.cpp:45-86): RGBA8, row 0 first
    void setTexture(int index, const unsigned char *rgba8, int w, int h) {
        check(rt_set_texture(ctx, index, rgba8, w, h), "rt_set_texture");
    }
    void setMaxDepth(int d) { check(rt_set_option(ctx, RT_OPT_MAX_DEPTH, d), "rt_set_option"); }

    // renderScene + __renderScene + join (pbrt_engine.cu:89-157); synchronous, as main.cpp:217-219 makes it
    void renderScene(bool _buildScene = true, const char *out_path = "./output.ppm") {
        std::cout << "rendering a " << nx << "[width]x" << ny << "[height]"
                  << " image with " << ns << " samples per pixel in ";
        std::cout << tx << "x" << ty << " blocks.\n";
        m_strRenderStatus = "PbrtEgnine status: Rendering.";
        if (_buildScene) {
            std::cout << "INFO: generate scene.\n";
            if (!m_vecScene.empty()) { // __renderScene, pbrt_engine.cu:96-98: clear, gather ("light" first), build
                m_vecTriangles.clear();
                for (const SceneMesh &p : m_vecScene)
                    if (p.name() == "light") p.genTriangles(m_vecTriangles);
                for (const SceneMesh &p : m_vecScene)
                    if (p.name() != "light") p.genTriangles(m_vecTriangles);
            }
            std::cout << "Total num of Triangles: " << m_vecTriangles.size() << std::endl;
            if (!m_vecTriangles.empty()) {
                check(rt_set_scene(ctx, m_vecTriangles.data(), (int)m_vecTriangles.size()), "rt_set_scene");
                check(rt_build_bvh(ctx, rand_calls_before), "rt_build_bvh");
            }
        } else {
            std::cout << "INFO: using scene generated.\n";
        }
        check(rt_render(ctx, h_fb.data(), &last_stats), "rt_render");
        std::cout << "took " << last_stats.kernel_ms / 1000.0 << " seconds.\n";
        check(rt_write_ppm(out_path, h_fb.data(), nx, ny), "rt_write_ppm");
        m_strRenderStatus = "PbrtEgnine status: finished";
        std::cout << "PbrtEngine: rendering finished.\n";
    }
    void join() {}                                                                 // renderScene is synchronous here
    void checkStatus() { std::cout << "PbrtEngine: rendering finished.\n"; }       // pbrt_engine.h:124-131

    struct SceneMesh { // one registered mesh, whatever its class
        std::function<std::string()> name;
        std::function<void(std::vector<rt_triangle148> &)> genTriangles;
    };
    std::vector<SceneMesh> m_vecScene;
    std::vector<rt_triangle148> m_vecTriangles;
    std::vector<float> h_fb; // nx*ny*3, row j = 0 at the bottom
    int nx, ny, ns, num_pixels = 0, tx = 16, ty = 16; // tx/ty are printed only: the launch geometry is the library's
    long long rand_calls_before = 0; // rand() calls the host made before the build (main.cpp:449-474 shares rand())
    rt_stats last_stats{};
    rt_ctx *ctx = nullptr;

  private:
    struct HostTexture {
        int index, w, h;
        std::vector<unsigned char> rgba;
    };
    std::vector<HostTexture> m_vecTextures;
    // calls m->genTriangles with the vector type the mesh class wants; a foreign 148-byte record is taken over byte for byte
    template <class M, class B, class T> static void gen(M *m, void (B::*f)(std::vector<T> &), std::vector<rt_triangle148> &out) {
        gen_into<T>(out, [&](std::vector<T> &v) { (m->*f)(v); });
    }
    template <class M, class B, class T> static void gen(M *m, void (B::*f)(std::vector<T> &) const, std::vector<rt_triangle148> &out) {
        gen_into<T>(out, [&](std::vector<T> &v) { (m->*f)(v); });
    }
    // the reference's own signature: virtual void genTriangles(std::vector<Triangle>&, bool _setNormal = false) (include/RT/mesh.h:241)
    template <class M, class B, class T> static void gen(M *m, void (B::*f)(std::vector<T> &, bool), std::vector<rt_triangle148> &out) {
        gen_into<T>(out, [&](std::vector<T> &v) { (m->*f)(v, false); });
    }
    template <class M, class B, class T> static void gen(M *m, void (B::*f)(std::vector<T> &, bool) const, std::vector<rt_triangle148> &out) {
        gen_into<T>(out, [&](std::vector<T> &v) { (m->*f)(v, false); });
    }
    template <class T, class F> static void gen_into(std::vector<rt_triangle148> &out, F call) {
        static_assert(sizeof(T) == sizeof(rt_triangle148), "genTriangles must emit the reference's 148-byte Triangle");
        std::vector<T> tmp;
        call(tmp);
        const size_t at = out.size();
        out.resize(at + tmp.size());
        if (!tmp.empty()) std::memcpy(static_cast<void *>(out.data() + at), static_cast<const void *>(tmp.data()), tmp.size() * sizeof(T));
    }
    void check(int rc, const char *what) {
        if (rc) {
            std::cerr << "RT error = " << rc << " in '" << what << "': " << (ctx ? rt_last_error(ctx) : "") << "\n";
            std::exit(99);
        }
    }
    bool nx_ny_changed = false;
    std::string m_strRenderStatus = "None";
};
