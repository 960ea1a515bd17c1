// rt_scene.hpp -- header-only C++ restatement of the host scene code that feeds the trace loop:
// Mesh / FlyWeight poses and Mesh::genTriangles (reference include/RT/mesh.h:243-364, 591-622; src/cxx/mesh.cpp:119-156),
// the gather order of PbrtEngine::__genTrianglesFromScene (src/pbrt/pbrt_engine.cu:167-178) over the name-sorted
// std::map of main.cpp:711-713, and the default scene of main() (main.cpp:408-500).  No GL, no assimp.
//
// glm is a system dependency of the reference, absent from its checkout (version unpinned): the few operations used
// are restated from its published float arithmetic (type_mat4x4.inl, matrix_transform.inl), column-major.
// Compile with -ffp-contract=off: the records must not depend on whether the compiler fuses a*b+c.
// rt_amd/scene_graph.py is the same restatement in numpy; tests compare the two byte for byte.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <istream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtscene {

struct Vec3 {
    float x = 0, y = 0, z = 0;
};
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }

struct Mat4 { // c[col][row]
    float c[4][4];
    static Mat4 identity() {
        Mat4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m.c[i][j] = i == j ? 1.0f : 0.0f;
        return m;
    }
};
// glm::translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
inline Mat4 translate(const Mat4 &m, Vec3 v) {
    Mat4 r = m;
    for (int k = 0; k < 4; k++) r.c[3][k] = ((m.c[0][k] * v.x + m.c[1][k] * v.y) + m.c[2][k] * v.z) + m.c[3][k];
    return r;
}
inline Mat4 scale(const Mat4 &m, float s) {
    Mat4 r = m;
    for (int col = 0; col < 3; col++)
        for (int k = 0; k < 4; k++) r.c[col][k] = m.c[col][k] * s;
    return r;
}
// mat4 * mat4: column j = a[0]*b[j][0] + a[1]*b[j][1] + a[2]*b[j][2] + a[3]*b[j][3], left to right
inline Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int j = 0; j < 4; j++)
        for (int k = 0; k < 4; k++)
            r.c[j][k] = ((a.c[0][k] * b.c[j][0] + a.c[1][k] * b.c[j][1]) + a.c[2][k] * b.c[j][2]) + a.c[3][k] * b.c[j][3];
    return r;
}
// glm::rotate(mat4(1), angle, axis), axis of unit length; sine / cosine are the library's (rt_test_sincosf)
inline Mat4 rotate_identity(float angle, Vec3 ax) {
    float s, c;
    rt_test_sincosf(angle, &s, &c);
    const float a[3] = {ax.x, ax.y, ax.z};
    float t[3];
    for (int k = 0; k < 3; k++) t[k] = (1.0f - c) * a[k];
    Mat4 r = Mat4::identity();
    r.c[0][0] = c + t[0] * a[0];
    r.c[0][1] = t[0] * a[1] + s * a[2];
    r.c[0][2] = t[0] * a[2] - s * a[1];
    r.c[1][0] = t[1] * a[0] - s * a[2];
    r.c[1][1] = c + t[1] * a[1];
    r.c[1][2] = t[1] * a[2] + s * a[0];
    r.c[2][0] = t[2] * a[0] + s * a[1];
    r.c[2][1] = t[2] * a[1] - s * a[0];
    r.c[2][2] = c + t[2] * a[2];
    return r;
}
// vec3(m * vec4(p, 1)): glm associates (m0*x + m1*y) + (m2*z + m3*w)
inline Vec3 apply(const Mat4 &m, Vec3 p) {
    float o[3];
    for (int k = 0; k < 3; k++) o[k] = (m.c[0][k] * p.x + m.c[1][k] * p.y) + (m.c[2][k] * p.z + m.c[3][k] * 1.0f);
    return Vec3{o[0], o[1], o[2]};
}

// Triangle(v0, v1, v2, matType, texIdx) -> setTriangle (include/RT/triangle.h:90-111)
struct Vec2 {
    float x = 0, y = 0;
};
inline rt_triangle148 make_triangle(Vec3 a, Vec3 b, Vec3 c, int mat, int tex, Vec2 ta = {}, Vec2 tb = {}, Vec2 tc = {}) {
    rt_triangle148 t;
    std::memset(&t, 0, sizeof(t));
    const float va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z}, vc[3] = {c.x, c.y, c.z};
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) {
        t.v0[k] = va[k];
        t.v1[k] = vb[k];
        t.v2[k] = vc[k];
        e1[k] = vb[k] - va[k];
        e2[k] = vc[k] - va[k];
        t.edge1[k] = e1[k];
        t.edge2[k] = e2[k];
    }
    const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], -(e1[0] * e2[2] - e1[2] * e2[0]), e1[0] * e2[1] - e1[1] * e2[0]};
    const float len = std::sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    for (int k = 0; k < 3; k++) t.normal[k] = n[k] / len;
    t.calNormal = -1;
    t.matType = mat;
    t.shapeType = 0;
    t.texIdx = tex;
    t.u[0] = ta.x; // Triangle(..., vec3(uv1[0], uv2[0], uv3[0]), vec3(uv1[1], uv2[1], uv3[1])), mesh.cpp:148-149
    t.u[1] = tb.x;
    t.u[2] = tc.x;
    t.v[0] = ta.y;
    t.v[1] = tb.y;
    t.v[2] = tc.y;
    return t;
}
// Triangle(centre, radius, matType) (triangle.h:95-100)
inline rt_triangle148 make_sphere(Vec3 c, float r, int mat) {
    rt_triangle148 t;
    std::memset(&t, 0, sizeof(t));
    t.v0[0] = c.x;
    t.v0[1] = c.y;
    t.v0[2] = c.z;
    t.v1[0] = r;
    t.calNormal = -1;
    t.matType = mat;
    t.shapeType = 1;
    t.texIdx = -1;
    return t;
}

class Mesh {
  public:
    Mesh(std::string name, std::vector<Vec3> v, std::vector<unsigned> idx, int mat = 0, std::vector<Vec2> tc = {})
        : m_strName(std::move(name)), vertices(std::move(v)), indices(std::move(idx)), texcoords(std::move(tc)), matType(mat) {
        texcoords.resize(vertices.size()); // Vertex::TexCoords; zero where the reference's loader would supply them
    }
    virtual ~Mesh() = default;

    // console commands, mesh.h:293-364
    void scaleCommand(float amp) { m_fScale *= amp; }
    void translateXCommand(float d) { m_fTranslationX += d; }
    void translateYCommand(float d) { m_fTranslationY += d; }
    void translateZCommand(float d) { m_fTranslationZ += d; }
    float getRadians(float d) const { return (float)(d / 180.f * 3.1415926); } // PI of mesh.h:38 is a double
    void rotationCommand(int axis, float d) {
        Vec3 a{axis == 0 ? 1.0f : 0.0f, axis == 1 ? 1.0f : 0.0f, axis == 2 ? 1.0f : 0.0f};
        m_mat4Rotation = mul(rotate_identity(getRadians(d), a), m_mat4Rotation);
    }
    void resetAll() {
        m_mat4Rotation = Mat4::identity();
        m_fTranslationX = m_fTranslationY = m_fTranslationZ = 0.0f;
        m_fScale = 1.0f;
    }
    void setPosition(Vec3 p) {
        m_fTranslationX = p.x;
        m_fTranslationY = p.y;
        m_fTranslationZ = p.z;
    }

    const std::string &getName() const { return m_strName; }
    const Mesh *who() const { return m_ptrTopMesh ? m_ptrTopMesh : this; } // setTopMesh, mesh.h:104-144
    int getMatType() const { return who()->matType; }
    bool isSphere() const { return who()->m_bIsSphere; }
    Mat4 getModel() const { // mesh.h:278-291
        const Mesh *w = who();
        Mat4 m = translate(Mat4::identity(), Vec3{w->m_fTranslationX, w->m_fTranslationY, w->m_fTranslationZ});
        m = mul(m, w->m_mat4Rotation);
        return scale(m, w->m_fScale);
    }
    Vec3 centroid() const { // resolvePosition, mesh.h:243-251 (0/0 for a mesh without vertices, as there)
        Vec3 s;
        for (const Vec3 &v : vertices) {
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
        }
        const float n = (float)vertices.size();
        return Vec3{s.x / n, s.y / n, s.z / n};
    }
    Vec3 getPosition() const { return apply(getModel(), centroid()); }
    float getAverageRadius() const { // mesh.h:253-265
        if (vertices.empty()) return std::nanf("");
        const Vec3 d = centroid() - vertices[0];
        const float dist = std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z);
        return dist * who()->m_fScale / 1.7320508075688772f;
    }
    virtual void genTriangles(std::vector<rt_triangle148> &out) const { // src/cxx/mesh.cpp:119-156
        const Mat4 model = getModel();
        if (isSphere()) {
            out.push_back(make_sphere(apply(model, centroid()), getAverageRadius(), getMatType()));
            return;
        }
        for (size_t i = 0; i + 2 < indices.size(); i += 3)
            out.push_back(make_triangle(apply(model, vertices[indices[i]]), apply(model, vertices[indices[i + 1]]),
                                        apply(model, vertices[indices[i + 2]]), getMatType(), texIdx, texcoords[indices[i]],
                                        texcoords[indices[i + 1]], texcoords[indices[i + 2]]));
    }

    std::string m_strName;
    std::vector<Vec3> vertices;
    std::vector<unsigned> indices;
    std::vector<Vec2> texcoords;
    int matType = 0, texIdx = -1;
    bool m_bIsSphere = false;
    float m_fTranslationX = 0, m_fTranslationY = 0, m_fTranslationZ = 0;
    Mat4 m_mat4Rotation = Mat4::identity();
    float m_fScale = 1.0f; // the reference's default is -1 until a pose file is loaded (mesh.h:85); poses are absent
    mutable const Mesh *m_ptrTopMesh = nullptr;
};
using MESH_PTR = std::shared_ptr<Mesh>;

class FlyWeight : public Mesh { // mesh.h:591-622
  public:
    FlyWeight(std::string name, MESH_PTR orig) : Mesh(std::move(name), {}, {}), _orig(std::move(orig)) {}
    void genTriangles(std::vector<rt_triangle148> &out) const override {
        _orig->m_ptrTopMesh = this;
        _orig->genTriangles(out);
        _orig->m_ptrTopMesh = nullptr;
    }

  private:
    MESH_PTR _orig;
};

// The reference's hand-rolled OBJ reader (include/RT/triangle.h:604-676): "v x y z" and
// "f a/b/c a/b/c a/b/c [a/b/c]" lines only, 1-based indices, quads split (1,2,3) + (1,3,4); getline(buffer, 255), so a
// last line without a newline is dropped; any other face syntax is an error.  Errors come back as text, not exit().
inline bool load_obj(std::istream &in, std::vector<Vec3> &vertices, std::vector<unsigned> &indices, std::string &err) {
    std::string line;
    while (std::getline(in, line)) {
        if (in.eof()) break; // no newline after it: the reference's loop tests eof() before using the buffer
        if (line.size() > 254) {
            err = "OBJ line longer than 254 characters";
            return false;
        }
        const char *buffer = line.c_str();
        if (line.size() >= 2 && buffer[0] == 'v' && buffer[1] == ' ') {
            float f1, f2, f3;
            if (std::sscanf(buffer, "v %f %f %f", &f1, &f2, &f3) != 3) {
                err = "ERROR: vertex not in wanted format in OBJLoader";
                return false;
            }
            vertices.push_back(Vec3{f1, f2, f3});
        } else if (line.size() >= 2 && buffer[0] == 'f' && buffer[1] == ' ') {
            int d1[3], d2[3], d3[3], d4[3];
            const int nt = std::sscanf(buffer, "f %d/%d/%d %d/%d/%d %d/%d/%d %d/%d/%d", &d1[0], &d1[1], &d1[2], &d2[0], &d2[1],
                                       &d2[2], &d3[0], &d3[1], &d3[2], &d4[0], &d4[1], &d4[2]);
            if (nt != 12 && nt != 9) {
                err = "ERROR: I don't know the format of that FaceMtl";
                return false;
            }
            const int tri[2][3] = {{d1[0], d2[0], d3[0]}, {d1[0], d3[0], d4[0]}};
            for (int t = 0; t < (nt == 12 ? 2 : 1); t++)
                for (int k = 0; k < 3; k++) indices.push_back((unsigned)(tri[t][k] - 1)); // f.x() - 1, pbrt_engine.cu:254-258
        }
    }
    for (unsigned i : indices)
        if (i >= vertices.size()) {
            err = "OBJ face index out of range";
            return false;
        }
    return true;
}

using Collection = std::map<std::string, MESH_PTR>; // meshesCollection, main.cpp:46; std::map: name order
inline void registerModel(Collection &c, const MESH_PTR &m) { c.emplace(m->m_strName, m); } // first name wins, mesh.h:232-239

// __genTrianglesFromScene, pbrt_engine.cu:167-178
inline std::vector<rt_triangle148> gather(const Collection &c) {
    std::vector<rt_triangle148> out;
    for (const auto &kv : c)
        if (kv.first == "light") kv.second->genTriangles(out);
    for (const auto &kv : c)
        if (kv.first != "light") kv.second->genTriangles(out);
    return out;
}

// make_a_cornell_light (include/RT/MeshFactory.h:423-429 -> 29-74): flipped quad at y = 10, material 3
inline std::vector<Vec2> quad_uv(bool flip) { // TexCoords of the four corners of every factory quad
    return flip ? std::vector<Vec2>{{1, 1}, {1, 0}, {0, 0}, {0, 1}} : std::vector<Vec2>{{0, 1}, {0, 0}, {1, 0}, {1, 1}};
}
inline MESH_PTR light_mesh() {
    return std::make_shared<Mesh>("light", std::vector<Vec3>{{4, 10, -4}, {-4, 10, -4}, {-4, 10, 4}, {4, 10, 4}},
                                  std::vector<unsigned>{3, 2, 1, 3, 1, 0}, 3, quad_uv(true));
}
// make_a_{xy,yz,zx}_plane(w, h, flip) as addXyPlane / addYzPlane / addZxPlane register it (MeshFactory.h:76-126,
// 181-230, 285-335; main.cpp:43-66): material LAMBERTIAN = 10 (main.cpp:20), texture index 0.  kind: 0 xy, 1 yz, 2 zx
inline MESH_PTR plane_mesh(int kind, float w = 1.0f, float h = 1.0f, int texIdx = 0, bool flip = false) {
    const float hw = w / 2, hh = h / 2;
    std::vector<Vec3> v;
    if (kind == 0)
        v = {{-hw, hh, 0}, {-hw, -hh, 0}, {hw, -hh, 0}, {hw, hh, 0}};
    else if (kind == 1)
        v = {{0, hh, -hw}, {0, -hh, -hw}, {0, -hh, hw}, {0, hh, hw}};
    else
        v = {{hh, 0, -hw}, {-hh, 0, -hw}, {-hh, 0, hw}, {hh, 0, hw}};
    const char *names[3] = {"xy", "yz", "zx"};
    auto m = std::make_shared<Mesh>(std::string(names[kind]) + (flip ? "_flip" : ""), v,
                                    flip ? std::vector<unsigned>{3, 2, 1, 3, 1, 0} : std::vector<unsigned>{0, 1, 2, 0, 2, 3}, 10,
                                    quad_uv(flip));
    m->texIdx = texIdx;
    return m;
}
// stand-in for test01.png (absent): an n x n RGBA8 checkerboard of 8 x 8 squares, integers only
inline std::vector<unsigned char> default_texture(int n = 64) {
    std::vector<unsigned char> t((size_t)n * n * 4);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const bool on = ((j * 8 / n + i * 8 / n) % 2) == 0;
            unsigned char *p = &t[4 * ((size_t)j * n + i)];
            p[0] = on ? 230 : 40;
            p[1] = on ? 230 : 60;
            p[2] = on ? 220 : 160;
            p[3] = 255;
        }
    return t;
}
// stand-in for models/cube.obj (absent): the canonical +-1 cube, 12 triangles
inline MESH_PTR unit_cube_mesh(const std::string &name) {
    return std::make_shared<Mesh>(
        name, std::vector<Vec3>{{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}},
        std::vector<unsigned>{0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 2, 3, 7, 2, 7, 6, 1, 2, 6, 1, 6, 5, 0, 4, 7, 0, 7, 3}, 8);
}

// rand() as the reference uses it, counted: the builder continues the same sequence (rt_build_bvh's second argument)
struct CountedRand {
    long long calls = 0;
    float drand48() { // the reference's wrapper, include/RT/triangle.h:12-15
        calls++;
        return rand() / (float)RAND_MAX;
    }
};

// main.cpp:408-500 (see scene_graph.py's module text for what is assumed where the reference's files are absent).
// Uses the process's rand(), never seeded, as the reference does.
inline Collection default_scene(CountedRand &rng) {
    Collection col;
    registerModel(col, light_mesh());
    MESH_PTR cube = unit_cube_mesh("cube"); // mesh3: matType 8, flagged as a sphere (main.cpp:435-445)
    cube->m_bIsSphere = true;
    registerModel(col, cube);
    std::vector<Vec3> centers{{0, -1000, 0}, {0, 1, 0}, {-4, 1, 0}, {4, 1, 0}};
    std::vector<float> radius{1000.0f, 1.0f, 1.0f, 1.0f};
    std::vector<int> mats{0, 7, 4, 6};
    std::vector<bool> flags(4, true);
    auto local_random = [&]() -> float { return (float)(2 * (rng.drand48() - 0.5)); };
    for (int a = -11; a < 11; a += 2) {
        for (int b = -11; b < 11; b += 2) {
            const float choose_mat = rng.drand48();
            const double r = 0.2 + rng.drand48() / 10;
            const float cx = (float)(a + 0.2 * local_random()); // x argument drawn first
            const float cz = (float)(b + 0.2 * local_random());
            const Vec3 center{cx, (float)r, cz};
            const Vec3 d = center - Vec3{4, 0.2f, 0};
            const float len = std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z);
            if (len > 0.9) {
                centers.push_back(center);
                radius.push_back((float)r);
                flags.push_back(!(rng.drand48() < 0.5));
                if (choose_mat < 0.3)
                    mats.push_back((int)(10 + rng.drand48() * 20));
                else if (choose_mat < 0.9)
                    mats.push_back((int)(30 + rng.drand48() * 20));
                else
                    mats.push_back(7);
            }
        }
    }
    for (size_t i = 0; i < centers.size(); i++) {
        auto fw = std::make_shared<FlyWeight>("c" + std::to_string(i), cube);
        fw->matType = mats[i];
        fw->setPosition(centers[i]);
        fw->m_fScale = radius[i];
        fw->m_bIsSphere = flags[i];
        registerModel(col, fw);
    }
    return col;
}

} // namespace rtscene
