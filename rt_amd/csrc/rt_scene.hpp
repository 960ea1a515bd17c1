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
#include <sstream>
#include <stdexcept>
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
    virtual Vec3 centroid() const { // resolvePosition, mesh.h:243-251 (0/0 for a mesh without vertices, as there)
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

// The positioned overloads make_a_xy_plane(lbx, lby, rtx, rty, z, flip) / make_a_yz_plane(lbz, lby, rtz, rty, x, flip) /
// make_a_zx_plane(lbz, lbx, rtz, rtx, y, flip) (MeshFactory.h:128-179, 232-283, 338-390): a quad between two corners in
// the plane `at`; material 0 and no texture index until the caller sets them (the Mesh constructor's defaults).
inline MESH_PTR positioned_plane(int kind, float a0, float b0, float a1, float b1, float c, bool flip = false) {
    std::vector<Vec3> v;
    if (kind == 0) // (lbx, lby, rtx, rty, z)
        v = {{a0, b1, c}, {a0, b0, c}, {a1, b0, c}, {a1, b1, c}};
    else if (kind == 1) // (lbz, lby, rtz, rty, x)
        v = {{c, b1, a0}, {c, b0, a0}, {c, b0, a1}, {c, b1, a1}};
    else // (lbz, lbx, rtz, rtx, y)
        v = {{b1, c, a0}, {b0, c, a0}, {b0, c, a1}, {b1, c, a1}};
    const char *names[3] = {"xy", "yz", "zx"};
    return std::make_shared<Mesh>(std::string(names[kind]) + (flip ? "_flip" : ""), v,
                                  flip ? std::vector<unsigned>{3, 2, 1, 3, 1, 0} : std::vector<unsigned>{0, 1, 2, 0, 2, 3}, 0,
                                  quad_uv(flip));
}

// Six faces under one pose (mesh.h:457-535).  Every face keeps the cube as its top mesh for good, so a face's triangles
// take the CUBE's model matrix and the CUBE's material (getMatType / getModel / isSphere look at the top mesh) --
// setMatTypeSubMesh writes the faces' own field, which nothing reads while the top mesh is set; the texture index is
// the face's own.
class CubeMesh : public Mesh {
  public:
    CubeMesh(std::string name, MESH_PTR front, MESH_PTR back, MESH_PTR left, MESH_PTR right, MESH_PTR up, MESH_PTR bottom)
        : Mesh(std::move(name), {}, {}), faces{std::move(front), std::move(back), std::move(left), std::move(right), std::move(up), std::move(bottom)} {
        for (const MESH_PTR &f : faces) f->m_ptrTopMesh = this; // APPLY_TO_ALL's order, mesh.h:492-497
    }
    void setMatTypeSubMesh(int m) {
        for (const MESH_PTR &f : faces) f->matType = m;
    }
    void genTriangles(std::vector<rt_triangle148> &out) const override {
        for (const MESH_PTR &f : faces) f->genTriangles(out);
    }
    Vec3 centroid() const override { // resolvePosition, mesh.h:511-522: the mean of the six faces' vertex means
        Vec3 s;
        for (const MESH_PTR &f : faces) {
            const Vec3 c = f->centroid();
            s.x += c.x;
            s.y += c.y;
            s.z += c.z;
        }
        return Vec3{s.x / 6.0f, s.y / 6.0f, s.z / 6.0f};
    }
    void shareMeshData(Mesh &target) const { // mesh.h:165-179, 524-526: vertices and re-based indices; no pose travels
        for (const MESH_PTR &f : faces) {
            const unsigned base = (unsigned)target.vertices.size();
            target.vertices.insert(target.vertices.end(), f->vertices.begin(), f->vertices.end());
            target.texcoords.insert(target.texcoords.end(), f->texcoords.begin(), f->texcoords.end());
            for (unsigned i : f->indices) target.indices.push_back(i + base);
        }
    }
    std::vector<MESH_PTR> faces;
};

// make_a_cube_mesh(lb, rt, ry, translation, _setup), MeshFactory.h:392-421
inline std::shared_ptr<CubeMesh> make_a_cube_mesh(Vec3 lb, Vec3 rt, float ry, Vec3 translation, const std::string &name = "None") {
    auto cube = std::make_shared<CubeMesh>(name, positioned_plane(0, lb.x, lb.y, rt.x, rt.y, rt.z, false),
                                           positioned_plane(0, lb.x, lb.y, rt.x, rt.y, lb.z, true),
                                           positioned_plane(1, lb.z, lb.y, rt.z, rt.y, lb.x, false),
                                           positioned_plane(1, lb.z, lb.y, rt.z, rt.y, rt.x, true),
                                           positioned_plane(2, lb.z, lb.x, rt.z, rt.x, rt.y, false),
                                           positioned_plane(2, lb.z, lb.x, rt.z, rt.x, lb.y, true));
    cube->rotationCommand(1, ry);
    cube->translateXCommand(translation.x);
    cube->translateYCommand(translation.y);
    cube->translateZCommand(translation.z);
    for (const MESH_PTR &f : cube->faces) f->texIdx = 3;
    return cube;
}

// Stand-in for .\\cache\\cube_locations.txt (absent): first line the cubes' footprint, then "z x" per cube -- the letters
// R and T on a 5 x 7 dot matrix (the same text as scene_graph.py's SIGNATURE_LOCATIONS).
inline std::string signature_locations() {
    const char *rows[5] = {"1110.11111", "1001.00100", "1110.00100", "1010.00100", "1001.00100"};
    std::string t = "60\n";
    for (int row = 0; row < 5; row++)
        for (int col = 0; rows[row][col]; col++)
            if (rows[row][col] == '1') t += std::to_string(-150 + 75 * row) + " " + std::to_string(-460 + 75 * col) + "\n";
    return t;
}
// signature_reader, main.cpp:334-369.  r stands for the reference's rand() / RAND_MAX after srand(time(nullptr)): a
// parameter here; the process's rand() sequence is not reseeded.
inline MESH_PTR signature_mesh(const std::string &text, double r = 0.5) {
    std::istringstream in(text);
    std::string line;
    std::getline(in, line);
    const float width = std::stof(line);
    auto list = std::make_shared<Mesh>("meshList", std::vector<Vec3>{}, std::vector<unsigned>{});
    const float hw = width / 2.5;
    const float h = width * (r * 2.0f + 1.0);
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        std::vector<float> v;
        for (std::string t; iss >> t;) v.push_back(std::stof(t));
        if (v.empty()) continue;
        if (v.size() != 2) throw std::runtime_error("failed in signature reader.");
        const float z = v[0], x = v[1];
        make_a_cube_mesh(Vec3{x - hw, 0, z - hw}, Vec3{x + hw, h, z + hw}, 0.0f, Vec3{0, 0, 0})->shareMeshData(*list);
    }
    return list;
}

// stand-ins for the four images the textured scene names (absent): n x n RGBA8, integers only (scene_graph.py's
// stand_in_texture): 0 the checkerboard above, 1 vertical stripes, 2 a two-way gradient, 3 a grid of dots
inline std::vector<unsigned char> stand_in_texture(int index, int n = 64) {
    if (index == 0) return default_texture(n);
    std::vector<unsigned char> t((size_t)n * n * 4);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            unsigned char *p = &t[4 * ((size_t)j * n + i)];
            if (index == 1) {
                const bool on = ((i * 8 / n) % 2) == 0;
                p[0] = on ? 200 : 60, p[1] = on ? 180 : 90, p[2] = on ? 60 : 200;
            } else if (index == 2) {
                p[0] = (unsigned char)(i * 255 / (n - 1)), p[1] = (unsigned char)(j * 255 / (n - 1)), p[2] = 128;
            } else {
                const bool on = ((i * 8) % n) * 2 < n && ((j * 8) % n) * 2 < n;
                p[0] = on ? 250 : 90, p[1] = on ? 240 : 140, p[2] = on ? 200 : 90;
            }
            p[3] = 255;
        }
    return t;
}

// The reference's second scene, the `#ifdef CAT` block of main() (main.cpp:596-662); see scene_graph.py's cat_meshes
// for the stand-ins.  Consumes no rand() call.
inline Collection cat_scene(double r = 0.5) {
    Collection col;
    auto box02 = make_a_cube_mesh(Vec3{0, 0, 0}, Vec3{800, 450, 450}, 15, Vec3{0, 0, 0}, "box02");
    box02->setMatTypeSubMesh(3); // LIGHT -- on the faces' own field, which the cube's material (0) overrides
    registerModel(col, box02);
    MESH_PTR s1 = unit_cube_mesh("s1"); // Model("models/sphere.obj", "s1"): material 5, flagged as a sphere
    s1->matType = 5;
    s1->m_bIsSphere = true;
    registerModel(col, s1);
    const float height = 800;
    auto named = [&](MESH_PTR m, const char *name, int mat, int tex) {
        m->m_strName = name;
        m->matType = mat;
        m->texIdx = tex;
        registerModel(col, m);
    };
    named(positioned_plane(1, -1000, 0, 1000, height, -1000, true), "left_wall", 4, 1);
    named(positioned_plane(1, -1000, 0, 1000, height, 1000), "right_wall", 3, 2);
    named(positioned_plane(0, -1000, 0, 1000, height, -1000), "back_wall", 3, 0);
    registerModel(col, light_mesh());
    MESH_PTR m = signature_mesh(signature_locations(), r);
    m->matType = 6; // METAL
    registerModel(col, m);
    named(positioned_plane(2, -1000, -1000, 1000, 1000, 0, false), "floor", 0, -1); // setMatType(10), then setMatType(0)
    return col;
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
