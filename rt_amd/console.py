"""Headless console: the caller of the trace loop.

The reference drives ``PbrtEngine`` from a console thread that reads one command per line (main.cpp:68-332) next to a
GLFW window.  This module keeps the commands that decide what the trace loop sees -- the selected mesh and its pose
(``m``, ``mx/my/mz``, ``rx/ry/rz``, ``s``, ``reset``, ``radius``), the GL camera's position / front / aperture /
focus distance (``camera ...``), the frame parameters (``render param``, ``render hd``), and ``render`` /
``render -cache`` / ``render status`` -- and drops the window.  The same words print the same lines.

What a headless run cannot mirror is said where it is dropped: mouse / WASD camera motion (the state is settable
through ``Console.camera`` instead), the outline / ambient / diffuse / specular commands (GL shading only: accepted,
no effect on the trace loop, as in the reference), ``new xy|yz|zx`` (textured planes: textures are outside the hot
path, rt_abi.h ``RT_E_UNSUPPORTED``) and the binary camera snapshot under ``.\\cache`` (``camera save|load`` keep the
state in a JSON file instead).

One more thing the console owns: glibc ``rand()`` is never seeded by the reference, scene setup and every BVH
build draw from the same sequence, so the n-th ``render`` of a session starts its build where the previous one
stopped.  ``Console.rand_calls`` carries that position into ``rt_build_bvh(ctx, rand_calls_before)``.
"""
from __future__ import annotations

import io
import json
import sys

import numpy as np

from . import scene_graph as sg

F = np.float32


class GlCamera:
    """The fields of the GL ``Camera`` singleton that ``__renderScene`` reads (include/RT/camera.h:32-47, 110-159,
    245-272; pbrt_engine.cu:104-106): lookfrom = position, lookat = position + front."""

    _VIEWS = {  # lookFrom{Pos,Neg}{X,Y,Z}: position, front
        "posx": ((3.0, 0.0, 0.0), (-1.0, 0.0, 0.0)),
        "negx": ((-3.0, 0.0, 0.0), (1.0, 0.0, 0.0)),
        "posy": ((0.0, 3.0, 0.0), (0.1, -1.0, 0.1)),
        "negy": ((0.0, -3.0, 0.0), (-0.1, 1.0, -0.1)),
        "posz": ((0.0, 0.0, 3.0), (0.0, 0.0, -1.0)),
        "negz": ((0.0, 0.0, -3.0), (0.0, 0.0, 1.0)),
    }

    def __init__(self):
        self.position = np.array([0.0, 0.0, 3.0], dtype=F)  # POS_Z
        self.front = np.array([0.0, 0.0, -1.0], dtype=F)  # FRONT
        self.zoom = F(45.0)  # passed to camera_update, which ignores it (pbrt_engine.cu:629)
        self.dist_to_focus = F(10.0)
        self.aperture = F(0.0)
        self.move_speed = F(2.5)
        self.mouse_sens = F(0.1)

    def look_from(self, which: str):
        p, f = self._VIEWS[which]
        self.position = np.array(p, dtype=F)
        self.front = np.array(f, dtype=F)

    def state(self) -> dict:
        return {"position": [float(x) for x in self.position], "front": [float(x) for x in self.front],
                "zoom": float(self.zoom), "dist_to_focus": float(self.dist_to_focus), "aperture": float(self.aperture)}

    def restore(self, s: dict):
        self.position = np.array(s["position"], dtype=F)
        self.front = np.array(s["front"], dtype=F)
        self.zoom = F(s["zoom"])
        self.dist_to_focus = F(s["dist_to_focus"])
        self.aperture = F(s["aperture"])

    def dist_to(self, mesh: sg.Mesh) -> np.float32:
        """getDistToFocus(Mesh*) (camera.h:261-265): |position - mesh.getPosition()|."""
        d = (self.position - mesh.get_position()).astype(F)
        return F(np.sqrt((d[0] * d[0] + d[1] * d[1] + d[2] * d[2]).astype(F)))


def _fmt(v) -> str:
    """operator<< of a float with the default format (%g)."""
    return "%g" % float(v)


class Console:
    """One console session over one engine.  ``execute(line)`` runs one command and returns what it printed."""

    def __init__(self, engine, meshes=None, rand_calls: int = 0, out_path: str = "./output.ppm",
                 snapshot_path: str = "./camera_snapshot.json", textures=None):
        self.engine = engine
        self.meshes = {}
        for m in (meshes or []):
            self.meshes.setdefault(m.name, m)  # registerThisModel, mesh.h:232-239
        self.rand_calls = int(rand_calls)
        self.camera = GlCamera()
        self.curr = None
        self.quick_shot = None
        self.out_path = out_path
        self.snapshot_path = snapshot_path
        self.loop = True
        self.last_frame = None
        self.last_triangles = None
        self._texture0_set = False
        # images the scene names (index -> RGBA8): handed to the engine before the first render (genTextures, main.cpp:715)
        self._scene_textures = dict(textures or {})

    # ---------------------------------------------------------------- render
    def _render(self, build_scene: bool, out):
        e = self.engine
        print(f"rendering a {e.nx}[width]x{e.ny}[height] image with {e.ns} samples per pixel in {e.tx}x{e.ty} blocks.",
              file=out)
        for idx, rgba in sorted(self._scene_textures.items()):
            e.setTexture(idx, rgba)
            self._texture0_set = self._texture0_set or idx == 0
        self._scene_textures = {}
        if build_scene:
            print("INFO: generate scene.", file=out)
            tris = sg.gather(self.meshes.values())
            e.setTriangles(tris)
            print(f"Total num of Triangles: {tris.shape[0]}", file=out)  # __genBvhNodes, pbrt_engine.cu:183
            e.buildBvh(self.rand_calls)
            self.rand_calls += sg.build_rand_calls(int(tris.shape[0]))
            self.last_triangles = tris
        else:
            if self.last_triangles is None:
                raise RuntimeError("render -cache before any scene was generated")
            print("INFO: using scene generated.", file=out)
        c = self.camera
        e.setCamera(c.position, (c.position + c.front).astype(F), float(c.zoom), float(c.dist_to_focus), float(c.aperture))
        fb = e.render()
        print(f"took {_fmt(e.last_stats.kernel_ms / 1000.0)} seconds.", file=out)
        if self.out_path:
            from .engine import write_ppm

            write_ppm(self.out_path, fb)
        print("PbrtEngine: rendering finished.", file=out)
        self.last_frame = fb

    # ---------------------------------------------------------------- one line
    def execute(self, line: str) -> str:
        out = io.StringIO()
        try:
            self._execute(line[:29].split(), out)  # cin.getline(input, 30), main.cpp:80
        except Exception as ex:  # catch (std::exception&), main.cpp:326-328
            print(f"Console Error: {ex}", file=out)
        return out.getvalue()

    def _execute(self, tokens, out):
        if not tokens:
            return
        first = tokens[0]
        cam = self.camera
        if first == "q":
            print("bye, press any buttion to end.", file=out)
            self.loop = False
            return
        if first == "m":
            model = tokens[1]
            if model not in self.meshes:
                print(f"ERROR: Model {model} not found.", file=out)
                return
            self.curr = self.meshes[model]
            print("setting model finished.", file=out)
            return
        if first == "camera":
            second = tokens[1]
            if second == "-h":
                print("[options]: posx, negx, log, restore, info, aperture, \ndist_to_focus, speed, sens", file=out)
            if second in GlCamera._VIEWS:
                cam.look_from(second)
            elif second == "log":
                self.quick_shot = cam.state()
                print("camera quick shot finished.", file=out)
            elif second == "restore":
                if self.quick_shot is not None:
                    cam.restore(self.quick_shot)
                print("camera state restored.", file=out)
            elif second == "info":
                print(" ".join(_fmt(x) for x in cam.position), file=out)
                print(" ".join(_fmt(x) for x in cam.front), file=out)
            elif second == "aperture":
                cam.aperture = F(float(tokens[2]))
            elif second == "dist_to_focus":
                if len(tokens) > 2:
                    cam.dist_to_focus = F(float(tokens[2]))
                else:
                    if self.curr is None:
                        raise RuntimeError("no model selected")  # the reference dereferences a null Mesh* here
                    f = cam.dist_to(self.curr)
                    cam.dist_to_focus = f
                    print(f"INFO: set camera distance_to_focus to {_fmt(f)}", file=out)
            elif second == "speed":
                if len(tokens) < 3:
                    print(f"curr camera speed: {_fmt(cam.move_speed)}", file=out)
                    return
                cam.move_speed = F(float(tokens[2]))
            elif second == "sens":
                cam.mouse_sens = F(float(tokens[2]))
            elif second == "save":
                with open(self.snapshot_path, "w") as fh:
                    json.dump(cam.state(), fh)
            elif second == "load":
                with open(self.snapshot_path) as fh:
                    cam.restore(json.load(fh))
            return
        if first == "render":
            e = self.engine
            if len(tokens) >= 2 and tokens[1] == "status":
                print("PbrtEngine: rendering finished.", file=out)  # renders are joined before the prompt returns
            elif len(tokens) >= 2 and tokens[1] == "param":
                e.setNx(int(tokens[2]))
                e.setNy(int(tokens[3]))
                e.setNs(int(tokens[4]))
                self._pixels_changed()
            elif len(tokens) >= 2 and tokens[1] == "hd":
                e.setNx(1280)
                e.setNy(720)
                e.setNs(100)
                self._pixels_changed()
            else:
                self._render(not (len(tokens) >= 2 and tokens[1] == "-cache"), out)
            return
        if self.curr is None:
            return
        curr = self.curr
        if first == "mx":
            curr.translate_command(x=float(tokens[1]))
        elif first == "my":
            curr.translate_command(y=float(tokens[1]))
        elif first == "mz":
            curr.translate_command(z=float(tokens[1]))
        elif first in ("rx", "ry", "rz"):
            curr.rotation_command("xyz".index(first[1]), float(tokens[1]))
        elif first == "s":
            curr.scale_command(float(tokens[1]))
        elif first == "reset":
            curr.reset_all()
        elif first == "radius":
            print(f"radius => {_fmt(curr.average_radius())}", file=out)
        elif first == "new":  # addXyPlane / addYzPlane / addZxPlane, main.cpp:309-319 (needs a selected model, as there)
            kind = {"xy": "xy", "yx": "xy", "yz": "yz", "zy": "yz", "zx": "zx", "xz": "zx"}.get(tokens[1])
            if kind:
                m = sg.plane_mesh(kind)
                self.meshes.setdefault(m.name, m)
                if not self._texture0_set:  # the planes name texture 0 ("test01.png" in the reference)
                    self.engine.setTexture(0, sg.default_texture())
                    self._texture0_set = True
        elif first in ("ambient", "specular", "diffuse"):
            [float(t) for t in tokens[1:4]]  # GL shading colours: parsed as the reference does, unused by the trace loop
        # hide / show / log position / viewport: window-side state only

    def _pixels_changed(self):
        e = self.engine
        if e.nx_ny_changed:  # nXnYChanged(), pbrt_engine.h:93-97
            e.nx_ny_changed = False
            e.uponPixelNumChange()

    def run(self, stream=sys.stdin, out=sys.stdout):
        """The console thread's loop (main.cpp:76-331)."""
        while self.loop:
            out.write(">>>:")
            out.flush()
            line = stream.readline()
            if not line:
                break
            out.write(self.execute(line.rstrip("\n")))
        out.write("Quitting this thread.\n")


def main(argv=None):
    """``python -m rt_amd.console``: the default scene (scene_graph.default_meshes) or the reference's second one
    (``--scene cat``, scene_graph.cat_meshes) behind the console."""
    import argparse

    from .engine import PbrtEngine

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="./output.ppm")
    ap.add_argument("--obj", action="append", default=[], metavar="NAME=FILE[:MAT]",
                    help="add a mesh through the reference's own OBJ reader (scene_graph.load_obj)")
    ap.add_argument("--scene", choices=["default", "cat"], default="default",
                    help="cat = the reference's second scene (main.cpp:596-662): textured light walls, a textured metal wall, "
                         "the signature boxes; stand-in textures 0..3")
    ap.add_argument("--camera", default=None, metavar="PX,PY,PZ,FX,FY,FZ",
                    help="position and front of the GL camera at start (stands in for the binary camera snapshot the "
                         "reference loads, main.cpp:663: absent)")
    args = ap.parse_args(argv)
    meshes, calls = sg.cat_meshes() if args.scene == "cat" else sg.default_meshes()
    textures = {i: sg.stand_in_texture(i) for i in range(4)} if args.scene == "cat" else None
    for spec in args.obj:
        name, _, rest = spec.partition("=")
        path, _, mat = rest.rpartition(":") if ":" in rest else (rest, "", "0")
        with open(path, newline="") as fh:
            meshes.append(sg.mesh_from_obj(name, fh.read(), int(mat)))
    con = Console(PbrtEngine(device=args.device), meshes, calls, out_path=args.out, textures=textures)
    if args.camera:
        f = [float(x) for x in args.camera.split(",")]
        if len(f) != 6:
            ap.error("--camera PX,PY,PZ,FX,FY,FZ")
        con.camera.position, con.camera.front = np.array(f[:3], dtype=F), np.array(f[3:], dtype=F)
    con.run()


if __name__ == "__main__":
    main()
