"""Writes the authored scene files under assets/ (deterministic; outputs are committed).

cornell.xml  — BASELINE.json configs[0]: Cornell-style box in the reference's XML schema (one material per
               <Mesh>, so one tiny OBJ per colour): 8 wall triangles, a 2-triangle emissive quad, 2 spheres.
               Camera for it: pos (0,1,3.4) fwd (0,0,-1) up (0,1,0) vfov 40 (tests pass it via uniforms).
bunny20.xml  — BASELINE.json configs[2]: 20 instances of assets/bunny.obj on a 5x4 grid (99,360 triangles)
               + ground sphere + emissive sphere, reference camera.
glass.xml    — small mirror + glass scene for the Scatter.h BSDF switch (configs[4] materials).
"""
import os

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def quad_obj(name, v):
    with open(os.path.join(ROOT, name), "w") as f:
        f.write("# %s - two triangles\n" % name)
        for p in v:
            f.write("v %.6f %.6f %.6f\n" % p)
        f.write("f 1 2 3\nf 1 3 4\n")


def main():
    os.makedirs(ROOT, exist_ok=True)
    # unit Cornell box: x in [-1,1], y in [0,2], z in [-1,1], open towards +z
    quad_obj("cornell_floor.obj", [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)])
    quad_obj("cornell_back.obj", [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)])
    quad_obj("cornell_left.obj", [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)])
    quad_obj("cornell_right.obj", [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)])
    # The light is tilted by 1 mm on purpose: the reference's slab test rejects a box when tMax <= tMin
    # (PathTracing.h:68), so an axis-aligned flat quad that ends up alone in a BVH leaf (zero-thickness box) can
    # never be hit — faithfully reproduced by the oracle and the HIP path (see DESIGN.md "Reference quirks").
    quad_obj("cornell_light.obj", [(-0.35, 1.98, 0.35), (0.35, 1.98, 0.35), (0.35, 1.979, -0.35), (-0.35, 1.979, -0.35)])
    with open(os.path.join(ROOT, "cornell.xml"), "w") as f:
        f.write("""<Scene>
    <!-- Cornell-style box in the reference schema (SURVEY.md 8d config 1). Open front/top: misses see the sky. -->
    <Mesh file="cornell_floor.obj" position="0,0,0" scale="1" albedo="0.73,0.73,0.73" emission="0,0,0" materialType="0" emissionPower="0" />
    <Mesh file="cornell_back.obj" position="0,0,0" scale="1" albedo="0.73,0.73,0.73" emission="0,0,0" materialType="0" emissionPower="0" />
    <Mesh file="cornell_left.obj" position="0,0,0" scale="1" albedo="0.65,0.05,0.05" emission="0,0,0" materialType="0" emissionPower="0" />
    <Mesh file="cornell_right.obj" position="0,0,0" scale="1" albedo="0.12,0.45,0.15" emission="0,0,0" materialType="0" emissionPower="0" />
    <Mesh file="cornell_light.obj" position="0,0,0" scale="1" albedo="0,0,0" emission="1,1,1" materialType="0" emissionPower="5" />
    <Sphere position="-0.45,0.35,-0.3" radius="0.35" albedo="0.73,0.73,0.73" emission="0,0,0" materialType="0" emissionPower="0" />
    <Sphere position="0.45,0.3,0.3" radius="0.3" albedo="0.73,0.73,0.73" emission="0,0,0" materialType="0" emissionPower="0" />
</Scene>
""")
    with open(os.path.join(ROOT, "bunny20.xml"), "w") as f:
        f.write("<Scene>\n    <!-- 20 bunnies, 5x4 grid, spacing 16, scale 10 (SURVEY.md 8d config 3) -->\n")
        f.write('    <Sphere position="0,-10000,0" radius="10000" albedo="0.8,0.8,0.8" emission="0,0,0" materialType="0" emissionPower="0" />\n')
        f.write('    <Sphere position="0,60,-20" radius="10" albedo="0.0,0.0,0.0" emission="1.0,0.9,0.7" materialType="0" emissionPower="5" />\n')
        cols = [(0.9, 0.5, 0.3), (0.3, 0.6, 0.9), (0.5, 0.8, 0.4), (0.8, 0.8, 0.3), (0.7, 0.4, 0.7)]
        for j in range(4):
            for i in range(5):
                x = (i - 2) * 16.0
                z = -j * 16.0
                c = cols[(i + j) % len(cols)]
                f.write('    <Mesh file="bunny.obj" position="%g,0,%g" scale="10.0" albedo="%g,%g,%g" emission="0,0,0" materialType="0" emissionPower="0" />\n'
                        % (x, z, c[0], c[1], c[2]))
        f.write("</Scene>\n")
    with open(os.path.join(ROOT, "glass.xml"), "w") as f:
        f.write("""<Scene>
    <!-- mirror (materialType < 0) and glass (materialType = index of refraction) per Scatter.h:22-43 -->
    <Sphere position="0,-10000,0" radius="10000" albedo="0.8,0.8,0.8" emission="0,0,0" materialType="0" emissionPower="0" />
    <Sphere position="-14,12,0" radius="12" albedo="0.95,0.95,0.95" emission="0,0,0" materialType="-1" emissionPower="0" />
    <Sphere position="14,12,0" radius="12" albedo="1.0,1.0,1.0" emission="0,0,0" materialType="1.5" emissionPower="0" />
    <Sphere position="0,45,-10" radius="8" albedo="0.0,0.0,0.0" emission="1.0,0.9,0.7" materialType="0" emissionPower="5" />
    <Mesh file="bunny.obj" position="0,0,18" scale="6.0" albedo="0.9,0.5,0.3" emission="0,0,0" materialType="0" emissionPower="0" />
</Scene>
""")


if __name__ == "__main__":
    main()
