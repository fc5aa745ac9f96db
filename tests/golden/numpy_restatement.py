"""A second, independent restatement of bevyray's fragment shader -- numpy float32 scalars,
written from the WGSL (reference assets/shaders/raytrace.wgsl, random.wgsl, const.wgsl), NOT from
oracle/bevyray_oracle.c.  It exists to pin the C oracle on complete paths (BVH walk, sphere hits,
metal / glass / diffuse scatter, sky, gamma, averaging, depth blend): make_golden.py renders tiny
frames with it and commits them; tests check the oracle (CPU suite) and the HIP path (GPU
suite) against those frames bit for bit.  Far too slow for anything but a few hundred pixels.

Same numeric policy as DESIGN.md section 3 (every np.float32 operation is separately rounded;
min/max = minNum/maxNum via np.fmin/np.fmax; pow(x,5) as multiplies; tan in double on the host).

POLICY: three choices of that policy are things WGSL leaves to the implementation and that the real
naga/wgpu lowering may make differently (nothing in the reference pins them).  The default
is what the product implements; the alternatives exist so that fixtures for them are on file the day a real
wgpu run can be compared (tests/golden/policy_frames.npz):
    or_short_circuit   raytrace.wgsl:269 `cannot_refract || reflectance(..) > rngNextFloat(..)`: False (default) =
                       both operands evaluated, the RNG draw always happens; True = WGSL-spec short circuit, no
                       draw when cannot_refract
    minmax             raytrace.wgsl:391-394, 263, 405 min()/max(): "minnum" (default: a NaN operand yields the
                       other one) or "select" (min(a,b) = b < a ? b : a, max(a,b) = a < b ? b : a: a NaN in the
                       SECOND operand is dropped, a NaN in the first is returned)
    pow                raytrace.wgsl:415 pow(1 - cosine, 5.0): "mul" (default: (x*x)*(x*x)*x) or "exp2log2"
                       (WGSL's definition exp2(5 * log2(x)), evaluated in double and rounded to f32 once)
"""
import math

import numpy as np

F = np.float32
U = np.uint32
INF = F(3.40282347e+38)          # const.wgsl:2


def v3(x, y, z):
    return np.array([x, y, z], F)


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def normalize(v):
    ln = np.sqrt(dot(v, v), dtype=F)
    return v3(v[0] / ln, v[1] / ln, v[2] / ln)


def cross(a, b):
    return v3(F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0]))


class Rng:
    """random.wgsl:3-15"""

    def __init__(self, state):
        self.state = int(state) & 0xFFFFFFFF

    def next_float(self):
        old = (self.state + 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((old >> ((old >> 28) + 4)) ^ old) * 277803737) & 0xFFFFFFFF
        self.state = ((word >> 22) ^ word) & 0xFFFFFFFF
        return F(F(self.state) / F(0xFFFFFFFF))

    def unit_ball(self):
        """random.wgsl:17-30: a point inside the unit ball"""
        while True:
            x, y, z = self.next_float(), self.next_float(), self.next_float()
            p = v3(F(F(2.0) * x) - F(1.0), F(F(2.0) * y) - F(1.0), F(F(2.0) * z) - F(1.0))
            if dot(p, p) <= F(1.0):
                return p


def f32_to_u32(f):
    if not (f > 0):
        return 0
    if f >= F(4294967296.0):
        return 0xFFFFFFFF
    return int(math.floor(float(f)))


DEFAULT_POLICY = dict(or_short_circuit=False, minmax="minnum", pow="mul")


class Shader:
    """models / materials / bvh: structured arrays with the reference's wire layouts; camera, window: single records."""

    def __init__(self, models, materials, bvh, camera, window, level, policy=None):
        self.policy = dict(DEFAULT_POLICY, **(policy or {}))
        self.models, self.materials, self.bvh = models, materials, bvh
        self.cam, self.win, self.level = camera, window, int(level)
        self.tan_half_fov = F(math.tan(float(F(camera["fov"]) * F(0.5))))
        self.rays = 0

    def fmin(self, a, b):
        if self.policy["minmax"] == "select":
            with np.errstate(all="ignore"):
                return np.where(b < a, b, a).astype(F)[()]
        return np.fmin(a, b)

    def fmax(self, a, b):
        if self.policy["minmax"] == "select":
            with np.errstate(all="ignore"):
                return np.where(a < b, b, a).astype(F)[()]
        return np.fmax(a, b)

    # raytrace.wgsl:387-398
    def ray_bounding_dst(self, o, d, bmin, bmax):
        with np.errstate(all="ignore"):
            inv = v3(F(1.0) / d[0], F(1.0) / d[1], F(1.0) / d[2])
            tmin = (bmin - o).astype(F) * inv
            tmax = (bmax - o).astype(F) * inv
            t1, t2 = self.fmin(tmin, tmax), self.fmax(tmin, tmax)
            t_near = self.fmax(self.fmax(t1[0], t1[1]), t1[2])
            t_far = self.fmin(self.fmin(t2[0], t2[1]), t2[2])
        hit = (t_far >= t_near) and (t_far > 0)
        return (t_near if t_near > 0 else F(0.0)) if hit else INF

    # raytrace.wgsl:371-383
    def hit_sphere(self, m, o, d):
        oc = (m["position"].astype(F) - o).astype(F)
        a = dot(d, d)
        h = dot(d, oc)
        c = F(dot(oc, oc) - F(F(m["radius"]) * F(m["radius"])))
        disc = F(F(h * h) - F(a * c))
        if disc < 0:
            return F(-1.0)
        return F(F(h - np.sqrt(disc, dtype=F)) / a)

    # raytrace.wgsl:313-362
    def raycast(self, o, d):
        self.rays += 1
        closest = dict(distance=INF, position=v3(0, 0, 0), normal=v3(0, 0, 0), material=0, front_face=True)
        stack = [0] * 32
        si = 1
        while 0 < si < 32:
            si -= 1
            node = self.bvh[stack[si]]
            if node["model_count"] > 0:
                for mi in range(int(node["index"]), int(node["index"]) + int(node["model_count"])):
                    m = self.models[mi]
                    t = self.hit_sphere(m, o, d)
                    if t != F(-1.0) and t > F(0.001) and t < closest["distance"]:
                        pos = (o + (t * d).astype(F)).astype(F)
                        nrm = normalize((pos - m["position"].astype(F)).astype(F))
                        closest = dict(distance=t, position=pos, normal=nrm, material=int(m["material_id"]),
                                       front_face=bool(dot(d, nrm) < 0))
            else:
                for child in (int(node["index"]), int(node["index"]) + 1):
                    n = self.bvh[child]
                    dst = self.ray_bounding_dst(o, d, n["bounds_min"].astype(F), n["bounds_max"].astype(F))
                    if dst != INF and dst < closest["distance"]:
                        stack[si] = child
                        si += 1
        return closest

    @staticmethod
    def reflect(v, n):          # raytrace.wgsl:400-402
        return (v - (F(F(2.0) * dot(v, n)) * n).astype(F)).astype(F)

    def refract(self, v, n, eta):     # raytrace.wgsl:404-409
        cos_theta = self.fmin(dot(-v, n), F(1.0))
        perp = (eta * (v + (cos_theta * n).astype(F)).astype(F)).astype(F)
        par = (F(-np.sqrt(np.abs(F(F(1.0) - dot(perp, perp))), dtype=F)) * n).astype(F)
        return (perp + par).astype(F)

    def reflectance(self, cosine, ri):  # raytrace.wgsl:411-416
        r0 = F(F(F(1.0) - ri) / F(F(1.0) + ri))
        r0 = F(r0 * r0)
        x = F(F(1.0) - cosine)
        if self.policy["pow"] == "exp2log2":       # pow(x, 5.0) = exp2(5 * log2(x)), in double, rounded once
            xf = float(x)
            if xf != xf or xf < 0.0:
                p5 = F(np.nan)
            elif xf == 0.0:
                p5 = F(0.0)
            elif xf == math.inf:
                p5 = F(np.inf)
            else:
                p5 = F(math.pow(2.0, 5.0 * math.log2(xf)))
        else:                                      # default: pow(x, 5) as multiplies
            x2 = F(x * x)
            p5 = F(F(x2 * x2) * x)
        return F(r0 + F(F(F(1.0) - r0) * p5))

    # raytrace.wgsl:231-299 -> (absorbed, origin, direction, attenuation)
    def scatter(self, d, hit, rng):
        mat = self.materials[hit["material"]]
        base = mat["base_color"].astype(F)
        n = hit["normal"]
        if rng.next_float() < F(mat["metallic"]):
            refl = (normalize(self.reflect(d, n)) + (F(mat["roughness"]) * rng.unit_ball()).astype(F)).astype(F)
            return bool(dot(refl, n) < 0), hit["position"], refl, base
        if rng.next_float() < F(mat["specular_transmission"]):
            ri = F(F(1.0) / F(mat["ior"])) if hit["front_face"] else F(mat["ior"])
            u = normalize(d)
            cos_theta = self.fmin(dot(-u, n), F(1.0))
            sin_theta = np.sqrt(F(F(1.0) - F(cos_theta * cos_theta)), dtype=F)
            cannot = F(ri * sin_theta) > F(1.0)
            if self.policy["or_short_circuit"]:
                reflects = bool(cannot) or bool(self.reflectance(cos_theta, ri) > rng.next_float())
            else:
                refl = self.reflectance(cos_theta, ri)
                draw = rng.next_float()        # both operands of || are evaluated
                reflects = bool(cannot or refl > draw)
            direction = self.reflect(u, n) if reflects else self.refract(u, n, ri)
            return False, hit["position"], direction, v3(1, 1, 1)
        b1 = rng.unit_ball()
        b2 = rng.unit_ball()
        sd = ((n + b1).astype(F) + (F(mat["roughness"]) * b2).astype(F)).astype(F)
        if abs(sd[0]) < F(1e-8) and abs(sd[1]) < F(1e-8) and abs(sd[2]) < F(1e-8):
            sd = n
        return bool(dot(sd, n) < 0), hit["position"], sd, base

    # raytrace.wgsl:174-224
    def raytrace(self, o, d, rng):
        far = F(self.cam["far"])
        fallback_far = F(far + F(10.0)) if self.level == 1 else F(far - F(1.0))
        first_depth = INF
        color = v3(1, 1, 1)
        light = v3(0, 0, 0)
        b = 0
        maxb = int(self.cam["bounce_count"])
        while b <= maxb:
            hit = self.raycast(o, d)
            if b == 0:
                first_depth = hit["distance"]
            if hit["distance"] == INF:
                u = normalize(d)
                a = F(F(0.5) * F(u[1] + F(1.0)))
                light = ((F(F(1.0) - a) * v3(1, 1, 1)).astype(F) + (a * v3(0.5, 0.7, 1.0)).astype(F)).astype(F)
                break
            absorbed, o, d, att = self.scatter(d, hit, rng)
            if absorbed:
                break
            color = (color * att).astype(F)
            b += 1
        if b == maxb + 1:
            color = v3(0, 0, 0)
        if first_depth == INF:
            first_depth = fallback_far
        return np.sqrt((color * light).astype(F), dtype=F), first_depth

    # raytrace.wgsl:93-123, 139-172
    def fragment(self, px, py, W, H, raster_rgba=None, raster_depth=None):
        cam, win = self.cam, self.win
        uvx = F(F(F(px) + F(0.5)) / F(W))
        uvy = F(F(F(py) + F(0.5)) / F(H))
        rng = Rng(f32_to_u32(F(F(F(F(win["random_seed"]) * F(10000.0)) * F(uvx * F(402.0))) * F(uvy * F(31.5)))))
        raster = np.zeros(4, F) if raster_rgba is None else raster_rgba[py, px].astype(F)
        if self.level == 0:
            return raster
        cd, cu = cam["direction"].astype(F), cam["up"].astype(F)
        right = cross(cd, cu)
        aspect = F(cam["aspect"])
        height = F(win["height"])
        width = F(height * aspect)
        total, total_d = v3(0, 0, 0), F(0.0)
        spp = int(cam["sample_count"])
        for _ in range(spp):
            rx = F(rng.next_float() - F(0.5))
            ry = F(rng.next_float() - F(0.5))
            ndc_x = F(F(F(uvx * F(2.0)) - F(1.0)) + F(F(F(1.0) / width) * rx))
            ndc_y = F(F(F(1.0) - F(uvy * F(2.0))) + F(F(F(1.0) / height) * ry))
            d = ((cd + (F(F(ndc_x * aspect) * self.tan_half_fov) * right).astype(F)).astype(F) +
                 (F(ndc_y * self.tan_half_fov) * cu).astype(F)).astype(F)
            col, depth = self.raytrace(cam["position"].astype(F), normalize(d), rng)
            total = (total + col).astype(F)
            total_d = F(total_d + depth)
        with np.errstate(all="ignore"):
            avg = (total / F(spp)).astype(F)
            avg_d = F(total_d / F(spp))
        if self.level in (1, 2):
            depth = F(0.0) if raster_depth is None else F(raster_depth[py, px])
            rd = F(-1.0) if avg_d > F(cam["far"]) else F(F(cam["near"]) / avg_d)
            if depth > rd:
                return raster
        return np.array([avg[0], avg[1], avg[2], F(1.0)], F)


def render(models, materials, bvh, camera, window, level, W, H, raster_rgba=None, raster_depth=None, policy=None):
    sh = Shader(models, materials, bvh, camera, window, level, policy)
    out = np.zeros((H, W, 4), F)
    for py in range(H):
        for px in range(W):
            out[py, px] = sh.fragment(px, py, W, H, raster_rgba, raster_depth)
    return out, sh.rays
