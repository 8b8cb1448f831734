"""Camera matrices with glam 0.20.5 semantics (the reference's utopian/src/camera.rs:90-107).

All arithmetic is float32. Matrices are returned column-major flattened (glam::Mat4 memory
order, the order ViewUniformData stores them in).
"""
import numpy as np

f32 = np.float32


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    return (v / np.sqrt(np.dot(v, v), dtype=f32)).astype(f32)


def look_at_rh(eye, center, up):
    """glam Mat4::look_at_rh(eye, center, up) = look_to_rh(eye, center - eye, up)."""
    eye = np.asarray(eye, dtype=f32)
    f = _normalize(np.asarray(center, dtype=f32) - eye)
    s = _normalize(np.cross(f, np.asarray(up, dtype=f32)).astype(f32))
    u = np.cross(s, f).astype(f32)
    m = np.zeros((4, 4), dtype=f32)  # m[row, col]
    m[0, 0:3] = s
    m[1, 0:3] = u
    m[2, 0:3] = -f
    m[0, 3] = -np.dot(s, eye)
    m[1, 3] = -np.dot(u, eye)
    m[2, 3] = np.dot(f, eye)
    m[3, 3] = 1
    return m


def perspective_rh(fov_y_radians, aspect, z_near, z_far):
    """glam Mat4::perspective_rh: right-handed, depth 0..1, y up (no Vulkan y flip)."""
    fov = f32(fov_y_radians)
    sin_fov, cos_fov = np.sin(f32(0.5) * fov, dtype=f32), np.cos(f32(0.5) * fov, dtype=f32)
    h = f32(cos_fov / sin_fov)
    w = f32(h / f32(aspect))
    r = f32(f32(z_far) / (f32(z_near) - f32(z_far)))
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = r
    m[3, 2] = -1
    m[2, 3] = r * f32(z_near)
    return m


def inverse(m):
    return np.linalg.inv(m.astype(np.float64)).astype(f32)


def to_glam(m):
    """row/col matrix -> 16 floats column-major."""
    return np.ascontiguousarray(m.T, dtype=f32).reshape(16)


class Camera:
    """utopian::Camera (camera.rs) reduced to what the path needs: the two matrices + position."""

    def __init__(self, position, target, fov_degrees=60.0, aspect_ratio=16.0 / 9.0, z_near=0.01, z_far=1000.0):
        self.position = np.asarray(position, dtype=f32)
        self.target = np.asarray(target, dtype=f32)
        self.fov_degrees, self.aspect_ratio, self.z_near, self.z_far = fov_degrees, aspect_ratio, z_near, z_far

    def get_view(self):
        return look_at_rh(self.position, self.target, (0.0, 1.0, 0.0))

    def get_projection(self):
        return perspective_rh(np.radians(f32(self.fov_degrees), dtype=f32), self.aspect_ratio, self.z_near, self.z_far)

    def get_position(self):
        return self.position
