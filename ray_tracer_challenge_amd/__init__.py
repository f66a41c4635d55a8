"""ray_tracer_challenge_amd -- MI355X-native render path for garfieldnate/ray_tracer_challenge.

Scope: Camera::render and everything it calls per pixel, as a
hand-written HIP kernel for gfx950 behind the C ABI in include/rtc.h, plus the
host-side scene math needed to feed it bit-identical inputs.  This package is
the Python mirror of the reference's World/Shape/Material/Camera/Canvas API on
top of that ABI.  The shared library must be built first
(`python -m ray_tracer_challenge_amd.build`); there is no fallback path.
"""
from ._lib import RtcError, lib  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import (AlignCheck, CubicMap, CylindricalMap, PlanarMap, SphericalMap, TextureMap, UVCheckers, UVImage,  # noqa: F401
                  acosf, acosf_host, atan2f, atan2f_host, canvas_from_ppm)
from .api import (BoundingBox, Camera, Canvas, Checkers, Cone, Cube, Cylinder, Gradient, GroupShape, Material,  # noqa: F401
                  Pattern, Plane, PointLight, RectangleLight, Rings, Shape, Sine2D, SmoothTriangle, Sphere, Stripes, Triangle, World, cosf, cosf_host,
                  default_world, device_count, glass, metal, powf, powf_host)
