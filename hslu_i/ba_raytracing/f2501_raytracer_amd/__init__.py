"""MI355X-native render loop of kije/HSLU_I.BA_RAYTRACING.F2501_raytracer.

Host-side mirror of the reference's `Scene` / `Renderer::render` / `ImageBuffer` API over the
C ABI in include/rt_hip.h (hand-written HIP kernels for gfx950, csrc/).
"""
from .config import DEFAULT_FEATURES, RenderConfig, expand_features  # noqa: F401
from .f32math import Isometry3, Rotor3, Similarity3, Vec3  # noqa: F401
from .scene import (BoundedPlane, ColorType, FlatScene, Material, PointLight, Scene, SphereData,  # noqa: F401
                    TransmissionProperties, TriangleData, maximize_value)
