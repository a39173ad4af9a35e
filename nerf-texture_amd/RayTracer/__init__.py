from .raytracer import RayTracer  # noqa: F401
