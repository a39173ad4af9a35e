from .raymarching import *  # noqa: F401,F403
from .raymarching import (  # noqa: F401
    near_far_from_aabb, polar_from_ray, morton3D, morton3D_invert, packbits, march_rays_train,
    march_rays_train_differentiable, composite_rays_train, march_rays, composite_rays, compact_rays,
)
