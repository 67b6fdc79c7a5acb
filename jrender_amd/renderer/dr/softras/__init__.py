from .rasterizer import *
from .soft_rasterize import *
