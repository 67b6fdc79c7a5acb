from .n3mr import *
from .rasterizer import *
