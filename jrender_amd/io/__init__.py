from .obj import *
