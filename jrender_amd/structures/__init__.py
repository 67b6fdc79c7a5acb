from .mesh import *
