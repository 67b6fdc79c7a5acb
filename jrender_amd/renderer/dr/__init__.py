from .softras import *
