from .softras import *
from .n3mr import *
