from .losses import *
