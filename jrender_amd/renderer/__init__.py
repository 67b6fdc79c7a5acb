from .dr import *
from .lighting import *
from .renderer import *
from .transform import *
from .gbuffer import *
