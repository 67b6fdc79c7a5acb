#!/usr/bin/env python3
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jrender_amd import _build          # noqa: E402
from variants import VARIANTS           # noqa: E402

names = sys.argv[1:] or list(VARIANTS)
with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(lambda n: _build.build(variant=n, defines=VARIANTS[n]), names):
        print(lib)
